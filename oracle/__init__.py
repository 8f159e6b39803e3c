"""CPU oracle for the ADT hot path -- TEST INFRASTRUCTURE ONLY.

Everything under ``oracle/`` is a CPU restatement of the reference algorithm
(pier-maker92/ADT_STR) used as the *checker* for the hand-written HIP path.
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it.  The product package (``adt_str_amd``) never does: it fails
loudly when the HIP extension is missing instead of falling back to this code.

Pinning status (see DESIGN.md "Oracle"):
  * log-mel post-processing, mixer, tokenizer, masks/collate, ADT network,
    loss, greedy sample: pinned by golden vectors captured from the reference's
    own Python (tools/make_golden.py, fixtures in tests/golden/).
  * STFT / mel-filterbank conventions come from torchaudio==2.8.0, which is not
    installed anywhere we can run: that part is "parity unpinned" (restated
    from torchaudio's published definition, cross-checked against
    transformers.audio_utils.mel_filter_bank).
"""
