"""The oracle against the golden vectors captured from the reference's own Python
(tools/make_golden.py): this is what pins the checker before it is trusted."""
import os

import numpy as np
import pytest
import torch

from oracle import adt as o_adt
from oracle import logmel as o_logmel


def test_logmel_oracle_reproduces_reference_outputs(golden_dir):
    g = np.load(os.path.join(golden_dir, "logmel.npz"))
    for name in ("16k", "24k", "16k_edge"):
        out = o_logmel.logmel(torch.from_numpy(g[f"{name}_wave"]), int(g[f"{name}_sr"]), 2048, 0.01, 128).numpy()
        assert out.shape == g[f"{name}_out"].shape
        assert np.abs(out - g[f"{name}_out"]).max() < 1e-6
        assert sorted(str(k) for k in g[f"{name}_state_keys"]) == ["compute_spec.mel_scale.fb", "compute_spec.spectrogram.window"]
    # frame geometry (model.py:79,95-97): 10 s @ 16 kHz -> 986 frames, 2.56 s @ 24 kHz -> 246
    assert o_logmel.n_out_frames(160000, 160, 2048) == 986 and o_logmel.n_out_frames(61440, 240, 2048) == 246
    assert g["24k_out"].shape[1] == 246


def test_mel_filterbank_cross_check_with_hf():
    """torchaudio's filterbank is third party and absent: cross-check the restatement against HF's
    documented equivalent (float64) -- the "parity unpinned" part of the log-mel row."""
    from transformers.audio_utils import mel_filter_bank
    for sr in (16000, 24000):
        fb = o_logmel.mel_filterbank(sr, 2048, 128).numpy()
        hf = mel_filter_bank(1025, 128, 20.0, float(sr // 2), sr, norm=None, mel_scale="htk")
        assert np.abs(fb - hf).max() < 5e-5
        assert ((fb > 0).sum(axis=0) >= 1).all() and ((fb > 0).sum(axis=1) <= 2).all()     # banded: <= 2 filters per bin


def tiny_state(g):
    state = {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("w::")}
    d = state["encoder.dense_layer.weight"].shape[0]
    state["encoder.positional_encoding.pos_embedding"] = o_adt.positional_encoding(d)
    state["decoder.positional_encoding.pos_embedding"] = o_adt.positional_encoding(d)
    return state


def test_adt_oracle_reproduces_reference_modules(golden_dir):
    g = np.load(os.path.join(golden_dir, "adt_tiny.npz"))
    state = tiny_state(g)
    cfg = dict(nhead=2, sample_rate=16000, win_length=2048, time_res=0.01, n_mels=128)
    st = {k: v.clone().requires_grad_(True) if "pos_embedding" not in k else v for k, v in state.items()}
    res = o_adt.compute_loss(st, cfg, {"wavs": g["wave"], "tokens": g["tokens"], "token_lengths": g["token_lengths"]})
    assert (res["logmel"] - torch.from_numpy(g["logmel"])).abs().max() < 1e-6
    assert (res["memory"].detach() - torch.from_numpy(g["memory"])).abs().max() < 5e-6
    assert (res["logits"].detach() - torch.from_numpy(g["logits"])).abs().max() < 5e-6
    assert abs(res["loss"].item() - float(g["loss"])) < 1e-5
    res["loss"].backward()
    for k in g.files:
        if k.startswith("g::"):
            ref = torch.from_numpy(g[k])
            assert (st[k[3:]].grad - ref).abs().max() <= 2e-5 * max(1.0, ref.abs().max().item()), k
    ids = o_adt.greedy_sample(state, cfg, torch.from_numpy(g["wave"]), max_length=10)
    assert np.array_equal(ids.numpy(), g["sample_ids"])


def test_positional_encoding_and_masks():
    pe = o_adt.positional_encoding(8, 16)[0]
    assert pe[0].tolist() == [0, 1, 0, 1, 0, 1, 0, 1]
    assert torch.allclose(pe[3, 0], torch.sin(torch.tensor(3.0))) and torch.allclose(pe[3, 1], torch.cos(torch.tensor(3.0)))
    assert o_adt.causal_mask(3).tolist() == [[False, True, True], [False, False, True], [False, False, False]]
    assert o_adt.key_padding_mask([1, 3], 3).tolist() == [[False, True, True], [False, False, False]]


def test_logmel_oracle_end_to_end_against_hf_audio_utils():
    """a1's STFT + filterbank half is torchaudio's and torchaudio is not installable: an independent end-to-end check against
    third-party code that documents itself as torchaudio-equivalent -- ``transformers.audio_utils.spectrogram`` (float64 numpy:
    periodic Hann via ``window_function``, ``center=True`` reflect padding, one-sided power spectrum) with
    ``mel_filter_bank(norm=None, mel_scale="htk")`` -- followed by the reference-owned post-processing (model.py:91-97)."""
    from transformers.audio_utils import mel_filter_bank, spectrogram, window_function
    rng = np.random.default_rng(3)
    for sr, L in ((16000, 16000), (24000, 24000)):
        hop = o_logmel.hop_length(0.01, sr)
        t = np.arange(L) / sr
        waves = [rng.standard_normal(L) * 0.05,
                 0.6 * np.exp(-t * 25.0) * np.sin(2 * np.pi * 180.0 * t) + rng.standard_normal(L) * 0.01,
                 np.clip(rng.standard_normal(L) * 0.6, -1, 1)]
        fb = mel_filter_bank(1025, 128, 20.0, float(sr // 2), sr, norm=None, mel_scale="htk")            # [1025, 128] float64
        win = window_function(2048, "hann", periodic=True)
        pad = o_logmel.trim_pad(2048, hop)
        for w in waves:
            power = spectrogram(w.astype(np.float64), win, frame_length=2048, hop_length=hop, fft_length=2048, power=2.0, center=True,
                                pad_mode="reflect", onesided=True)                                   # [1025, frames]
            assert power.shape == (1025, 1 + L // hop)
            mel = power.T @ fb                                                                       # [frames, 128]
            ref = (np.clip(np.log(mel + 1e-10), -23.0, 12.0) + 23.0) / 35.0
            ref = ref[pad:-(pad + 1)]
            got = o_logmel.logmel(torch.from_numpy(w.astype(np.float32))[None], sr, 2048, 0.01, 128)[0].numpy()
            assert got.shape == ref.shape
            # fp32 torch.stft + fp32 filterbank vs float64: 1.3e-5 on the filterbank weights, amplified by the log only where a band
            # is near silence; on the [0, 1] output the two agree to 2e-4 everywhere and 2e-5 in the mean
            assert np.abs(got - ref).max() < 2e-4 and np.abs(got - ref).mean() < 2e-5, (sr, np.abs(got - ref).max())


def _g7_cases(g):
    from oracle import clap as o_clap
    shorts = o_clap.golden_short_clips()
    for i, c in enumerate(shorts):                               # the stored inputs ARE the recipe's output (the long clip travels as its recipe)
        assert np.array_equal(c, g[f"a_clip{i}"])
    return {"a": shorts, "b": [shorts[1], o_clap.golden_long_clip()]}


def test_clap_oracle_reproduces_the_references_wrapper(golden_dir):
    """G7: oracle/clap.py (features -> audio_embeddings) against what the reference's own ``ClapWrapper.get_audio_features`` returned
    (modules/clap_encoder.py:21-54, captured by tools/make_golden.py:g7_clap) for the same clips, numpy seed and seeded weights."""
    from oracle import clap as o_clap
    g = np.load(os.path.join(golden_dir, "clap.npz"))
    model = o_clap.random_clap_model(int(g["model_seed"]))
    assert np.allclose(o_clap.weights_checksum(model), g["weights_checksum"], rtol=1e-12), "seeded ClapModel differs from the golden run's"
    for case, clips in _g7_cases(g).items():
        assert [len(c) for c in clips] == g[f"{case}_lengths"].tolist()
        np.random.seed(int(g[f"{case}_np_seed"]))
        feats, longer = o_clap.features(clips)
        assert np.array_equal(longer, g[f"{case}_is_longer"])
        assert np.array_equal(feats[:, 0, ::8], g[f"{case}_feat_ch0_every8"]) and np.array_equal(feats[:, 3, ::8], g[f"{case}_feat_ch3_every8"])
        assert np.allclose(feats.astype(np.float64).sum(axis=(2, 3)), g[f"{case}_feat_sum"], rtol=1e-12)
        out = o_clap.audio_embeddings(model, torch.from_numpy(feats), torch.from_numpy(longer))
        assert np.abs(out["pooled"].numpy() - g[f"{case}_pooled"]).max() <= 1e-5 * np.abs(g[f"{case}_pooled"]).max()
        assert np.abs(out["embedding"].numpy() - g[f"{case}_embedding"]).max() < 2e-6
    assert g["a_is_longer"].sum() == 1 and g["b_is_longer"].ravel().tolist() == [False, True]
