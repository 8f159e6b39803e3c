#!/usr/bin/env python3
"""bench.py -- headline benchmark of the ADT hot path on MI355X.

Contract (driver): ``python bench.py --gpus N --steps K --warmup W``; for N > 1 it
is launched under ``torch.distributed.run`` with one rank per GPU.  W untimed
steps, then exactly K timed steps bracketed by barrier + synchronize, MAX over
ranks, rank 0 prints ONE JSON line.

Workloads (``--workload``):

  train   (default) BASELINE config[3]: one ADT training step per GPU on 64 clips of
          10 s @ 16 kHz -- the on-GPU one-shot mixer renders the batch (K2), fused
          log-mel (K1), setting-1 network forward + backward in bf16/fp32-accumulate
          (K3-K8), gradient all-reduce over RCCL when N > 1, global-norm clip + AdamW.
          Weak scaling: 64 clips per GPU, value = clips/s over all GPUs.
  logmel  BASELINE config[1]: the fused STFT->log-mel kernel alone, 256 clips per GPU.
  clap    BASELINE config[2]: CLAP curation embedding pass, 512 one-shots @ 48 kHz per GPU -> HF-extractor log-mel (K9),
          fused HTSAT audio tower + projection (K10/K11 + GEMMs), cosine arg-max against 48 class means (K12); embeds/s.

The JSON line also carries ``roofline`` (dominant kernel measured live with HIP
events on the stream it runs on) and, at N = 1, ``cpu_baseline`` (the oracle
restatement of the same step timed on the host cores on a bounded sample).

``value`` of the train workload is measured with the step's inputs (mixer plans, tokens, one-shot bank) resident in
HBM, as the contract asks.  The same line carries ``e2e``: the same step fed by the real input pipeline -- synthetic
note chunks -> ``GpuBatcher.item`` (pitch map, random velocities, tokenise) -> ``SynthDrum.plan`` -> upload -> render
-> step, nothing pre-planned, host work one batch ahead on a background thread (what ``train.py --native`` runs).

``python bench.py --gpus N`` without ``WORLD_SIZE`` in the environment starts the N ranks itself (a child
``python -m torch.distributed.run``, spawned before anything touches the GPU) and relays rank 0's JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP32_VECTOR_PEAK_TF = 157.3
BF16_MFMA_PEAK_TF = 2500.0     # dense bf16 MFMA peak (no sparsity)

def newest_profile(name):
    """Path (relative to the repo root) of ``name`` in the newest ``profiles/rNN/`` that holds it, or None."""
    import glob
    import re
    rounds = sorted((int(m.group(1)), d) for d in glob.glob(os.path.join(ROOT, "profiles", "r*"))
                    for m in [re.fullmatch(r"r(\d+)", os.path.basename(d))] if m)
    for _, d in reversed(rounds):
        if os.path.exists(os.path.join(d, name)):
            return os.path.relpath(os.path.join(d, name), ROOT)
    return None


def pmc_traffic(name):
    """(HBM-side bytes per launch of the roofline kernel, where that number comes from).  PMC counters cannot be collected from
    inside this process, so the figure is REPLAYED from the newest committed rocprofv3 --pmc summary of the same launch
    (tools/run_pmc_*.sh; FETCH_SIZE / WRITE_SIZE are in KiB, and gfx950 tallies 128-byte read requests at 64 bytes:
    MI355X_MICROARCH.md, HBM section) and the bench line says so.  (None, None) when there is no summary."""
    rel = newest_profile(name)
    if rel is None:
        return None, None
    try:
        with open(os.path.join(ROOT, rel)) as f:
            d = json.load(f)
        return (2.0 * d["FETCH_SIZE"]["mean"] + d["WRITE_SIZE"]["mean"]) * 1024.0, f"replayed from {rel} (separate rocprofv3 --pmc passes of this launch; not measured in this run)"
    except (OSError, KeyError, ValueError):
        return None, None


SETTING1 = dict(enc_layers=4, dec_layers=4, nhead=6, d_query=128, tgt_vocab_size=1400, n_mels=128)


# ----------------------------------------------------------------------------- log-mel workload (config[1])
def logmel_setup(dev, seed):
    from adt_str_amd.frontend import ComputeMelSpectrogram
    B, L, sr = 256, 160000, 16000
    g = torch.Generator().manual_seed(1234 + seed)
    wave = (torch.randn(B, L, generator=g) * 0.05)
    t = torch.arange(L) / sr
    for b in range(0, B, 4):                      # decaying bursts on a quarter of the clips
        t0 = float(torch.rand(1, generator=g)) * 9.0
        wave[b] += 0.5 * torch.exp(-(t - t0).clamp(min=0) * 30.0) * (t >= t0) * torch.sin(2 * torch.pi * 180.0 * (t - t0))
    wave[::16] = 0.0                              # every 16th clip silent
    wave = wave.clamp_(-1, 1)
    mod = ComputeMelSpectrogram(sr, 2048, 0.01, 128)
    wave_d = wave.to(dev)
    F = mod(wave_d[:1]).shape[1]
    algo_bytes = B * (4 * L + 4 * F * 128)        # read the wave once + write the output once
    # per frame: real FFT 2.5*N*log2(N) + power 3/bin + banded mel 4/bin + log/scale 3/mel
    algo_flops = B * F * (2.5 * 2048 * 11 + 3 * 1025 + 4 * 1025 + 3 * 128)

    def roofline():
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        for _ in range(20):
            mod(wave_d)
        ev1.record()
        torch.cuda.synchronize()
        ms = ev0.elapsed_time(ev1) / 20
        ach = algo_bytes / (ms * 1e-3) / 1e9
        traffic, src = pmc_traffic("logmel_pmc_summary.json")
        return {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                "traffic": traffic, "traffic_source": src, "kernel": "adt::logmel_kernel", "kernel_ms": ms, "algorithmic_bytes_per_launch": algo_bytes,
                "fp32_vector_tflops": algo_flops / (ms * 1e-3) / 1e12,
                "fp32_vector_frac": algo_flops / (ms * 1e-3) / 1e12 / FP32_VECTOR_PEAK_TF}

    def cpu_baseline(budget_s=12.0):
        from oracle import logmel as o_logmel
        torch.set_num_threads(min(os.cpu_count() or 1, 16))   # 16 = the reference's DataLoader worker count (setting-1.yaml:11)
        n = 64
        o_logmel.logmel(wave[:8], 16000, 2048, 0.01, 128)
        done, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < budget_s:
            i = (done // n) % (wave.shape[0] // n)
            o_logmel.logmel(wave[i * n:(i + 1) * n], 16000, 2048, 0.01, 128)
            done += n
        dt = time.perf_counter() - t0
        return {"value": done / dt, "unit": "clips/s", "cores": torch.get_num_threads(), "kind": "port",
                "sample": f"{done} clips of the same batch in {dt:.1f} s (oracle/logmel.py: torch.stft + dense mel matmul, fp32)"}

    return {"step": lambda: mod(wave_d), "units": B, "dtype": "f32", "roofline": roofline, "cpu_baseline": cpu_baseline,
            "metric": "ADT hot path clips/sec (10 s @16 kHz), log-mel front end stage",
            "config": {"workload": "logmel config[1]: 256 clips x 10 s @ 16 kHz -> [256,986,128], n_fft 2048, hop 160",
                       "clips_per_gpu": B, "samples": L, "sample_rate": sr}}


# ----------------------------------------------------------------------------- CLAP curation workload (config[2])
HTSAT_FLOPS_PER_CLIP = 2 * 5.91e9          # 5.91 GMAC per clip


def clap_setup(dev, seed):
    from adt_str_amd.clap_encoder import ClapWrapper, random_init_clap_model
    from adt_str_amd import _ffi
    B, n_classes = 512, 48
    rng = np.random.default_rng(7 + seed)
    clips = []
    for _ in range(B):                        # decaying noise / sine one-shots, peak-normalised
        n = int(rng.integers(4800, 96001))
        t = np.arange(n, dtype=np.float32) / 48000.0
        x = np.exp(-t * rng.uniform(5.0, 40.0)) * (rng.standard_normal(n).astype(np.float32) * rng.uniform(0.0, 1.0)
                                                    + np.sin(2 * np.pi * rng.uniform(40.0, 4000.0) * t))
        clips.append(torch.from_numpy((x / np.abs(x).max()).astype(np.float32)).unsqueeze(0))
    model = random_init_clap_model(0)
    wrap = ClapWrapper("random-init laion/clap-htsat-fused architecture", dev, 48000, clap_model=model)
    is_longer = torch.zeros(B, dtype=torch.bool)
    is_longer[int(rng.integers(0, B))] = True  # the extractor flags one random clip of an all-short batch
    means = torch.nn.functional.normalize(torch.randn(n_classes, 512, generator=torch.Generator().manual_seed(1)), dim=-1).to(dev)
    clips_d = [c.to(dev) for c in clips]
    best = torch.empty(B, dtype=torch.int32, device=dev)
    score = torch.empty(B, dtype=torch.float32, device=dev)
    state = {}

    def step():
        emb = wrap.get_audio_features(clips_d, is_longer=is_longer)
        _ffi.call("adt_cosine_argmax_f32", _ffi.dptr(emb), emb.stride(0), _ffi.dptr(means), B, 512, n_classes, 1e-8, _ffi.dptr(best),
                  _ffi.dptr(score), None, _ffi.current_stream())
        state["emb"] = emb

    def roofline():
        mel = wrap.features.mel(clips_d)
        for _ in range(2):
            wrap.encoder.forward(mel, is_longer)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        for _ in range(5):
            wrap.encoder.forward(mel, is_longer)
        ev1.record()
        torch.cuda.synchronize()
        ms = ev0.elapsed_time(ev1) / 5
        fl = HTSAT_FLOPS_PER_CLIP * B
        ach = fl / (ms * 1e-3) / 1e12
        traffic, src = pmc_traffic("clap_pmc_summary.json")
        return {"bound": "mfma", "achieved": ach, "peak": BF16_MFMA_PEAK_TF, "unit": "TFLOP/s", "frac": ach / BF16_MFMA_PEAK_TF, "traffic": traffic,
                "traffic_source": src,
                "kernel": "HTSAT encoder forward (all launches of HtsatEncoder.forward, %d clips)" % B, "kernel_ms": ms,
                "algorithmic_flops_per_launch": fl}

    def cpu_baseline(budget_s=20.0):
        from oracle import clap as o_clap          # the checker, timed as the CPU baseline (the only use of oracle/ here)
        torch.set_num_threads(min(os.cpu_count() or 1, 16))
        nb = 4
        done, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < budget_s or done == 0:
            sub = [c.reshape(-1).numpy() for c in clips[done % B:done % B + nb]]
            mel = torch.from_numpy(o_clap.logmel_db(sub)).contiguous()
            o_clap.audio_embeddings(model, mel.unsqueeze(1).repeat(1, 4, 1, 1), torch.zeros(len(sub), 1, dtype=torch.bool))
            done += len(sub)
        dt = time.perf_counter() - t0
        return {"value": done / dt, "unit": "embeds/s", "cores": torch.get_num_threads(), "kind": "port",
                "sample": f"{done} clips (batches of {nb}) in {dt:.1f} s: transformers ClapFeatureExtractor (float64 numpy) + "
                          "ClapAudioModel + audio_projection, fp32 (oracle/clap.py)"}

    return {"step": step, "units": B, "unit": "embeds/s", "dtype": "bf16", "roofline": roofline, "cpu_baseline": cpu_baseline,
            "metric": "CLAP embeds/sec",
            "config": {"workload": "clap config[2]: 512 one-shots @ 48 kHz (4 800..96 000 samples) per GPU -> HF-extractor log-mel -> fused "
                                   "HTSAT (random-init laion/clap-htsat-fused architecture, one is_longer item) -> [512,512] unit embeddings",
                       "clips_per_gpu": B, "sample_rate": 48000}}


def clap_leg(dev, steps=10, warmup=3, cpu_budget_s=8.0, cpu_baseline=True):
    """The second half of BASELINE's metric ("...; CLAP embeds/sec") inside the default line: config[2] on this rank's GPU, timed like the
    main workload (``warmup`` untimed, then ``steps`` passes of 512 clips bracketed by synchronize), with its own roofline and a short
    CPU-baseline sample.  ``bench.py --workload clap`` is the same workload as a line of its own (longer CPU sample)."""
    wl = clap_setup(dev, 0)
    for _ in range(warmup):
        wl["step"]()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        wl["step"]()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out = {"metric": wl["metric"], "value": wl["units"] * steps / dt, "unit": wl["unit"], "ms_per_step": dt / steps * 1e3, "steps": steps,
           "warmup": warmup, "dtype": wl["dtype"], "config": wl["config"], "roofline": wl["roofline"]()}
    if cpu_baseline:
        out["cpu_baseline"] = wl["cpu_baseline"](cpu_budget_s)
    return out


# ----------------------------------------------------------------------------- training-step workload (config[3])
def synthetic_notes(rng, n_clips, n_notes=40):
    """Onsets in [0, 2.95] s (the tokenizer's range, midi_tokenizer.py:54-56), custom-GM pitches, velocities 1..127."""
    batch = []
    for _ in range(n_clips):
        onset = np.sort(rng.uniform(0.0, 2.95, n_notes))
        batch.append([[float(o), float(o + 0.1), float(rng.integers(35, 61)), float(rng.integers(1, 128))] for o in onset])
    return batch


def synthetic_tokens(rng, B, T):
    """tokens[B, T+1] with PAD=1 tails, lengths U[32, T+1]; collate's max -> max-1 rule applied."""
    lens = rng.integers(32, T + 2, B)
    lens[0] = T + 1
    tokens = np.full((B, T + 1), 1, np.int64)
    for b in range(B):
        n = int(lens[b])
        body = np.empty(n - 2, np.int64)
        body[0::3] = rng.integers(4, 300, len(body[0::3]))
        body[1::3] = rng.integers(335, 361, len(body[1::3]))
        body[2::3] = rng.integers(410, 527, len(body[2::3]))
        tokens[b, :n] = np.concatenate([[2], body, [3]])
    tl = np.where(lens == lens.max(), lens - 1, lens).astype(np.int64)
    return tokens, tl


def train_flops_per_clip(F, T, d=768, ffn=3072, V=1400, n_mels=128, enc=4, dec=4):
    """Forward MACs of one clip (SURVEY 8d formula) -> fwd+bwd FLOPs = 2 * 3 * MACs."""
    macs = F * n_mels * d + F * d * d
    macs += enc * (F * d * 3 * d + 2 * F * F * d + F * d * d + 2 * F * d * ffn)
    macs += dec * (T * d * 3 * d + 2 * T * T * d + T * d * d + T * d * d + F * d * 2 * d + 2 * T * F * d + T * d * d + 2 * T * d * ffn)
    macs += T * d * V
    return 6.0 * macs


FP32_MFMA_PEAK_TF = 157.3                  # MI355X_MICROARCH.md: f32-input MFMA = the fp32 vector rate, 1/16 of bf16


def ring_allreduce_model(grad_bytes, step_ms, world=8):
    """Ring arithmetic for the step's gradient all-reduce at ``world`` ranks of one xGMI node (MI355X_MICROARCH.md: 7 links x ~153 GB/s per
    GPU, point to point): a ring moves 2 (N - 1) / N of the buffer over every GPU's links; ONE ring is bound by one link, RCCL's parallel
    rings by all seven at best.  Reported as the share of this run's measured step the collective would need if nothing overlapped it --
    an upper bound on what the reducer (segments reduced under the remaining backward pass) has to hide, not a measurement."""
    out = {"grad_bytes_f32": grad_bytes, "world": world, "step_ms": step_ms, "link_gb_s": 153.0, "links": 7,
           "how": "2 (N-1)/N x bytes / bandwidth; one ring = one link, seven rings = all links (upper bound of RCCL on xGMI); arithmetic, not measured"}
    for wire, nbytes in (("f32", grad_bytes), ("bf16", grad_bytes / 2)):
        moved = 2.0 * (world - 1) / world * nbytes
        one, seven = moved / 153e9 * 1e3, moved / (7 * 153e9) * 1e3
        out[wire] = {"one_ring_ms": one, "seven_rings_ms": seven, "share_of_step_one_ring": one / step_ms, "share_of_step_seven_rings": seven / step_ms}
    return out


def train_setup(dev, seed, world, dropout, fx_prob=0.0, process_group=None, grad_compress=None, precision="bf16", input_sec=10.0, sample_rate=16000):
    from adt_str_amd import kernels as K
    from adt_str_amd.bank import OneShotBank, synthetic_tree
    from adt_str_amd.network import ADTModel, ADTModelConfig
    from adt_str_amd.synth import SynthDrum, SynthDrumConfig
    from adt_str_amd.trainer import FlatTrainer
    B, sr, T = 64, int(sample_rate), 128
    L = int(round(input_sec * sr))
    torch.manual_seed(0)                                   # same initial weights on every rank (then broadcast anyway)
    cfg = ADTModelConfig(input_sec=input_sec, time_res=0.01, win_length=2048, sample_rate=sr, dropout=dropout, plain=True, **SETTING1)
    model = ADTModel(cfg).to(dev)
    if precision != "bf16":
        model.set_precision(precision)                    # the fp32-operand parity arm (csrc/precise.hip): logits within 1e-3 rel of the CPU reference
    trainer = FlatTrainer(model, lr=1e-4, weight_decay=1e-5, max_grad_norm=1.0, total_steps=10000, warmup_ratio=0.1,
                          process_group=process_group, grad_compress=grad_compress, comm_timing=True)
    bank = OneShotBank.from_tree(synthetic_tree(7, sr), sr)
    synth = SynthDrum(SynthDrumConfig(input_sec=input_sec, time_res=0.01, win_length=2048, sample_rate=sr, oneshot_path="synthetic",
                                      similarity_threshold=0.8, max_hat_std_velocity=0.15, max_hat_mean_velocity=0.1,
                                      max_cymbals_std_velocity=0.15, max_cymbals_mean_velocity=0.65, ADTOF_mapping=False,
                                      mixup_range=0.8, use_fx_prob=fx_prob, use_reverb_prob=0.5, use_limiter_prob=0.5,
                                      use_compression_prob=0.5), bank=bank, device=str(dev))
    rng = np.random.default_rng(100 + seed)
    random.seed(100 + seed)
    n_plans = 4                                             # distinct synthetic batches, cycled
    plans = [synth.plan(synthetic_notes(rng, B)) for _ in range(n_plans)]
    toks = [synthetic_tokens(rng, B, T) for _ in range(n_plans)]
    toks_d = [(torch.from_numpy(t).to(dev), torch.from_numpy(l).to(dev)) for t, l in toks]
    wav_buf = torch.empty((B, L), dtype=torch.float32, device=dev)
    state = {"i": 0, "loss": None}

    def step():
        i = state["i"] % n_plans
        state["i"] += 1
        # K2: render the batch on the GPU (with FX: on the synth's own stream into a fresh buffer, see SynthDrum.render_plan)
        wav = synth.render_plan(plans[i], width=L) if fx_prob > 0 else synth.render_plan(plans[i], width=L, out=wav_buf)
        tok, tl = toks_d[i]
        state["loss"] = trainer.train_step(wav, tok, tl)                   # K1 + network fwd/bwd + all-reduce + clip + AdamW

    F = model.compute_spectrogram(wav_buf[:1].zero_()).shape[1]
    flops_clip = train_flops_per_clip(F, T)

    def e2e(steps, warmup, fence):
        """The same step behind the real input pipeline (nothing pre-planned): note chunks -> item -> plan -> upload -> render ->
        step, host work one batch ahead on a background thread.  Returns seconds for ``steps`` steps."""
        from adt_str_amd.data import GpuBatcher, NoteChunkDataset, Prefetcher
        from adt_str_amd.tokenizer import MidiTokenizer, MidiTokenizerConfig
        r2 = np.random.default_rng(500 + seed)
        gm_keys = np.array([35, 36, 38, 40, 41, 42, 43, 44, 45, 46, 47, 48, 49, 50, 51, 53, 57, 59], np.float32)   # GM drum keys
        rows = []
        for _ in range(B * (steps + warmup)):
            n = 42                                                          # 3 * 42 + 2 = 128 tokens per clip
            onset = np.sort(r2.uniform(0.0, 2.95, n)).astype(np.float32)
            rows.append(np.stack([onset, onset + np.float32(0.1), r2.choice(gm_keys, n), r2.integers(1, 128, n).astype(np.float32)], 1)
                        .astype(np.float32).tobytes())
        tk = MidiTokenizer(MidiTokenizerConfig(ADTOF_mapping=False, BOS_token=2, EOS_token=3, pad_token=1, silence_token=0, add_velocity=True))
        ds = NoteChunkDataset(rows, GpuBatcher(tk, synth, empty_tokens_percentage=0.05, random_velocity_prob=0.5))
        pf = Prefetcher(lambda s: ds.host_batch(range(s * B, (s + 1) * B)), steps + warmup, depth=2)
        t0 = None
        try:
            for s, hb in enumerate(pf):
                if s == warmup:
                    fence()
                    t0 = time.perf_counter()
                batch = ds.batcher.upload(hb, width=L, device_tokens=True)
                state["loss"] = trainer.train_step(batch["wavs"], batch["tokens"], batch["token_lengths"])
        finally:
            pf.close()
        fence()
        return time.perf_counter() - t0

    # The dominant kernel -- the NT bf16 GEMM at the encoder's FFN linear1 (bias + GELU + dropout + saved gelu' * keep factor) -- is
    # timed WHERE THE STEP LAUNCHES IT: every such launch inside the timed region is bracketed by HIP events on the launch stream
    # (kernels.gemm_tap), 4 per step.  `--roofline-loop` adds the earlier rounds' measurement (the same launch back to back, alone).
    Mr, Nr, Kr = B * F, 3072, 768
    tap = {"match": lambda trans, M, N, Kd, ep: (not trans) and (M, N, Kd) == (Mr, Nr, Kr) and ep.act == 1 and ep.act_grad_mode == 1,
           "events": []}

    def tap_on(on):
        K.gemm_tap = tap if on else None

    def roofline_fp32():
        # the parity arm launches no bf16 GEMM: its roofline line is the whole step against the f32-input MFMA peak (every product of the
        # step runs on v_mfma_f32_32x32x2_f32 in precise.hip)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        for _ in range(2):
            step()
        ev1.record()
        torch.cuda.synchronize()
        ms = ev0.elapsed_time(ev1) / 2
        ach = flops_clip * B / (ms * 1e-3) / 1e12
        peak = FP32_MFMA_PEAK_TF if precision == "fp32" else BF16_MFMA_PEAK_TF / 3.0
        return {"bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak, "traffic": None,
                "kernel": "whole training step on the fp32-operand kernels (adt_gemm_f32 / adt_attn_fwd_f32 / adt_attn_bwd_f32, " +
                          ("v_mfma_f32_32x32x2_f32)" if precision == "fp32" else "three v_mfma_f32_32x32x16_bf16 per product on hi / lo splits)"),
                "kernel_ms": ms, "algorithmic_flops_per_launch": flops_clip * B}

    def roofline(loop=False):
        if precision != "bf16":
            return roofline_fp32()
        M, N, Kd = Mr, Nr, Kr
        torch.cuda.synchronize()
        in_step = sorted(e0.elapsed_time(e1) for e0, e1 in tap["events"])
        fl = 2.0 * M * N * Kd
        site = K.drop_site(dropout, 1, 5)            # the step's own call: bias + GELU + dropout + saved gelu' * keep factor
        a = torch.randn((M, Kd), device=dev).bfloat16()
        w = torch.randn((N, Kd), device=dev).bfloat16()
        bias = torch.zeros(N, device=dev)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        back_to_back = None
        if loop or not in_step:
            u = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
            # The chip idles while the operands above are made; the first ~30 launches after that are a boost -> throttle transient
            # (0.42 -> 0.54 -> 0.46 ms, profiles/r01/README.md), so they are not timed: what is reported is the sustained-load duration.
            n_warm, n_timed = 40, 60
            for _ in range(n_warm):
                K.gemm(a, w, bias=bias, act=1, act_grad_out=u, drop=site)
            ev0.record()
            for _ in range(n_timed):
                K.gemm(a, w, bias=bias, act=1, act_grad_out=u, drop=site)
            ev1.record()
            torch.cuda.synchronize()
            t = ev0.elapsed_time(ev1) / n_timed
            back_to_back = {"kernel_ms": t, "achieved": fl / (t * 1e-3) / 1e12, "launches": n_timed,
                            "how": "the same launch alone, back to back, after 40 untimed launches"}
            del u
        ms = sum(in_step) / len(in_step) if in_step else back_to_back["kernel_ms"]
        ach = fl / (ms * 1e-3) / 1e12
        # the same GEMM in the two cheaper forms earlier rounds quoted: round 1's epilogue (bias + GELU + saved pre-activation, no
        # dropout) and the bare bf16 product (what a library GEMM does), so that rounds can be compared like for like
        z = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
        also = {}
        for name, fn in (("bias_gelu_preact_no_dropout_r01_form", lambda: K.gemm(a, w, bias=bias, act=1, pre_act_out=z)),
                         ("plain_bf16_product", lambda: K.gemm(a, w, out=z))):
            for _ in range(10):
                fn()
            ev0.record()
            for _ in range(30):
                fn()
            ev1.record()
            torch.cuda.synchronize()
            t = ev0.elapsed_time(ev1) / 30
            also[name] = {"kernel_ms": t, "achieved": fl / (t * 1e-3) / 1e12}
        traffic, traffic_src = pmc_traffic("gemm_pmc_summary.json")
        return {"bound": "mfma", "achieved": ach, "peak": BF16_MFMA_PEAK_TF, "unit": "TFLOP/s", "frac": ach / BF16_MFMA_PEAK_TF, "same_shape_other_epilogues": also,
                "traffic": traffic, "traffic_source": traffic_src,
                "kernel": "adt::gemm_nt_256_kernel<%s, false> (FFN linear1 + bias + GELU%s + saved gelu' factor, M=%d N=%d K=%d)"
                          % ("true" if site else "false", " + dropout" if site else "", M, N, Kd),
                "kernel_ms": ms, "algorithmic_flops_per_launch": fl,
                "algorithmic_bytes_per_launch": 2.0 * (M * Kd + N * Kd + 2 * M * N),
                "timed": ("%d launches inside the timed steps, HIP events around each on the launch stream (median %.4f, min %.4f, max %.4f ms)"
                          % (len(in_step), in_step[len(in_step) // 2], in_step[0], in_step[-1])) if in_step else "back-to-back loop (no launch of this form inside the timed steps)",
                "back_to_back": back_to_back,
                "profile": "%s (rocprofv3 --kernel-trace --stats of this command: the kernel's launches among the step's others); %s (the kernel alone, back to back, 300 launches)"
                           % (newest_profile("train_step_kernel_stats.csv") or "no committed profile", newest_profile("roofline_gemm_kernel_stats.csv") or "no committed profile")}

    def cpu_baseline(budget_s=20.0):
        from oracle import adt as o_adt
        torch.set_num_threads(min(os.cpu_count() or 1, 16))
        sd = {k: v.detach().float().cpu().clone() for k, v in model.state_dict().items()}
        st = {k: (v.requires_grad_(True) if v.is_floating_point() and "pos_embedding" not in k and "compute_spec" not in k else v)
              for k, v in sd.items()}
        ocfg = dict(nhead=6, sample_rate=sr, win_length=2048, time_res=0.01, n_mels=128)
        nb = 2
        wav = wav_buf[:nb].cpu()
        batch = {"wavs": wav.numpy(), "tokens": toks[0][0][:nb], "token_lengths": toks[0][1][:nb]}
        done, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < budget_s or done == 0:
            out = o_adt.compute_loss(st, ocfg, batch)
            out["loss"].backward()
            done += nb
        dt = time.perf_counter() - t0
        return {"value": done / dt, "unit": "clips/s", "cores": torch.get_num_threads(), "kind": "port",
                "sample": f"{done} clips (batches of {nb}) forward+backward in {dt:.1f} s, fp32, no optimizer step "
                          "(oracle/adt.py restatement of model.py:240-258)"}

    def comm():
        return trainer.reducer.comm_stats() if trainer.reducer is not None else None

    return {"step": step, "units": B, "dtype": {"bf16": "bf16", "fp32": "f32", "bf16x3": "bf16x3"}[precision], "roofline": roofline, "tap": tap_on, "cpu_baseline": cpu_baseline, "state": state,
            "flops_per_step": flops_clip * B, "e2e": e2e, "comm": comm, "grad_bytes": sum(p.numel() for p in model.parameters() if p.requires_grad) * 4,
            "metric": "ADT training clips/sec (%g s @%g kHz)" % (input_sec, sr / 1000.0),
            "config": {"workload": ("train config[3]" if (input_sec, sr) == (10.0, 16000) else "train, the reference's own operating point (configs/train/setting-1.yaml:9-11)")
                                   + ": ADT train step, setting-1 network (69.0M params), per-GPU batch 64 x %g s @ %g kHz " % (input_sec, sr / 1000.0) +
                                   "mixer-rendered clips (F=%d frames), T=128 target tokens, %s, "
                                   "AdamW + clip 1.0, dropout %.2f, use_fx_prob %.2f" % (F, {"bf16": "bf16 GEMM/attention with fp32 accumulate",
                                   "fp32": "fp32 operands everywhere (the exact parity arm that meets logits within 1e-3 rel of the CPU reference)",
                                   "bf16x3": "fp32 activations, three bf16 MFMAs per product on hi / lo splits (the fast parity arm: logits within 1e-3 rel of the CPU reference)"}[precision], dropout, fx_prob),
                       "global_batch": B * world, "clips_per_gpu": B, "samples": L, "sample_rate": sr, "target_len": T}}


BF16X3_PEAK_TF = BF16_MFMA_PEAK_TF / 3.0    # three bf16 MFMAs per product


def parity_arm(dev, args, bf16_value, precision, steps=3, warmup=1):
    """The SAME workload on a parity arm -- the path that meets BASELINE's "logits within 1e-3 rel-tol of the CPU reference"
    (tests/test_precision_gpu.py) -- timed like the main loop (``warmup`` untimed, ``steps`` timed steps bracketed by synchronize), so that
    the benchmarked bf16 arm and the parity arms stand in one driver line.  ``precision`` = "fp32" (exact f32-input MFMA products,
    csrc/precise.hip) or "bf16x3" (the same kernels and fp32 activations, every product as three bf16 MFMAs on hi / lo splits)."""
    wl = train_setup(dev, 0, 1, args.dropout, args.fx_prob, None, None, precision, args.input_sec, args.sample_rate)
    for _ in range(warmup):
        wl["step"]()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        wl["step"]()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    tf = wl["flops_per_step"] / (dt / steps) / 1e12
    value = wl["units"] * steps / dt
    peak = FP32_MFMA_PEAK_TF if precision == "fp32" else BF16X3_PEAK_TF
    what = ("the same step with fp32 operands everywhere (v_mfma_f32_32x32x2_f32): the exact parity arm, logits within 1e-3 rel of the CPU reference "
            "(measured ~1e-6); frac against the f32-input MFMA peak") if precision == "fp32" else \
           ("the same step with fp32 activations and every product as three bf16 MFMAs on hi / lo splits of the fp32 operands (bf16x3): the fast "
            "parity arm, logits within 1e-3 rel of the CPU reference (asserted at 1e-4, measured 9e-6 at config[3], in tests/test_precision_gpu.py); the large "
            "linear products run on the persistent bf16 GEMM kernels over [hi | lo] planes split once per tensor (adt_gemm_bf16x3: one GEMM over three "
            "segments of K), the attention products and the small shapes on the tiled split kernels of csrc/precise.hip; frac against a third of the "
            "dense bf16 MFMA peak")
    return {"value": value, "unit": "clips/s", "ms_per_step": dt / steps * 1e3, "steps": steps, "warmup": warmup, "dtype": "f32" if precision == "fp32" else "bf16x3",
            "precision": precision, "final_loss": float(wl["state"]["loss"].item()),
            "step_tflops_per_gpu": tf, "step_mfma_frac": tf / peak, "peak": peak, "ratio_to_bf16_value": value / bf16_value, "what": what}


def launcher_command(n_ranks: int, port: int, argv):
    """The driver's own launch line (one rank per GPU of ONE node, rendezvous on 127.0.0.1)."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_ranks}", "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def spawn_ranks(n_ranks: int, argv) -> int:
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.call(launcher_command(n_ranks, port, argv), env=env)


class ClockPoll:
    """Shader clock and board power of one GPU while the timed steps run, sampled ~50 times a second IN THIS PROCESS from sysfs (the amdgpu
    hwmon node of the card: ``freq1_input`` = sclk in Hz, ``power1_average`` / ``power1_input`` in microwatts).  No child process: under
    rocprofv3 a spawned ``rocm-smi`` (a ``#!/usr/bin/env python3`` script) would inherit the profiler's preload and exec after GPU
    initialisation -- the hop this pool forbids -- and its interpreter start-ups would compete with the launch thread inside the timed
    region.  Every compute kernel of this repo runs the board into its power limit (profiles/r04/clock_under_load.txt: 1 400 W, 1.87-2.27 GHz
    depending on the kernel), so what the MFMA pipe can deliver is the 2.4 GHz peak scaled by the clock the chip holds; the line reports both.
    No readable node (another driver layout, no permission): the line goes out without the object."""
    NOMINAL_MHZ = 2400.0
    SYSFS_DRM = "/sys/class/drm"

    def __init__(self, index):
        self.index, self.samples, self._stop, self._th = index, [], False, None
        self.freq_path, self.power_path = self.find_nodes(index, self.pci_address(index))

    @staticmethod
    def pci_address(index):
        """"dddd:bb:dd" of HIP device ``index`` (a GPU box shows the sysfs nodes of every GPU of its host, the process sees one), or None."""
        try:
            p = torch.cuda.get_device_properties(index)
            return "%04x:%02x:%02x" % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)
        except Exception:          # noqa: BLE001  (no GPU, or a torch without the PCI fields)
            return None

    @staticmethod
    def find_nodes(index, pci=None):
        """(sclk node, power node) of the amdgpu card at PCI address ``pci`` ("dddd:bb:dd"), else of the ``index``-th card in PCI order
        -- the order HIP enumerates by default -- or (None, None)."""
        import glob
        cards = []
        for dev_dir in glob.glob(os.path.join(ClockPoll.SYSFS_DRM, "card[0-9]*", "device")):
            hw = sorted(glob.glob(os.path.join(dev_dir, "hwmon", "hwmon*")))
            freq = next((q for q in ([os.path.join(hw[0], "freq1_input")] if hw else []) + [os.path.join(dev_dir, "pp_dpm_sclk")] if os.path.exists(q)), None)
            if freq is not None:
                cards.append((os.path.realpath(dev_dir), hw[0] if hw else None, freq))
        cards.sort()
        by_pci = [c for c in cards if pci is not None and os.path.basename(c[0]).lower().startswith(pci.lower())]
        if by_pci:
            cards, index = by_pci, 0
        if index >= len(cards):
            return None, None
        _, hw, freq = cards[index]
        power = next((os.path.join(hw, n) for n in ("power1_average", "power1_input") if hw and os.path.exists(os.path.join(hw, n))), None)
        return freq, power

    @staticmethod
    def _read(path):
        with open(path) as f:
            return float(f.read().strip())

    @staticmethod
    def _read_mhz(path):
        """``freq1_input``: one number in Hz; ``pp_dpm_sclk``: the level table, the current level marked with ``*`` ("1: 2100Mhz *")."""
        import re
        with open(path) as f:
            text = f.read()
        if path.endswith("pp_dpm_sclk"):
            m = re.search(r"(\d+)\s*Mhz\s*\*", text, re.I)
            if not m:
                raise ValueError("no current level in pp_dpm_sclk")
            return float(m.group(1))
        return float(text.strip()) / 1e6

    def _run(self):
        while not self._stop:
            try:
                mhz = self._read_mhz(self.freq_path)
                watts = self._read(self.power_path) / 1e6 if self.power_path else float("nan")
                self.samples.append((time.perf_counter(), int(round(mhz)), watts))
            except (OSError, ValueError):          # the node went away / is not readable: the line goes out without the object
                return
            time.sleep(0.02)

    def start(self):
        import threading
        if self.freq_path is None:
            return self
        self._th = threading.Thread(target=self._run, daemon=True)
        self._th.start()
        return self

    def stop(self, t_from):
        self._stop = True
        if self._th is not None:
            self._th.join(timeout=2)
        s = [(c, p) for (t, c, p) in self.samples if t >= t_from]
        if not s:
            return None
        clk, pw = sorted(c for c, _ in s), sorted(p for _, p in s if p == p)
        med = clk[len(clk) // 2]
        return {"sclk_mhz": {"min": clk[0], "median": med, "max": clk[-1]}, "power_w_median": pw[len(pw) // 2] if pw else None, "samples": len(s),
                "nominal_mhz": self.NOMINAL_MHZ, "held_over_nominal": med / self.NOMINAL_MHZ,
                "source": "amdgpu hwmon (freq1_input, power1_average) read in-process while the timed steps ran (samples of the timed region only)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="train", choices=["train", "logmel", "clap"])
    ap.add_argument("--dropout", type=float, default=0.1, help="model dropout (0.1 = configs/train/setting-1.yaml of the reference)")
    ap.add_argument("--fx-prob", type=float, default=0.0, help="use_fx_prob of the mixer (the reference's setting-1 trains with 0.3; SURVEY's "
                                                                 "benchmark configuration is 0)")
    ap.add_argument("--input-sec", type=float, default=10.0, help="train workload: clip length in seconds (BASELINE config[3]: 10; the reference's "
                                                                   "configs/train/setting-1.yaml trains at 2.56)")
    ap.add_argument("--sample-rate", type=int, default=16000, help="train workload: sample rate (BASELINE config[3]: 16000; setting-1.yaml: 24000)")
    ap.add_argument("--no-fp32-arm", action="store_true", help="train workload, N = 1, bf16: skip the three fp32-operand steps of the \"fp32_arm\" object")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-e2e", action="store_true", help="skip the end-to-end (real input pipeline) leg of the train workload")
    ap.add_argument("--no-clock", action="store_true", help="do not sample the shader clock / board power (sysfs, in-process) during the timed steps (the \"clock\" object of the line)")
    ap.add_argument("--no-clap", action="store_true", help="train workload, N = 1: skip the CLAP embeds/sec leg (the \"clap\" object of the line)")
    ap.add_argument("--roofline-loop", action="store_true", help="train workload: also time the roofline kernel alone, back to back (the "
                                                                  "earlier rounds' measurement; adds 100 launches of it to a profile of this command)")
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32", "bf16x3"], help="train workload: fp32 / bf16x3 = the fp32-operand parity arms (precise.hip: "
                                                                                   "exact f32 MFMA products / three bf16 MFMAs per product), the ones that meet "
                                                                                   "BASELINE's 1e-3 logits tolerance; not the benchmarked default")
    ap.add_argument("--no-parity-arm", action="store_true", help="train workload, N = 1, bf16: skip the bf16x3 steps of the \"parity_arm\" object")
    ap.add_argument("--grad-compress", default=None, choices=["bf16"], help="send bf16 copies of the gradient segments (N > 1)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # Not under a launcher: start the N ranks ourselves, as a CHILD process and before this process has touched the GPU
        # (never exec from a process that has initialised HIP), relay its output, exit with its code.
        raise SystemExit(spawn_ranks(args.gpus, sys.argv[1:]))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started {world} ranks (WORLD_SIZE={world})")
    assert torch.cuda.is_available(), "bench.py needs a GPU (there is no CPU path)"
    # ADT_BENCH_SHARE_GPU=1 (debug only): every rank on GPU 0 with the gloo backend, to exercise the N > 1 code path on a
    # one-GPU box; numbers from that mode are meaningless and the JSON line says so
    share = os.environ.get("ADT_BENCH_SHARE_GPU") == "1"
    if share:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # Under a launcher (RANK / WORLD_SIZE in the environment) the process group is created even for ONE rank, and the training step
    # then goes through the gradient reducer: ``torchrun --nproc-per-node 1 bench.py`` runs the whole RCCL path on one GPU.
    launched = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if world > 1 or launched:
        if share:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
    pg = dist.group.WORLD if dist.is_initialized() else None

    wl = {"train": lambda: train_setup(dev, rank, world, args.dropout, args.fx_prob, pg, args.grad_compress, args.precision, args.input_sec, args.sample_rate), "logmel": lambda: logmel_setup(dev, rank),
          "clap": lambda: clap_setup(dev, rank)}[args.workload]()
    step = wl["step"]

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    if "comm" in wl:
        wl["comm"]()                                     # drop the warm-up steps' wait events
    if "tap" in wl:
        wl["tap"](True)                                  # event pairs around the roofline kernel's launches inside the timed region
    poll = ClockPoll(local_rank).start() if (rank == 0 and not args.no_clock) else None
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    clock = poll.stop(t0) if poll is not None else None
    if "tap" in wl:
        wl["tap"](False)
    rank_ms = None
    if world > 1:
        # value is computed from the MAX over ranks; the spread is reported next to it so that a slow rank / a skewed launch is
        # visible in the first multi-GPU line (per-rank milliseconds per step of the timed region)
        each = [torch.zeros(1, device=dev, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(each, torch.tensor([dt], device=dev, dtype=torch.float64))
        per = [float(t.item()) / args.steps * 1e3 for t in each]
        rank_ms = {"min": min(per), "max": max(per), "per_rank": per}
        dt = max(float(t.item()) for t in each)
    comm = wl["comm"]() if "comm" in wl else None     # the timed steps' collective waits (read before the e2e leg adds its own)
    dt_e2e = None
    if "e2e" in wl and not args.no_e2e:
        dt_e2e = wl["e2e"](args.steps, max(2, args.warmup), fence)
        if world > 1:
            tt = torch.tensor([dt_e2e], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt_e2e = float(tt.item())

    if rank == 0:
        units = wl["units"] * world * args.steps
        line = {"metric": wl["metric"], "value": units / dt, "unit": wl.get("unit", "clips/s"), "n_gpus": world, "steps": args.steps,
                "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
                "vs_baseline": None, "dtype": wl["dtype"], "data": "synthetic",
                "config": dict(wl["config"], parallelism=f"dp{world}")}
        if dist.is_initialized():                        # what the collective library itself reports
            line["collective"] = {"backend": dist.get_backend(), "world_size": dist.get_world_size()}
        if rank_ms is not None:
            line["ms_per_step_ranks"] = rank_ms
        if comm is not None:
            # bytes one rank hands to the collective per step; exposed_wait_ms = time the compute stream sat behind the
            # all-reduces in GradReducer.finish() (event-bracketed on that stream), mean over the timed steps on rank 0
            line["comm"] = comm
        if dt_e2e is not None:
            line["e2e"] = {"value": units / dt_e2e, "unit": wl.get("unit", "clips/s"), "ms_per_step": dt_e2e / args.steps * 1e3,
                           "ratio_to_value": dt / dt_e2e,
                           "pipeline": "synthetic note chunks (42 notes, GM keys) -> GpuBatcher.item (5 % empty, random velocities, tokenise) -> "
                                       "SynthDrum.plan -> upload + render -> train step; host work one batch ahead on a background thread; "
                                       "nothing pre-planned"}
        if "flops_per_step" in wl:
            tf = wl["flops_per_step"] / (dt / args.steps) / 1e12
            line["step_tflops_per_gpu"] = tf
            line["step_mfma_frac"] = tf / {"bf16": BF16_MFMA_PEAK_TF, "f32": FP32_MFMA_PEAK_TF, "bf16x3": BF16_MFMA_PEAK_TF / 3.0}.get(wl["dtype"], BF16_MFMA_PEAK_TF)
            loss = wl["state"]["loss"]
            line["final_loss"] = float(loss.item()) if loss is not None else None
        if os.environ.get("ADT_BENCH_SHARE_GPU") == "1":
            line["data"] = "synthetic; DEBUG RUN: all ranks share GPU 0 over gloo (ADT_BENCH_SHARE_GPU=1), not a measurement"
        line["roofline"] = wl["roofline"](args.roofline_loop) if args.workload == "train" else wl["roofline"]()
        if clock is not None:
            line["clock"] = clock
            if line["roofline"].get("bound") == "mfma" and line["roofline"].get("frac") is not None:
                # the same achieved rate against the MFMA peak at the clock the chip held during the timed steps (power-limited: see ClockPoll)
                line["roofline"]["frac_at_held_clock"] = line["roofline"]["frac"] / clock["held_over_nominal"]
            if "step_mfma_frac" in line:
                line["step_mfma_frac_at_held_clock"] = line["step_mfma_frac"] / clock["held_over_nominal"]
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = wl["cpu_baseline"]()
        if args.workload == "train" and "grad_bytes" in wl:
            line["allreduce_model_8_ranks"] = ring_allreduce_model(wl["grad_bytes"], line["ms_per_step"])
        if world == 1 and args.workload == "train" and args.precision == "bf16" and not args.no_parity_arm:
            line["parity_arm"] = parity_arm(dev, args, line["value"], "bf16x3", steps=5, warmup=2)
        if world == 1 and args.workload == "train" and args.precision == "bf16" and not args.no_fp32_arm:
            line["fp32_arm"] = parity_arm(dev, args, line["value"], "fp32")
        if world == 1 and args.workload == "train" and not args.no_clap:
            line["clap"] = clap_leg(dev, cpu_baseline=not args.no_cpu_baseline)
        print(json.dumps(line), flush=True)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
