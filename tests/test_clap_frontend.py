"""K9 CLAP log-mel: device logic on the CPU (host emulation of the kernel's phases) and the kernel on the GPU, both
against transformers' ClapFeatureExtractor (float64), the code the reference calls (clap_encoder.py:22-23).
Tolerance: 2e-3 dB (fp32 FFT vs float64) on a dB scale that spans ~100 dB."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest
import torch

from adt_str_amd.clap_frontend import HOP, MAX_SAMPLES, N_FFT, N_FRAMES, N_MELS, htk_mel_filterbank
from adt_str_amd.frontend import MelBands
from oracle import clap as o_clap

HERE = os.path.dirname(os.path.abspath(__file__))


def make_clips(seed, lengths):
    rng = np.random.default_rng(seed)
    clips = []
    for n in lengths:
        t = np.arange(n) / 48000.0
        x = np.exp(-t * rng.uniform(5, 60)) * (0.5 * rng.standard_normal(n) + np.sin(2 * np.pi * rng.uniform(60, 9000) * t))
        clips.append((x / np.abs(x).max()).astype(np.float32))
    return clips


def check_db(got, ref):
    """Every value is within 2e-3 dB of the float64 extractor, or within an absolute power error of
    max(3e-11, 1e-13 * P) where P is the largest mel power of the two frames that share one packed FFT: the kernel
    transforms frames 2p and 2p+1 as one complex signal, so in fp32 a silent frame next to a loud onset (110 dB apart
    in a repeat-padded one-shot) inherits rounding noise ~140 dB below the loud frame."""
    pw_got, pw_ref = 10.0 ** (got / 10.0), 10.0 ** (ref / 10.0)
    frames = ref.shape[1]
    pair_peak = np.zeros_like(ref)
    peak = pw_ref.max(axis=2)
    for f in range(frames):
        mate = f + 1 if f % 2 == 0 else f - 1
        pair_peak[:, f, :] = np.maximum(peak[:, f], peak[:, min(mate, frames - 1)])[:, None]
    db_ok = np.abs(got - ref) < 2e-3
    pw_ok = np.abs(pw_got - pw_ref) < np.maximum(3e-11, 1e-13 * pair_peak)
    assert np.all(db_ok | pw_ok), float(np.abs(got - ref)[~(db_ok | pw_ok)].max())
    assert db_ok.mean() > 0.995


def test_filterbank_matches_hf():
    fe = o_clap.feature_extractor()
    assert np.allclose(htk_mel_filterbank(), fe.mel_filters, atol=1e-12)
    assert (fe.fft_window_size, fe.hop_length, fe.nb_max_samples, fe.feature_size) == (N_FFT, HOP, MAX_SAMPLES, N_MELS)
    assert MelBands.from_dense(htk_mel_filterbank().astype(np.float32)).weights.size <= 1536


@pytest.fixture(scope="module")
def emu(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("emu") / "libclap_emu.so")
    subprocess.check_call(["g++", "-O2", "-shared", "-fPIC", "-o", so, os.path.join(HERE, "emu", "clap_logmel_emu.cpp")])
    lib = C.CDLL(so)
    lib.emu_clap_logmel.restype = C.c_int
    return lib


def test_emu_matches_hf_extractor(emu):
    clips = make_clips(0, [4800, 30011, 777])
    n_frames = 40                                                 # first frames (incl. the reflected left edge) of each clip
    ref = o_clap.logmel_db(clips)[:, :n_frames]
    bands = MelBands.from_dense(htk_mel_filterbank().astype(np.float32))
    win = (0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(N_FFT) / N_FFT)).astype(np.float32)
    flat = np.concatenate(clips)
    offs = np.zeros(len(clips) + 1, np.int64); offs[1:] = np.cumsum([len(c) for c in clips])
    out = np.zeros((len(clips), n_frames, N_MELS), np.float32)
    P = lambda a: a.ctypes.data_as(C.c_void_p)
    emu.emu_clap_logmel(P(flat), P(offs), C.c_long(len(clips)), C.c_int(MAX_SAMPLES), C.c_int(HOP), C.c_int(n_frames), P(win),
                        P(bands.meta), P(bands.weights), C.c_int(N_MELS), C.c_float(1e-10), P(out))
    check_db(out, ref)


@pytest.mark.gpu
def test_gpu_matches_hf_extractor():
    from adt_str_amd.clap_frontend import ClapLogMel
    clips = make_clips(1, [4800, 96000, 31337, 480000, 1000, 239999])
    fe = ClapLogMel("cuda:0")
    feats, is_longer = fe([torch.from_numpy(c) for c in clips])
    assert feats.shape == (6, 4, N_FRAMES, N_MELS) and is_longer.shape == (6, 1) and not is_longer.any()
    ref = o_clap.logmel_db(clips)
    got = feats[:, 0].cpu().numpy()
    check_db(got, ref)
    assert torch.equal(feats[:, 1], feats[:, 0])
    with pytest.raises(ValueError):                         # mel() is the <= 10 s path; features() takes any length
        fe.mel([torch.zeros(480001)])
    # clips already on the device go through the pointer table (adt_clap_logmel_db_ptrs_f32: no concatenation pass): the same kernel, the same bits
    dev_clips = [torch.from_numpy(c).to("cuda:0") for c in clips]
    assert torch.equal(fe.mel(dev_clips), fe.mel([torch.from_numpy(c) for c in clips]))
    assert torch.equal(fe.mel([dev_clips[2][None, :], dev_clips[0]]), fe.mel([torch.from_numpy(clips[2]), torch.from_numpy(clips[0])]))


@pytest.mark.gpu
def test_ln_mean_tokens_is_layernorm_then_mean():
    """adt_ln_mean_tokens (the tower's final LayerNorm + average pool in one pass) against torch on the same rows."""
    from adt_str_amd import _ffi
    g = torch.Generator().manual_seed(5)
    for (B, T, D) in ((5, 64, 768), (3, 7, 96), (2, 1, 1024)):
        x = (torch.randn((B * T, D), generator=g) * 2 + 0.3).to("cuda:0")
        gamma, beta = (1 + 0.1 * torch.randn(D, generator=g)).to("cuda:0"), (0.1 * torch.randn(D, generator=g)).to("cuda:0")
        out32 = torch.empty((B, D), device="cuda:0")
        out16 = torch.empty((B, D), dtype=torch.bfloat16, device="cuda:0")
        _ffi.call("adt_ln_mean_tokens", x.data_ptr(), B, T, D, gamma.data_ptr(), beta.data_ptr(), 1e-5, out32.data_ptr(), out16.data_ptr(), 0)
        ref = torch.nn.functional.layer_norm(x.double(), (D,), gamma.double(), beta.double(), 1e-5).view(B, T, D).mean(1)
        assert float((out32.double() - ref).abs().max()) < 2e-6 * max(1.0, float(ref.abs().max()))
        assert torch.equal(out16, out32.bfloat16())
