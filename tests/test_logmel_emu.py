"""CPU check of K1's device logic: the kernel's per-lane phase bodies
(adt_str_amd/csrc/logmel_phases.h) are compiled for the host and run lane by
lane (tests/emu/logmel_emu.cpp); the result must match the oracle.  Catches
index-map / twiddle / band errors without a GPU."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest
import torch

from adt_str_amd.frontend import MelBands, frame_geometry, melscale_fbanks
from oracle import logmel as o_logmel

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def emu(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("emu") / "liblogmel_emu.so")
    subprocess.check_call(["g++", "-O2", "-shared", "-fPIC", "-o", so, os.path.join(HERE, "emu", "logmel_emu.cpp")])
    lib = C.CDLL(so)
    lib.emu_logmel.restype = C.c_int
    return lib


def run_emu(lib, wave, sr, n_mels=128):
    hop = int(0.01 * sr)
    B, L = wave.shape
    frame_lo, n_out = frame_geometry(L, hop, 2048)
    fb = melscale_fbanks(sr, 2048, n_mels, 20.0).numpy()
    bands = MelBands.from_dense(fb)
    win = torch.hann_window(2048, periodic=True).numpy()
    out = np.zeros((B, n_out, n_mels), np.float32)
    P = lambda a: a.ctypes.data_as(C.c_void_p)
    lib.emu_logmel(P(wave), C.c_long(B), C.c_int(L), C.c_long(L), C.c_int(hop), C.c_int(frame_lo), C.c_int(n_out),
                   P(win), P(bands.meta), P(bands.weights), C.c_int(n_mels), C.c_float(1e-10), C.c_float(-23.0),
                   C.c_float(12.0), P(out))
    return out


def test_emu_matches_golden(emu, golden_dir):
    g = np.load(os.path.join(golden_dir, "logmel.npz"))
    for name in ("16k", "24k"):
        wave = np.ascontiguousarray(g[f"{name}_wave"])
        got = run_emu(emu, wave, int(g[f"{name}_sr"]))
        assert got.shape == g[f"{name}_out"].shape
        assert np.abs(got - g[f"{name}_out"]).max() < 2e-5


def test_emu_edge_clips_and_reflection(emu, golden_dir):
    """All-zero clip (clamp floor), full-scale square wave, and a hop for which
    kept frames reach into the reflect padding (hop 700 -> pad 2, frame 2 starts
    at sample 376 - 1024 < 0)."""
    g = np.load(os.path.join(golden_dir, "logmel.npz"))
    wave = np.ascontiguousarray(g["16k_edge_wave"])
    got = run_emu(emu, wave, 16000)
    ref64 = o_logmel.logmel_f64(wave, 16000, 2048, 0.01, 128)
    assert np.all(got[1] == 0.0)                                  # silent clip -> exactly the clamp floor
    assert np.abs(got[0] - g["16k_edge_out"][0]).max() < 2e-5
    # square wave: fp32 rounding dominates the weak bands; judge both against float64
    err_emu = np.abs(got[2] - ref64[2]).max()
    err_ref = np.abs(g["16k_edge_out"][2] - ref64[2]).max()
    assert err_emu < max(2.0 * err_ref, 1e-3)
    rng = np.random.default_rng(0)
    w = (rng.standard_normal((2, 9000)) * 0.1).astype(np.float32)
    sr = 70000                                                     # hop = 700
    got = run_emu(emu, w, sr)
    ref = o_logmel.logmel(torch.from_numpy(w), sr, 2048, 0.01, 128).numpy()
    assert got.shape == ref.shape and got.shape[1] > 0
    assert np.abs(got - ref).max() < 2e-5


def test_bands_roundtrip_and_sparsity():
    for sr in (16000, 24000, 48000):
        fb = melscale_fbanks(sr, 2048, 128, 20.0).numpy()
        bands = MelBands.from_dense(fb)
        assert np.array_equal(bands.to_dense(1025), fb)
        assert bands.weights.size <= 2304 and bands.meta[:, 1].max() <= 127
        assert np.array_equal(fb, o_logmel.mel_filterbank(sr, 2048, 128).numpy())
