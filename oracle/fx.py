"""Oracle (test infrastructure): the reference's optional FX chain (modules/synthetiser.py:30-87,121-137,154-155).

Two parts:
  * ``sample_board`` -- ``BoardChain.get_board``: which effects and which parameters, drawn from Python's ``random`` and
    ``torch.randn`` in the reference's order (``utils/utils.py:266-269`` for the normal draws).  Pinned by
    ``tests/golden/fx_params.npz`` (captured from the reference with recording stand-ins for the pedalboard classes).
  * ``reverb_mono`` / ``compressor`` / ``limiter`` -- what ``pedalboard.Reverb / Compressor / Limiter`` do to a mono float32
    signal.  pedalboard wraps JUCE (``juce::Reverb``, ``juce::dsp::Compressor``, ``juce::dsp::Limiter``); neither is available in
    this image, so these restate JUCE's published algorithms sample by sample.  **Parity unpinned**: no vector pins them.

Not reproduced: the reference keeps ONE ``Pedalboard`` per ``VolumeMixer`` and appends to it on every call
(synthetiser.py:40,81-87), so its chain grows during a run; here every clip gets a fresh chain."""
from __future__ import annotations

import math
import random
from typing import List, Tuple

import numpy as np
import torch

COMB_TUNINGS = (1116, 1188, 1277, 1356, 1422, 1491, 1557, 1617)
ALLPASS_TUNINGS = (556, 441, 341, 225)


def draw_from_normal_distribution(std: float, mean: float, high_bound: float, low_bound: float) -> float:
    """utils/utils.py:266-269 (one ``torch.randn(1)`` from the global CPU generator)."""
    return torch.clamp(torch.clamp(torch.randn(1) * std + mean, -1.0, 1.0).abs() * high_bound, low_bound, high_bound).item()


def sample_board(use_reverb_prob: float, use_compression_prob: float, use_limiter_prob: float) -> List[Tuple[str, dict]]:
    """``BoardChain.get_board`` on a fresh board (synthetiser.py:44-87)."""
    board = []
    if random.random() < use_reverb_prob:
        room_size = random.uniform(0.2, 0.8)
        damping = random.uniform(0.2, 0.8)
        wet_level = random.uniform(0.1, 0.4)
        width = random.uniform(0.6, 1.0)
        board.append(("Reverb", dict(room_size=room_size, damping=damping, wet_level=wet_level, dry_level=1 - wet_level, width=width,
                                     freeze_mode=0.0)))
    if random.random() < use_compression_prob:
        threshold = -draw_from_normal_distribution(std=0.15, mean=0.5, high_bound=10, low_bound=0)
        ratio = draw_from_normal_distribution(std=0.15, mean=0.5, high_bound=10, low_bound=1.0)
        attack = draw_from_normal_distribution(std=0.05, mean=0.1, high_bound=1000, low_bound=0)
        release = draw_from_normal_distribution(std=0.15, mean=0.2, high_bound=1000, low_bound=0)
        board.append(("Compressor", dict(threshold_db=threshold, ratio=ratio, attack_ms=attack, release_ms=release)))
    if random.random() < use_limiter_prob:
        threshold = -draw_from_normal_distribution(std=0.2, mean=0.4, high_bound=3, low_bound=0)
        board.append(("Limiter", dict(threshold_db=threshold)))
    return board


# ------------------------------------------------------------------------------------------------ JUCE restatements (float32)
def reverb_sizes(sample_rate: int):
    sr = int(sample_rate)
    return [(sr * t) // 44100 for t in COMB_TUNINGS], [(sr * t) // 44100 for t in ALLPASS_TUNINGS]


def reverb_mono(x: np.ndarray, sample_rate: int, room_size: float, damping: float, wet_level: float, dry_level: float, width: float,
                freeze_mode: float = 0.0) -> np.ndarray:
    """juce::Reverb::processMono after setParameters + setSampleRate (no parameter ramp: prepare() snaps the smoothed values)."""
    f = np.float32
    wet = f(wet_level) * f(3.0)
    dry = f(dry_level) * f(2.0)
    wet1 = f(0.5) * wet * (f(1.0) + f(width))
    gain = f(0.015)
    damp = f(damping) * f(0.4)
    feedback = f(room_size) * f(0.28) + f(0.7)
    cs, aps = reverb_sizes(sample_rate)
    comb_buf = [np.zeros(n, np.float32) for n in cs]
    comb_idx = [0] * 8
    comb_last = [f(0.0)] * 8
    ap_buf = [np.zeros(n, np.float32) for n in aps]
    ap_idx = [0] * 4
    y = np.empty_like(x, dtype=np.float32)
    one_minus_damp = f(1.0) - damp
    for n in range(len(x)):
        inp = f(x[n]) * gain
        out = f(0.0)
        for j in range(8):
            o = comb_buf[j][comb_idx[j]]
            comb_last[j] = f(o * one_minus_damp + comb_last[j] * damp)
            comb_buf[j][comb_idx[j]] = f(inp + comb_last[j] * feedback)
            comb_idx[j] = (comb_idx[j] + 1) % cs[j]
            out = f(out + o)
        for j in range(4):
            bv = ap_buf[j][ap_idx[j]]
            ap_buf[j][ap_idx[j]] = f(out + bv * f(0.5))
            ap_idx[j] = (ap_idx[j] + 1) % aps[j]
            out = f(bv - out)
        y[n] = f(out * wet1 + f(x[n]) * dry)
    return y


def _ballistics_cte(time_ms: float, sample_rate: int) -> np.float32:
    if time_ms < 1.0e-3:
        return np.float32(0.0)
    return np.float32(math.exp((-2.0 * math.pi * 1000.0 / sample_rate) / time_ms))


def _compress(x: np.ndarray, sample_rate: int, threshold_db: float, ratio: float, attack_ms: float, release_ms: float) -> np.ndarray:
    """juce::dsp::Compressor::processSample with a peak BallisticsFilter."""
    f = np.float32
    thr = f(10.0 ** (threshold_db / 20.0)) if threshold_db > -200.0 else f(0.0)
    thr_inv = f(1.0) / thr
    expo = f(1.0 / ratio) - f(1.0)
    c_at, c_rl = _ballistics_cte(attack_ms, sample_rate), _ballistics_cte(release_ms, sample_rate)
    yold = f(0.0)
    y = np.empty_like(x, dtype=np.float32)
    for n in range(len(x)):
        v = f(x[n])
        a = f(abs(v))
        cte = c_at if a > yold else c_rl
        yold = f(a + cte * (yold - a))
        g = f(1.0) if yold < thr else f(math.pow(float(yold * thr_inv), float(expo)))
        y[n] = f(g * v)
    return y


def compressor(x, sample_rate, threshold_db, ratio, attack_ms, release_ms):
    return _compress(np.asarray(x, np.float32), sample_rate, threshold_db, ratio, attack_ms, release_ms)


def limiter(x, sample_rate, threshold_db, release_ms: float = 100.0):
    """juce::dsp::Limiter: compressor(-10 dB, 4:1, 2 ms, 200 ms) -> compressor(threshold, 1000:1, 0.001 ms, release) ->
    make-up gain 10^(10 (1 - 1/4) / 40) * 10^(-threshold / 20) -> hard clip to [-1, 1]."""
    y = _compress(np.asarray(x, np.float32), sample_rate, -10.0, 4.0, 2.0, 200.0)
    y = _compress(y, sample_rate, threshold_db, 1000.0, 0.001, release_ms)
    gain = np.float32(math.pow(10.0, 10.0 * (1.0 - 0.25) / 40.0) * math.pow(10.0, -threshold_db / 20.0))
    return np.clip(y * gain, np.float32(-1.0), np.float32(1.0)).astype(np.float32)


def apply_board(x: np.ndarray, sample_rate: int, board) -> np.ndarray:
    y = np.asarray(x, np.float32)
    for name, kw in board:
        if name == "Reverb":
            y = reverb_mono(y, sample_rate, **kw)
        elif name == "Compressor":
            y = compressor(y, sample_rate, **kw)
        elif name == "Limiter":
            y = limiter(y, sample_rate, **kw)
        else:
            raise ValueError(name)
    return y
