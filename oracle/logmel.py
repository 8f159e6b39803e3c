"""Oracle (test infrastructure): CPU restatement of the ADT log-mel front end.

Follows ``ComputeMelSpectrogram`` in the reference, ``model.py:68-97``:
``torchaudio.transforms.MelSpectrogram(sample_rate, n_fft=win_length,
hop_length=int(time_res*sr), n_mels, f_min=20.0, power=2)`` (``model.py:71-78``)
in fp32, then ``log(mel + 1e-10)`` (``:91``), ``clamp(-23, 12)`` (``:92``),
``(x + 23) / 35`` (``:93``), ``permute(0, 2, 1)[:, pad:-(pad+1), :]``
(``:95-97``) with ``pad = int((win_length / 2) // hop + 1)`` (``:79``).

PARITY STATUS: the post-processing lines above are reference-owned and pinned by
``tests/golden/logmel_*.npz``.  The STFT / filterbank conventions are
torchaudio==2.8.0's (``requirements.txt:2``), which is not installed here, so
that part is restated from torchaudio's published definition (periodic Hann,
``center=True`` reflect padding, onesided power spectrum, htk mel scale,
``norm=None``) and is "parity unpinned"; it is cross-checked in the tests
against ``transformers.audio_utils.mel_filter_bank``.
"""
from __future__ import annotations

import math

import numpy as np
import torch


def hop_length(time_res: float, sample_rate: int) -> int:
    """``int(time_res * sample_rate)`` -- float multiply then int (model.py:74)."""
    return int(time_res * sample_rate)


def trim_pad(win_length: int, hop: int) -> int:
    """``int((win_length / 2) // hop + 1)`` (model.py:79)."""
    return int((win_length / 2) // hop + 1)


def n_out_frames(L: int, hop: int, win_length: int) -> int:
    """Frames left after the ``[pad:-(pad+1)]`` trim (model.py:95-97)."""
    n_frames = 1 + L // hop
    pad = trim_pad(win_length, hop)
    return max(n_frames - pad - (pad + 1), 0)


def hann_window(n_fft: int) -> torch.Tensor:
    """Periodic Hann window, fp32 (torchaudio Spectrogram default window_fn)."""
    return torch.hann_window(n_fft, periodic=True, dtype=torch.float32)


def mel_filterbank(sample_rate: int, n_fft: int, n_mels: int, f_min: float = 20.0,
                   f_max: float | None = None) -> torch.Tensor:
    """``torchaudio.functional.melscale_fbanks(n_freqs, f_min, f_max, n_mels,
    sample_rate, norm=None, mel_scale="htk")`` restated in the same fp32 torch
    arithmetic.  Returns ``fb[n_fft//2+1, n_mels]`` fp32."""
    n_freqs = n_fft // 2 + 1
    if f_max is None:
        f_max = float(sample_rate // 2)
    all_freqs = torch.linspace(0, sample_rate // 2, n_freqs)
    m_min = 2595.0 * math.log10(1.0 + (f_min / 700.0))
    m_max = 2595.0 * math.log10(1.0 + (f_max / 700.0))
    m_pts = torch.linspace(m_min, m_max, n_mels + 2)
    f_pts = 700.0 * (10 ** (m_pts / 2595.0) - 1.0)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)
    zero = torch.zeros(1)
    down_slopes = (-1.0 * slopes[:, :-2]) / f_diff[:-1]
    up_slopes = slopes[:, 2:] / f_diff[1:]
    return torch.max(zero, torch.min(down_slopes, up_slopes))


def mel_power(wave: torch.Tensor, n_fft: int, hop: int, window: torch.Tensor,
              fb: torch.Tensor) -> torch.Tensor:
    """Power mel spectrogram ``[B, n_mels, n_frames]`` exactly as torchaudio
    composes it: ``torch.stft`` (center, reflect, onesided) -> ``abs()**2`` ->
    ``(spec^T @ fb)^T``; all fp32."""
    wave = wave.float()
    spec = torch.stft(wave, n_fft=n_fft, hop_length=hop, win_length=n_fft, window=window,
                      center=True, pad_mode="reflect", normalized=False, onesided=True,
                      return_complex=True)
    power = spec.abs().pow(2.0)                      # [B, n_freqs, n_frames]
    mel = torch.matmul(power.transpose(-1, -2), fb).transpose(-1, -2)
    return mel


def logmel(wave: torch.Tensor, sample_rate: int, win_length: int, time_res: float,
           n_mels: int, window: torch.Tensor | None = None,
           fb: torch.Tensor | None = None) -> torch.Tensor:
    """Full front end: ``wave[B, L]`` -> normalised log-mel ``[B, F, n_mels]``
    fp32 (model.py:81-97)."""
    hop = hop_length(time_res, sample_rate)
    if window is None:
        window = hann_window(win_length)
    if fb is None:
        fb = mel_filterbank(sample_rate, win_length, n_mels)
    mel = mel_power(wave, win_length, hop, window, fb)
    x = torch.log(mel + 1e-10)
    x = torch.clamp(x, -23, 12)
    x = (x + 23) / (12 + 23)
    pad = trim_pad(win_length, hop)
    return x.permute(0, 2, 1)[:, pad:-(pad + 1), :].contiguous()


def logmel_f64(wave: np.ndarray, sample_rate: int, win_length: int, time_res: float,
               n_mels: int) -> np.ndarray:
    """Same definition evaluated in float64 numpy (explicit framing + rfft).
    Not the reference's arithmetic: it is the "truth" both fp32 implementations
    are compared against when fp32 rounding dominates (near-silent bands)."""
    hop = hop_length(time_res, sample_rate)
    n_fft = win_length
    x = np.asarray(wave, dtype=np.float64)
    B, L = x.shape
    xp = np.pad(x, ((0, 0), (n_fft // 2, n_fft // 2)), mode="reflect")
    n_frames = 1 + L // hop
    n = np.arange(n_fft)
    win = 0.5 * (1.0 - np.cos(2.0 * np.pi * n / n_fft))
    idx = (np.arange(n_frames) * hop)[:, None] + n[None, :]
    frames = xp[:, idx] * win                         # [B, T, n_fft]
    power = np.abs(np.fft.rfft(frames, axis=-1)) ** 2  # [B, T, n_freqs]
    fb = mel_filterbank(sample_rate, n_fft, n_mels).double().numpy()
    mel = power @ fb                                  # [B, T, n_mels]
    y = (np.clip(np.log(mel + 1e-10), -23.0, 12.0) + 23.0) / 35.0
    pad = trim_pad(win_length, hop)
    return y[:, pad:-(pad + 1), :]
