"""Log-mel front end: drop-in for the reference's ``ComputeMelSpectrogram``
(``model.py:68-97``) on top of the fused gfx950 kernel ``adt_logmel_f32``.

Constructor, call signature, output layout ``[B, F, n_mels]`` and state-dict
buffer names (``compute_spec.spectrogram.window``, ``compute_spec.mel_scale.fb``
-- torchaudio's module tree) follow the reference.  The two buffers are the
source of truth: the banded filterbank the kernel consumes is re-derived from
``fb`` whenever it changes (e.g. after ``load_state_dict``).
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np
import torch
from torch import nn

from . import _ffi

N_FFT_SUPPORTED = 2048
LOG_EPS, CLAMP_LO, CLAMP_HI = 1e-10, -23.0, 12.0      # model.py:91-93


def melscale_fbanks(sample_rate: int, n_fft: int, n_mels: int, f_min: float, f_max: float | None = None) -> torch.Tensor:
    """htk-scale triangular filterbank, no area normalisation, in fp32 torch
    arithmetic -- what ``torchaudio.transforms.MelSpectrogram(...).mel_scale.fb``
    holds for the arguments the reference passes (model.py:71-78)."""
    n_freqs = n_fft // 2 + 1
    f_max = float(sample_rate // 2) if f_max is None else f_max
    freqs = torch.linspace(0, sample_rate // 2, n_freqs)
    m_lo = 2595.0 * math.log10(1.0 + f_min / 700.0)
    m_hi = 2595.0 * math.log10(1.0 + f_max / 700.0)
    edges = 700.0 * (10 ** (torch.linspace(m_lo, m_hi, n_mels + 2) / 2595.0) - 1.0)
    width = edges[1:] - edges[:-1]
    dist = edges.unsqueeze(0) - freqs.unsqueeze(1)
    falling = -dist[:, :-2] / width[:-1]
    rising = dist[:, 2:] / width[1:]
    return torch.clamp(torch.minimum(falling, rising), min=0.0)


@dataclass
class MelBands:
    """Banded (CSR) form of a filterbank ``fb[n_freqs, n_mels]``: filter ``j``
    is non-zero only on bins ``lo[j] .. lo[j]+cnt[j]-1``."""
    meta: np.ndarray      # int32 [n_mels, 4] = lo, cnt, offset into weights, 0
    weights: np.ndarray   # float32 [nnz]

    @staticmethod
    def from_dense(fb: np.ndarray) -> "MelBands":
        fb = np.asarray(fb, dtype=np.float32)
        n_freqs, n_mels = fb.shape
        meta = np.zeros((n_mels, 4), np.int32)
        chunks, off = [], 0
        for j in range(n_mels):
            nz = np.nonzero(fb[:, j])[0]
            if nz.size == 0:
                continue
            lo, hi = int(nz[0]), int(nz[-1])
            meta[j] = (lo, hi - lo + 1, off, 0)
            chunks.append(fb[lo:hi + 1, j])
            off += hi - lo + 1
        weights = np.concatenate(chunks).astype(np.float32) if chunks else np.zeros(0, np.float32)
        if meta[:, 1].max(initial=0) > 127 or off > 2304:
            raise ValueError("filterbank is not banded enough for the log-mel kernel (band > 127 bins or > 2304 non-zeros)")
        return MelBands(meta=meta, weights=weights)

    def to_dense(self, n_freqs: int) -> np.ndarray:
        fb = np.zeros((n_freqs, self.meta.shape[0]), np.float32)
        for j, (lo, cnt, off, _) in enumerate(self.meta):
            fb[lo:lo + cnt, j] = self.weights[off:off + cnt]
        return fb


class _Spectrogram(nn.Module):
    def __init__(self, n_fft: int):
        super().__init__()
        self.register_buffer("window", torch.hann_window(n_fft, periodic=True, dtype=torch.float32))


class _MelScale(nn.Module):
    def __init__(self, sample_rate: int, n_fft: int, n_mels: int, f_min: float):
        super().__init__()
        self.register_buffer("fb", melscale_fbanks(sample_rate, n_fft, n_mels, f_min))


class _MelSpectrogramState(nn.Module):
    """Holds the two buffers under torchaudio's names so checkpoints interchange."""

    def __init__(self, sample_rate, n_fft, hop_length, n_mels, f_min):
        super().__init__()
        self.n_fft, self.hop_length, self.n_mels = n_fft, hop_length, n_mels
        self.spectrogram = _Spectrogram(n_fft)
        self.mel_scale = _MelScale(sample_rate, n_fft, n_mels, f_min)


def frame_geometry(L: int, hop: int, win_length: int):
    """(first kept frame, number of kept frames) of the ``[pad:-(pad+1)]`` trim (model.py:79,95-97)."""
    pad = int((win_length / 2) // hop + 1)
    n_frames = 1 + L // hop
    return pad, max(n_frames - pad - (pad + 1), 0)


class ComputeMelSpectrogram(nn.Module):
    """``ComputeMelSpectrogram(sample_rate, win_length, time_res, n_mels)(wave[B, L]) -> [B, F, n_mels]``."""

    def __init__(self, sample_rate, win_length, time_res, n_mels):
        super().__init__()
        hop = int(time_res * sample_rate)
        self.compute_spec = _MelSpectrogramState(sample_rate, win_length, hop, n_mels, f_min=20.0)
        self.window_pad_idxs = int((win_length / 2) // hop + 1)
        self._bands_key = None
        self._bands_dev = None

    # banded filterbank on the device, rebuilt when fb's storage/version/device changes
    def _bands(self, device):
        fb = self.compute_spec.mel_scale.fb
        key = (fb.data_ptr(), fb._version, str(device))
        if self._bands_key != key:
            bands = MelBands.from_dense(fb.detach().float().cpu().numpy())
            self._bands_dev = (torch.from_numpy(bands.meta).to(device), torch.from_numpy(bands.weights).to(device),
                               int(bands.weights.size))
            self._bands_key = key
        return self._bands_dev

    def forward(self, wave: torch.Tensor) -> torch.Tensor:
        if wave.dim() != 2:
            raise ValueError("wave must be [batch, samples]")
        cs = self.compute_spec
        if cs.spectrogram.window.device != wave.device:
            self.compute_spec = cs = cs.to(wave.device)          # model.py:82-83
        wave = wave.float()                                       # model.py:88 (always fp32, autocast off)
        if wave.stride(1) != 1:
            wave = wave.contiguous()
        B, L = wave.shape
        frame_lo, n_out = frame_geometry(L, cs.hop_length, cs.n_fft)
        out = torch.empty((B, n_out, cs.n_mels), dtype=torch.float32, device=wave.device)
        if B == 0 or n_out == 0:
            return out
        meta, weights, nnz = self._bands(wave.device)
        _ffi.call("adt_logmel_f32", _ffi.dptr(wave), B, L, wave.stride(0), cs.n_fft, cs.hop_length, frame_lo, n_out,
                  _ffi.dptr(cs.spectrogram.window), _ffi.dptr(meta), _ffi.dptr(weights), cs.n_mels, nnz,
                  LOG_EPS, CLAMP_LO, CLAMP_HI, _ffi.dptr(out), _ffi.current_stream())
        return out
