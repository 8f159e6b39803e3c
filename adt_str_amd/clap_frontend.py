"""CLAP feature extraction on the GPU: what ``ClapProcessor(audio=list_of_clips, sampling_rate=48000)`` returns
for clips of at most 10 s (reference ``modules/clap_encoder.py:22-23``), computed by ``adt_clap_logmel_db_f32``.
"""
from __future__ import annotations

from typing import List, Sequence

import numpy as np
import torch

from . import _ffi
from .frontend import MelBands

SAMPLE_RATE, N_FFT, HOP, N_MELS, MAX_SAMPLES = 48000, 1024, 480, 64, 480000
N_FRAMES = 1 + MAX_SAMPLES // HOP            # 1001


def htk_mel_filterbank(n_freqs: int = N_FFT // 2 + 1, n_mels: int = N_MELS, f_min: float = 0.0, f_max: float = 14000.0,
                       sample_rate: int = SAMPLE_RATE) -> np.ndarray:
    """``transformers.audio_utils.mel_filter_bank(..., norm=None, mel_scale="htk")`` (float64 triangles), the
    ``mel_filters`` of ClapFeatureExtractor (feature_extraction_clap.py:88-101)."""
    hz2mel = lambda f: 2595.0 * np.log10(1.0 + f / 700.0)
    mel2hz = lambda m: 700.0 * (10.0 ** (m / 2595.0) - 1.0)
    freqs = np.linspace(0, sample_rate // 2, n_freqs)
    edges = mel2hz(np.linspace(hz2mel(f_min), hz2mel(f_max), n_mels + 2))
    width = np.diff(edges)
    slopes = edges[None, :] - freqs[:, None]
    down = -slopes[:, :-2] / width[:-1]
    up = slopes[:, 2:] / width[1:]
    return np.maximum(0.0, np.minimum(down, up))


class ClapLogMel:
    """``ClapLogMel(device)(clips) -> (input_features [B, 4, 1001, 64] fp32 on the GPU, is_longer [B, 1] bool)``.

    ``clips``: list of 1-D float tensors / arrays at 48 kHz, each at most 10 s.  ``is_longer`` is all False; the
    HF extractor would flip one random entry to True (feature_extraction_clap.py:347-350) -- callers that want
    that behaviour pass their own ``is_longer`` to the encoder."""

    def __init__(self, device="cuda"):
        self.device = torch.device(device)
        fb = htk_mel_filterbank().astype(np.float32)
        bands = MelBands.from_dense(fb)
        self.meta = torch.from_numpy(bands.meta).to(self.device)
        self.weights = torch.from_numpy(bands.weights).to(self.device)
        self.nnz = int(bands.weights.size)
        n = np.arange(N_FFT)
        self.window = torch.from_numpy((0.5 - 0.5 * np.cos(2.0 * np.pi * n / N_FFT)).astype(np.float32)).to(self.device)

    def mel(self, clips: Sequence) -> torch.Tensor:
        """[B, 1001, 64] fp32."""
        arrs = [torch.as_tensor(c, dtype=torch.float32).reshape(-1) for c in clips]
        for a in arrs:
            if a.numel() == 0 or a.numel() > MAX_SAMPLES:
                raise NotImplementedError("clips must have 1..480000 samples (longer clips use the fusion crop path, not built)")
        offs = np.zeros(len(arrs) + 1, np.int64)
        offs[1:] = np.cumsum([a.numel() for a in arrs])
        flat = torch.cat(arrs).to(self.device) if arrs else torch.zeros(1, device=self.device)
        off_d = torch.from_numpy(offs).to(self.device)
        out = torch.empty((len(arrs), N_FRAMES, N_MELS), dtype=torch.float32, device=self.device)
        if arrs:
            _ffi.call("adt_clap_logmel_db_f32", _ffi.dptr(flat), _ffi.dptr(off_d), len(arrs), MAX_SAMPLES, N_FFT, HOP, N_FRAMES,
                      _ffi.dptr(self.window), _ffi.dptr(self.meta), _ffi.dptr(self.weights), N_MELS, self.nnz, 1e-10, _ffi.dptr(out),
                      _ffi.current_stream())
        return out

    def __call__(self, clips: Sequence):
        mel = self.mel(clips)
        feats = mel.unsqueeze(1).expand(-1, 4, -1, -1)           # the 4 fusion channels are the same mel for short clips
        return feats, torch.zeros((mel.shape[0], 1), dtype=torch.bool, device=self.device)
