"""CLAP feature extraction on the GPU: what ``ClapProcessor(audio=list_of_clips, sampling_rate=48000)`` returns
(reference ``modules/clap_encoder.py:22-23``; ``truncation="fusion"``, ``padding="repeatpad"``, the class defaults),
computed by ``adt_clap_logmel_db_f32``: clips of at most 10 s are repeat-padded and carry the same mel in all four
fusion channels; longer clips get the mel of the whole clip, three random 1001-frame crops (front / middle / back third,
``np.random.choice`` like the extractor) and a bilinearly shrunk copy, and are flagged ``is_longer``.
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

    ``clips``: list of 1-D float tensors / arrays at 48 kHz of any length."""

    def __init__(self, device="cuda"):
        self.device = torch.device(device)
        fb = htk_mel_filterbank().astype(np.float32)
        bands = MelBands.from_dense(fb)
        self.meta = torch.from_numpy(bands.meta).to(self.device)
        self.weights = torch.from_numpy(bands.weights).to(self.device)
        self.nnz = int(bands.weights.size)
        n = np.arange(N_FFT)
        self.window = torch.from_numpy((0.5 - 0.5 * np.cos(2.0 * np.pi * n / N_FFT)).astype(np.float32)).to(self.device)

    def _up(self, t: torch.Tensor) -> torch.Tensor:
        """Small host tensor -> device through pinned memory: a copy from pageable memory blocks the host until the GPU has worked off
        everything queued before it (the previous batch's tower, in the curation loop)."""
        if torch.device(self.device).type != "cuda":
            return t.to(self.device)
        return t.contiguous().pin_memory().to(self.device, non_blocking=True)

    def mel(self, clips: Sequence) -> torch.Tensor:
        """[B, 1001, 64] fp32."""
        arrs = [torch.as_tensor(c, dtype=torch.float32).reshape(-1) for c in clips]
        for a in arrs:
            if a.numel() == 0 or a.numel() > MAX_SAMPLES:
                raise ValueError("mel() takes clips of 1..480000 samples; features() handles longer ones")
        offs = np.zeros(len(arrs) + 1, np.int64)
        offs[1:] = np.cumsum([a.numel() for a in arrs])
        out = torch.empty((len(arrs), N_FRAMES, N_MELS), dtype=torch.float32, device=self.device)
        if not arrs:
            return out
        dev_index = self.device.index if self.device.index is not None else (torch.cuda.current_device() if self.device.type == "cuda" else -1)
        if all(a.is_cuda and a.device.index == dev_index and a.is_contiguous() for a in arrs):
            # clips already on the device: the kernel reads them where they are, through one table [pointers | offsets] (no concatenation
            # pass: four copy launches of 16 us per 512 clips)
            table = np.concatenate([np.fromiter((a.data_ptr() for a in arrs), np.int64, len(arrs)), offs])
            tab_d = self._up(torch.from_numpy(table))
            _ffi.call("adt_clap_logmel_db_ptrs_f32", _ffi.dptr(tab_d), _ffi.dptr(tab_d) + 8 * len(arrs), len(arrs), MAX_SAMPLES, N_FFT, HOP, N_FRAMES,
                      _ffi.dptr(self.window), _ffi.dptr(self.meta), _ffi.dptr(self.weights), N_MELS, self.nnz, 1e-10, _ffi.dptr(out),
                      _ffi.current_stream())
            self._keep = (arrs, tab_d)               # (the launch is asynchronous: the clips and the table stay referenced until the next call)
            return out
        flat = torch.cat(arrs).to(self.device)
        off_d = self._up(torch.from_numpy(offs))
        _ffi.call("adt_clap_logmel_db_f32", _ffi.dptr(flat), _ffi.dptr(off_d), len(arrs), MAX_SAMPLES, N_FFT, HOP, N_FRAMES,
                  _ffi.dptr(self.window), _ffi.dptr(self.meta), _ffi.dptr(self.weights), N_MELS, self.nnz, 1e-10, _ffi.dptr(out),
                  _ffi.current_stream())
        return out

    def _full_mel(self, clip: torch.Tensor) -> torch.Tensor:
        """[1 + L // 480, 64] fp32 of ONE clip of any length (no padding: target length = the clip's own)."""
        L = clip.numel()
        n_frames = 1 + L // HOP
        x = clip.to(self.device)
        off_d = torch.tensor([0, L], dtype=torch.int64, device=self.device)
        out = torch.empty((n_frames, N_MELS), dtype=torch.float32, device=self.device)
        _ffi.call("adt_clap_logmel_db_f32", _ffi.dptr(x), _ffi.dptr(off_d), 1, L, N_FFT, HOP, n_frames, _ffi.dptr(self.window),
                  _ffi.dptr(self.meta), _ffi.dptr(self.weights), N_MELS, self.nnz, 1e-10, _ffi.dptr(out), _ffi.current_stream())
        return out

    def _fusion_of_long_clip(self, clip: torch.Tensor):
        """``_get_input_mel`` for a clip longer than 10 s (feature_extraction_clap.py, truncation="fusion") -> ([4, 1001, 64], longer)."""
        mel = self._full_mel(clip)
        total = mel.shape[0]
        if total == N_FRAMES:                                     # 480000 < L < 480480: "we just use the whole audio", not longer
            return mel.unsqueeze(0).expand(4, -1, -1), False
        ranges = np.array_split(list(range(0, total - N_FRAMES + 1)), 3)
        if len(ranges[1]) == 0:
            ranges[1] = [0]
        if len(ranges[2]) == 0:
            ranges[2] = [0]
        i_front, i_mid, i_back = (int(np.random.choice(r)) for r in ranges)       # the extractor's draws, in its order
        shrink = torch.empty((N_FRAMES, N_MELS), dtype=torch.float32, device=self.device)       # F.interpolate(bilinear, align_corners=False)
        _ffi.call("adt_bilinear_resize_f32", _ffi.dptr(mel), total, N_MELS, mel.stride(0), _ffi.dptr(shrink), N_FRAMES, N_MELS, N_MELS,
                  _ffi.current_stream())
        return torch.stack([shrink, mel[i_front:i_front + N_FRAMES], mel[i_mid:i_mid + N_FRAMES], mel[i_back:i_back + N_FRAMES]]), True

    def features(self, clips: Sequence):
        """list of 1-D clips (any length >= 1) -> (input_features [B, 4, 1001, 64] fp32, is_longer [B] bool on the CPU)."""
        arrs = [torch.as_tensor(c, dtype=torch.float32).reshape(-1) for c in clips]
        short = [i for i, a in enumerate(arrs) if a.numel() <= MAX_SAMPLES]
        feats = torch.empty((len(arrs), 4, N_FRAMES, N_MELS), dtype=torch.float32, device=self.device)
        longer = torch.zeros(len(arrs), dtype=torch.bool)
        if len(short) == len(arrs) and short:                   # (no index tensor: its upload would wait for the GPU's queue)
            feats.copy_(self.mel(arrs).unsqueeze(1).expand(-1, 4, -1, -1))
        elif short:
            feats[short] = self.mel([arrs[i] for i in short]).unsqueeze(1).expand(-1, 4, -1, -1)
        for i, a in enumerate(arrs):                              # in clip order: the crops consume numpy's global RNG like the extractor
            if a.numel() > MAX_SAMPLES:
                f, lg = self._fusion_of_long_clip(a)
                feats[i] = f
                longer[i] = lg
        return feats, longer

    def __call__(self, clips: Sequence):
        """-> (input_features [B, 4, 1001, 64], is_longer [B, 1] bool on the device): what the extractor computes per clip, without its
        "flag one random clip of an all-short batch" step (``ClapWrapper.get_audio_features`` does that)."""
        feats, longer = self.features(clips)
        return feats, self._up(longer.view(-1, 1))
