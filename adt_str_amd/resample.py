"""Polyphase sinc resampler on the GPU: ``Resample(orig_freq, new_freq)(waveform)`` with the semantics of
``torchaudio.transforms.Resample`` at its default arguments, which is how the reference uses it
(utils/audio_utils.py:18-20, inference.py:89-90, data_modules/augment_data_with_CLAP.py:56-59).

The kernel bank is built once per (orig, new) on the host in float64 -- windowed sinc, Hann window, lowpass_filter_width 6,
rolloff 0.99, restated from torchaudio's published ``_get_sinc_resample_kernel`` -- and applied by ``adt_resample_f32`` (K13)."""
from __future__ import annotations

import math

import numpy as np
import torch

from . import _ffi

LOWPASS_FILTER_WIDTH, ROLLOFF = 6, 0.99


def sinc_kernel_bank(orig_freq: int, new_freq: int):
    """-> (bank [new, K] float32, tap_range [new, 2] int32, width, orig, new); orig/new are the rates over their gcd."""
    g = math.gcd(int(orig_freq), int(new_freq))
    orig, new = int(orig_freq) // g, int(new_freq) // g
    base = min(orig, new) * ROLLOFF
    width = int(math.ceil(LOWPASS_FILTER_WIDTH * orig / base))
    idx = np.arange(-width, width + orig, dtype=np.float64)[None, :] / orig
    t = (np.arange(0, -new, -1, dtype=np.float64)[:, None] / new + idx) * base
    t = np.clip(t, -LOWPASS_FILTER_WIDTH, LOWPASS_FILTER_WIDTH)
    window = np.cos(t * math.pi / LOWPASS_FILTER_WIDTH / 2) ** 2
    t = t * math.pi
    with np.errstate(invalid="ignore", divide="ignore"):
        bank = np.where(t == 0, 1.0, np.sin(t) / t) * window * (base / orig)
    bank = bank.astype(np.float32)
    rng = np.zeros((new, 2), np.int32)
    for p in range(new):
        nz = np.nonzero(bank[p])[0]
        rng[p] = (nz[0], nz[-1] + 1) if len(nz) else (0, 0)
    return bank, rng, width, orig, new


class Resample(torch.nn.Module):
    """``Resample(orig_freq=16000, new_freq=16000)``; ``forward(waveform [..., L]) -> [..., ceil(new * L / orig)]`` (GPU tensors)."""

    def __init__(self, orig_freq: int = 16000, new_freq: int = 16000):
        super().__init__()
        self.orig_freq, self.new_freq = int(orig_freq), int(new_freq)
        if self.orig_freq <= 0 or self.new_freq <= 0:
            raise ValueError("Original frequency and desired frequecy should be positive integers")
        if self.orig_freq != self.new_freq:
            bank, rng, self.width, self.orig, self.new = sinc_kernel_bank(self.orig_freq, self.new_freq)
            self.register_buffer("kernel", torch.from_numpy(bank), persistent=False)
            self.register_buffer("tap_range", torch.from_numpy(rng), persistent=False)

    def forward(self, waveform: torch.Tensor) -> torch.Tensor:
        if self.orig_freq == self.new_freq:
            return waveform
        shape = waveform.shape
        x = waveform.reshape(-1, shape[-1]).float().contiguous()
        if self.kernel.device != x.device:
            self.to(x.device)
        n, length = x.shape
        target = int(math.ceil(self.new * length / self.orig))
        out = torch.empty((n, target), dtype=torch.float32, device=x.device)
        for lo in range(0, n, 65535):
            hi = min(n, lo + 65535)
            _ffi.call("adt_resample_f32", _ffi.dptr(x[lo:hi]), hi - lo, length, x.stride(0), _ffi.dptr(self.kernel), _ffi.dptr(self.tap_range),
                      self.kernel.shape[1], self.width, self.orig, self.new, _ffi.dptr(out[lo:hi]), target, out.stride(0), _ffi.current_stream())
        return out.reshape(shape[:-1] + (target,))
