"""Oracle (test infrastructure): ``torchaudio.transforms.Resample(orig_freq, new_freq)`` with default arguments, as the
reference calls it (utils/audio_utils.py:18-20, inference.py:89-90, data_modules/augment_data_with_CLAP.py:56-59).

torchaudio==2.8.0 (requirements.txt:2) is not installable in this image, so this restates its published algorithm
(``torchaudio.functional.functional._get_sinc_resample_kernel`` / ``_apply_sinc_resample_kernel``: sinc_interp_hann,
lowpass_filter_width 6, rolloff 0.99, kernel built in float64 and cast to float32, zero padding (width, width + orig),
``conv1d`` with stride orig, cut to ceil(new * L / orig)).  **Parity unpinned**: no reference test or vector pins it."""
from __future__ import annotations

import math

import torch


def sinc_resample_kernel(orig_freq: int, new_freq: int, lowpass_filter_width: int = 6, rolloff: float = 0.99):
    """-> (kernel [new, 1, 2*width + orig] float32, width, orig, new) with orig/new divided by their gcd."""
    g = math.gcd(int(orig_freq), int(new_freq))
    orig, new = int(orig_freq) // g, int(new_freq) // g
    base_freq = min(orig, new) * rolloff
    width = math.ceil(lowpass_filter_width * orig / base_freq)
    idx = torch.arange(-width, width + orig, dtype=torch.float64)[None, None] / orig
    t = torch.arange(0, -new, -1, dtype=torch.float64)[:, None, None] / new + idx
    t = t * base_freq
    t = t.clamp_(-lowpass_filter_width, lowpass_filter_width)
    window = torch.cos(t * math.pi / lowpass_filter_width / 2) ** 2
    t = t * math.pi
    scale = base_freq / orig
    kernels = torch.where(t == 0, torch.tensor(1.0, dtype=torch.float64), t.sin() / t)
    kernels = kernels * window * scale
    return kernels.to(torch.float32), width, orig, new


def resample(waveform: torch.Tensor, orig_freq: int, new_freq: int) -> torch.Tensor:
    """[..., L] -> [..., ceil(new * L / orig)] (fp32, CPU)."""
    if int(orig_freq) == int(new_freq):
        return waveform
    kernel, width, orig, new = sinc_resample_kernel(orig_freq, new_freq)
    shape = waveform.shape
    x = waveform.reshape(-1, shape[-1]).float()
    n, length = x.shape
    x = torch.nn.functional.pad(x, (width, width + orig))
    y = torch.nn.functional.conv1d(x[:, None], kernel, stride=orig)
    y = y.transpose(1, 2).reshape(n, -1)
    target = int(math.ceil(new * length / orig))
    return y[..., :target].reshape(shape[:-1] + (target,))
