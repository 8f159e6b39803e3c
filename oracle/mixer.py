"""Oracle (test infrastructure): CPU restatement of the one-shot mixer.

Follows ``SynthDrum.__call__`` (reference ``modules/synthetiser.py:255-292``),
``drum_rendering`` (``:214-239``), ``_vel_to_vol`` (``:204-212``) and
``VolumeMixer.init_tracks / instrument_mixer / _normalize_audio``
(``:142-156``) in the same fp32 torch arithmetic, but with every random draw
passed in explicitly (the reference draws them from Python's ``random``:
timbre choice ``:192-202`` once per pitch and clip, mixup ``:217`` per note).
Pinned by tests/golden/mixer.npz (reference outputs + recorded draws).
"""
from __future__ import annotations

from typing import Optional, Callable, List, Sequence

import torch

# per-class mix volume (synthetiser.py:104-113) keyed by ADTOF class pitch
CLASS_VOLUME = {35: 1.0, 38: 1.0, 41: 1.0, 42: 0.7, 48: 0.7, 52: 0.7, 58: 0.7, 61: 1.0}
# custom-GM pitch -> ADTOF class pitch (utils/mapping_utils.py:57-85)
ADTOF_MAP = {35: 35, 36: 35, 37: 38, 38: 38, 39: 38, 40: 38, 41: 41, 42: 42, 43: 42, 44: 42, 45: 41, 46: 48, 47: 41,
             48: 48, 49: 48, 50: 42, 51: 48, 52: 52, 53: 61, 54: 61, 55: 61, 56: 61, 57: 61, 58: 58, 59: 61, 60: 61,
             61: 61}


def vel_to_vol(velocity, min_volume=0.1, max_volume=1.0, base=6):
    """synthetiser.py:204-212 (fp32 tensor arithmetic; velocity 0 -> 0)."""
    if velocity == 0:
        return 0
    v = torch.clamp(torch.as_tensor(velocity, dtype=torch.float32), 0, 127)
    nv = v / 127.0
    return min_volume + (max_volume - min_volume) * (base ** nv - 1) / (base - 1)


def clip_length(notes: torch.Tensor, input_sec: float, sample_rate: int) -> int:
    """synthetiser.py:262-263,243: ``int(max(max_offset + 0.1, input_sec) * sr)``."""
    end = max(notes[:, 1].max() + 0.1, input_sec)
    return int(end * sample_rate)


def render(notes: Sequence[Sequence[float]], input_sec: float, sample_rate: int, adtof_mapping: bool,
           timbres: Callable[[int], tuple], mixups: List[float], fx: Optional[Callable] = None) -> torch.Tensor:
    """``timbres(pitch) -> (main, sub)`` float32 arrays for a pitch (asked once
    per pitch, in order of first appearance); ``mixups[i]`` is note i's draw; ``fx(wav) -> wav`` stands for
    ``VolumeMixer._add_fx`` on the un-normalised mix (synthetiser.py:154-155)."""
    if len(notes) == 0:
        return torch.zeros(int(input_sec * sample_rate))
    notes = torch.tensor(notes)
    W = clip_length(notes, input_sec, sample_rate)
    tracks = {}
    for n in notes:
        if 35 <= n[2].item() <= 61 and n[1].item() >= n[0].item():
            tracks.setdefault(n[2].item(), torch.zeros(W))
    chosen = {}
    max_vel = 0
    for i, n in enumerate(notes):
        onset, offset, pitch, vel = n
        max_vel = max(max_vel, vel)
        if not (35 <= pitch.item() <= 61 and offset.item() >= onset.item()):
            raise ValueError(f"Invalid note: {n}")
        p = int(pitch.item())
        if p not in chosen:
            chosen[p] = timbres(p)
        main, sub = (torch.as_tensor(x, dtype=torch.float32) for x in chosen[p])
        m = mixups[i]
        L = max(len(main), len(sub))
        main = torch.nn.functional.pad(main, (0, L - len(main)))
        sub = torch.nn.functional.pad(sub, (0, L - len(sub)))
        o = main * (1 - m) + m * sub
        o = o / o.abs().max()
        o = o * vel_to_vol(vel)
        start = int(onset * sample_rate)
        seg = tracks[p]
        if start + L > W:
            seg[start:] += o[: W - start]
        else:
            seg[start:start + L] += o
    wav = torch.zeros(W)
    for p, tr in tracks.items():
        key = ADTOF_MAP[int(p)] if not adtof_mapping else int(p)
        wav += tr * CLASS_VOLUME[key]
    if fx is not None:
        wav = torch.as_tensor(fx(wav.numpy().copy()), dtype=torch.float32)
    wav = wav / wav.abs().max()
    return wav * vel_to_vol(max_vel)
