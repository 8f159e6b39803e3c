"""Drop-in import path of the hot-path helpers of the reference's ``utils/audio_utils.py`` (:10-24)."""
from typing import Union

import torch

from adt_str_amd.audio_io import read_wav
from adt_str_amd.resample import Resample


def resample(wav_seg: torch.Tensor, orig_sr: int, target_sr: int) -> torch.Tensor:
    """``T.Resample(orig_freq=orig_sr, new_freq=target_sr)(wav_seg)`` on the GPU (K13); the result stays on the input's device."""
    dev = wav_seg.device
    return Resample(orig_sr, target_sr)(wav_seg.cuda()).to(dev)


def load_and_resample(wav_file: str, target_sr: Union[int, None]) -> torch.Tensor:
    audio, orig_sr = read_wav(wav_file)
    wav_seg = torch.from_numpy(audio).mean(0)
    return wav_seg if target_sr is None else resample(wav_seg, orig_sr, target_sr)


def normalize(wav_seg: torch.Tensor) -> torch.Tensor:
    return wav_seg / wav_seg.abs().max()
