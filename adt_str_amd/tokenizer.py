"""Note <-> token codec: drop-in for the reference's ``MidiTokenizer``
(``modules/midi_tokenizer.py:19-103``).  Pure host integer code.

Vocabulary (``midi_tokenizer.py:25-34``; configs/train/setting-1.yaml:42-48):
0 silence, 1 PAD, 2 BOS, 3 EOS, 4..299 onset in 10 ms steps (``int(onset*100)+4``),
300+pitch, 400+velocity.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np
import torch

from .mapping import ADTOF_MAPPING, GM_TO_CUSTOM

TIME_OFFSET, PITCH_OFFSET, VELOCITY_OFFSET = 4, 300, 400


@dataclass
class MidiTokenizerConfig:
    ADTOF_mapping: bool
    BOS_token: int
    EOS_token: int
    pad_token: int
    silence_token: int
    add_velocity: bool


class MidiTokenizer:
    def __init__(self, config: MidiTokenizerConfig):
        self.ADTOF_mapping = config.ADTOF_mapping
        self.ADTOF_map = ADTOF_MAPPING
        self.GM_standard_midi_to_Gm_custom_map = GM_TO_CUSTOM
        self.adt_tokens_offset_dict = {"time": TIME_OFFSET, "pitch": PITCH_OFFSET, "velocity": VELOCITY_OFFSET}
        self.BOS_token, self.EOS_token = config.BOS_token, config.EOS_token
        self.pad_token, self.silence_token = config.pad_token, config.silence_token
        self.add_velocity = config.add_velocity
        lut = np.full(128, -1, np.int64)                    # GM key -> custom pitch (-> ADTOF class) as one table
        for k, v in GM_TO_CUSTOM.items():
            lut[k] = ADTOF_MAPPING[v] if self.ADTOF_mapping else v
        self._key_lut = lut

    def map_notes_to_Gm_custom(self, notes: torch.Tensor, random_velocity: bool = False) -> torch.Tensor:
        """GM key -> custom pitch (-> ADTOF class); optional ``randint(10, 127)`` velocities (midi_tokenizer.py:36-47).
        Modifies and returns ``notes`` like the reference."""
        if notes.shape[0]:
            src = notes[:, 2].numpy().astype(np.int64)      # int(k): truncation
            if src.min() < 0 or src.max() > 127 or (self._key_lut[src] < 0).any():
                raise KeyError(int(src[(src < 0) | (src > 127) | (self._key_lut[src.clip(0, 127)] < 0)][0]))
            notes[:, 2] = torch.from_numpy(self._key_lut[src])
        else:
            notes[:, 2] = torch.tensor([])
        if random_velocity:
            notes[:, 3] = torch.randint(10, 127, (notes.shape[0],))
        return notes

    def notes_to_adt_tokens(self, notes, **kwargs) -> torch.Tensor:
        """[BOS, (time, pitch[, velocity])..., EOS] (midi_tokenizer.py:49-64)."""
        a = notes.detach().numpy() if isinstance(notes, torch.Tensor) else np.asarray(notes, dtype=np.float32)
        a = a.reshape(-1, 4)
        n = a.shape[0]
        if n == 0:
            return torch.tensor([self.BOS_token, self.EOS_token])
        # int(onset * 100) + 4 on the fp32 element, as the reference's loop over tensor rows computes it
        t = (a[:, 0].astype(np.float32) * np.float32(100)).astype(np.int64) + TIME_OFFSET
        assert bool((t < PITCH_OFFSET).all()), "Time token is out of range"
        per = 3 if self.add_velocity else 2
        # the reference's list mixes Python ints with 0-dim float tensors (pitch + 300), so torch.tensor() makes it float32
        out = np.empty(per * n + 2, np.float32)
        out[0], out[-1] = self.BOS_token, self.EOS_token
        out[1:-1:per] = t
        out[2:-1:per] = a[:, 2] + np.float32(PITCH_OFFSET)
        if self.add_velocity:
            out[3:-1:per] = a[:, 3] + np.float32(VELOCITY_OFFSET)
        return torch.from_numpy(out)

    def empty_adt_tokens(self) -> torch.Tensor:
        return torch.tensor([self.BOS_token, self.silence_token, self.EOS_token])

    def decode(self, tokens) -> torch.Tensor:
        """Inverse with positional pairing (midi_tokenizer.py:69-100): a pitch token belongs to the time token
        right before it, a velocity token to the time token two positions back; offset = onset + 0.1."""
        onsets, pitches, velocities = {}, {}, {}
        for i, tok in enumerate(tokens):
            if tok in (self.BOS_token, self.EOS_token):
                continue
            if TIME_OFFSET <= tok < PITCH_OFFSET:
                onsets[i] = (tok - TIME_OFFSET) / 100
            elif PITCH_OFFSET <= tok < VELOCITY_OFFSET:
                p = tok - PITCH_OFFSET
                if self.ADTOF_mapping:
                    p = self.ADTOF_map[p]
                if i - 1 in onsets:
                    pitches[i - 1] = p
            elif tok >= VELOCITY_OFFSET:
                if i - 2 in onsets:
                    velocities[i - 2] = tok - VELOCITY_OFFSET
        if not velocities:
            velocities = {i: 100 for i in range(len(onsets))}
        notes = [[o, o + 0.1, p, v] for o, p, v in zip(onsets.values(), pitches.values(), velocities.values())]
        return torch.tensor(notes)

    def batch_decode(self, tokens):
        return [self.decode(t) for t in tokens]
