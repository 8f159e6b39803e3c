"""Batch assembly: drop-in for the reference's ``collate_fn`` (``data_modules/train_dataset.py:41-56``)
and the per-item logic of ``LakhDataset.__getitem__`` (``:213-229``), re-cut for a GPU-side renderer:
items carry *notes*, and the whole batch is rendered by one ``SynthDrum.render_batch`` call instead of
one Python render per item inside DataLoader workers.
"""
from __future__ import annotations

import random
from typing import List, Sequence

import numpy as np
import torch
from torch.nn.utils.rnn import pad_sequence

PAD_TOKEN = 1


def collate_fn(batch):
    """list of (wav[W_i], tokens[T_i]) -> dict(wavs [B, Wmax] zero padded, tokens [B, Tmax] PAD padded,
    token_lengths [B]) with the reference's rule that lengths equal to the batch maximum are
    decremented by one (train_dataset.py:46-51; the lengths index tokens[:, :-1])."""
    wavs = [item[0] for item in batch]
    lengths = [len(item[1]) for item in batch]
    tokens = [torch.as_tensor(item[1]) for item in batch]
    top = max(lengths) if lengths else 0
    if top > 0:
        lengths = [n - 1 if n == top else n for n in lengths]
    return {"wavs": pad_sequence(wavs, batch_first=True, padding_value=0.0),
            "tokens": pad_sequence(tokens, batch_first=True, padding_value=PAD_TOKEN).long(),
            "token_lengths": torch.tensor(lengths).long()}


def notes_from_bytes(blob: bytes) -> torch.Tensor:
    """Row schema of the Lakh parquet shards: ``notes`` = float32 [N, 4] bytes (data_modules/midi_parser.py:57-63)."""
    return torch.from_numpy(np.frombuffer(blob, dtype=np.float32).copy()).reshape(-1, 4)


class GpuBatcher:
    """Turns note chunks into training batches on the GPU.

    ``item(notes)`` reproduces ``LakhDataset.__getitem__`` (train_dataset.py:213-229) up to the render:
    with probability ``empty_tokens_percentage`` an empty item ([BOS, SIL, EOS], silent clip), else GM->custom
    pitch mapping, optional random velocities, tokenisation.  ``batch(items)`` renders all clips in one
    mixer call and pads tokens like ``collate_fn``."""

    def __init__(self, tokenizer, synthetiser, empty_tokens_percentage: float = 0.05, random_velocity_prob: float = 0.5):
        self.tokenizer, self.synth = tokenizer, synthetiser
        self.empty_p, self.rand_vel_p = empty_tokens_percentage, random_velocity_prob

    def item(self, notes: torch.Tensor):
        if random.random() < self.empty_p:
            return [], self.tokenizer.empty_adt_tokens()
        notes = self.tokenizer.map_notes_to_Gm_custom(notes.clone(), random_velocity=random.random() < self.rand_vel_p)
        return notes.tolist(), self.tokenizer.notes_to_adt_tokens(notes)

    def batch(self, items: Sequence):
        note_lists = [it[0] for it in items]
        wavs, _ = self.synth.render_batch(note_lists)
        out = collate_fn([(torch.empty(0), it[1]) for it in items])
        out["wavs"] = wavs
        return out
