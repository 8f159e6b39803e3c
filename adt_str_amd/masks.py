"""Decoder masks (reference ``utils/utils.py:28-57``): bool, True = masked."""
from __future__ import annotations

import torch


def _causal_mask(seq_len: int, device=None) -> torch.Tensor:
    return torch.triu(torch.ones(seq_len, seq_len, device=device, dtype=torch.bool), diagonal=1)


def _key_padding_mask_from_lengths(lengths, seq_len: int, device=None) -> torch.Tensor:
    if device is None and isinstance(lengths, torch.Tensor):
        device = lengths.device
    lengths_t = lengths.detach().clone() if isinstance(lengths, torch.Tensor) else torch.tensor(lengths, device=device)
    return torch.arange(seq_len, device=device).unsqueeze(0) >= lengths_t.unsqueeze(1)


def create_mask_plain(tgt_seq_len: int, tgt_lengths=None, device=None):
    """-> (causal [T, T], key padding [N, T] or None)."""
    tgt_mask = _causal_mask(tgt_seq_len, device)
    if tgt_lengths is None:
        return tgt_mask, None
    lens = tgt_lengths.detach().clone() if isinstance(tgt_lengths, torch.Tensor) else torch.tensor(tgt_lengths, device=device)
    return tgt_mask, _key_padding_mask_from_lengths(lens, tgt_seq_len, device)


def select_inference_device() -> torch.device:
    """The HIP path needs a GPU; there is no mps / cpu fallback (reference utils/utils.py:10-17 picks cuda first too)."""
    if not torch.cuda.is_available():
        raise RuntimeError("adt_str_amd needs an AMD GPU (no CPU path)")
    return torch.device("cuda")
