"""CLAP-based one-shot curation: the arithmetic of the reference's
``data_modules/augment_data_with_CLAP.py`` (class means :116-121, cosine similarity + global sort
:139-151, similarity bins :162-169, greedy unique assignment :182-193) with the O(N*C) part on the GPU
(``adt_cosine_argmax_f32``).  File copying stays with the caller.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Sequence

import numpy as np
import torch

from . import _ffi


def class_mean_embeddings(embeddings_by_class: Dict[int, List[torch.Tensor]]):
    """Mean of the (unit) embeddings of every non-empty class, not re-normalised (augment...:116-121)."""
    labels = [k for k, v in embeddings_by_class.items() if len(v) > 0]
    if not labels:
        raise RuntimeError("No reference embeddings found. Please check reference_root.")
    return labels, torch.stack([torch.mean(torch.stack(list(embeddings_by_class[k])), dim=0) for k in labels])


def score_to_bin_label(score_value: float, num_bins: int = 10) -> str:
    """cosine [-1, 1] -> percent [0, 100] -> ``"{upper}-{lower}"`` (augment...:162-169)."""
    if num_bins <= 0 or 100 % num_bins != 0:
        raise ValueError("num_bins must be a positive integer that divides 100 evenly")
    size = 100 // num_bins
    pct = int(round((max(min(score_value, 1.0), -1.0) + 1.0) * 50.0))
    idx = min(pct // size, num_bins - 1)
    return f"{(idx + 1) * size}-{idx * size}"


@dataclass
class Assignment:
    order: np.ndarray        # sample indices in the reference's copy order (best score first)
    label: np.ndarray        # class label of each sample, in that order
    bin: List[str]           # similarity bin of each sample, in that order
    score: np.ndarray        # best cosine of each sample, in that order


def rank_assignments(best_class: np.ndarray, best_score: np.ndarray, labels: Sequence[int], num_bins: int = 10) -> Assignment:
    """Host part: the reference sorts all (class, sample, score) triples by score (stable, descending) and keeps
    the first occurrence of each sample; with per-sample best classes known that is a sort of N items whose ties
    fall back to (class position, sample index) -- the insertion order of the reference's list."""
    n = len(best_score)
    order = np.lexsort((np.arange(n), best_class, -best_score.astype(np.float64)))
    lab = np.asarray(labels)[best_class[order]]
    return Assignment(order=order, label=lab, bin=[score_to_bin_label(float(s), num_bins) for s in best_score[order]],
                      score=best_score[order])


def assign(sample_embeddings: torch.Tensor, reference_embeddings: torch.Tensor, labels: Sequence[int], num_bins: int = 10,
           return_scores: bool = False):
    """sample_embeddings [N, D], reference_embeddings [C, D] (GPU, fp32) -> Assignment (+ all scores [N, C])."""
    x = sample_embeddings.float().contiguous()
    r = reference_embeddings.float().contiguous().to(x.device)
    N, D = x.shape
    C = r.shape[0]
    bc = torch.empty(N, dtype=torch.int32, device=x.device)
    bs = torch.empty(N, dtype=torch.float32, device=x.device)
    sc = torch.empty((N, C), dtype=torch.float32, device=x.device) if return_scores else None
    _ffi.call("adt_cosine_argmax_f32", _ffi.dptr(x), x.stride(0), _ffi.dptr(r), N, D, C, 1e-8, _ffi.dptr(bc), _ffi.dptr(bs),
              _ffi.dptr(sc) if sc is not None else None, _ffi.current_stream())
    res = rank_assignments(bc.cpu().numpy().astype(np.int64), bs.cpu().numpy(), labels, num_bins)
    return (res, sc) if return_scores else res


def copy_originals_to_gold(reference_root: str, augmented_root: str, overwrite: bool = True):
    """``copy_originals_to_augmented.py:62-80``: every ``<reference_root>/<label>/`` tree becomes ``<augmented_root>/<label>/gold``;
    an existing ``gold`` is replaced (``overwrite``) or skipped.  Directories are made here, the files go through the library's
    batched copy (contents, mode, times: shutil.copytree's copy2).  Returns ``(copied labels, skipped labels)``."""
    import os
    import shutil
    from .audio_io import copy_files
    srcs, dsts, copied, skipped = [], [], 0, 0
    os.makedirs(augmented_root, exist_ok=True)
    for label in sorted(os.listdir(reference_root)):
        src = os.path.join(reference_root, label)
        if not os.path.isdir(src):
            continue
        dst = os.path.join(augmented_root, label, "gold")
        if os.path.exists(dst):
            if not overwrite:
                print(f"Destination already exists, skipping copy: {dst}. Use --overwrite to replace.")
                skipped += 1
                continue
            shutil.rmtree(dst)
        for root, _dirs, files in os.walk(src):
            out_dir = os.path.join(dst, os.path.relpath(root, src)) if root != src else dst
            os.makedirs(out_dir, exist_ok=True)
            for f in files:
                srcs.append(os.path.join(root, f))
                dsts.append(os.path.join(out_dir, f))
        copied += 1
    status = copy_files(srcs, dsts)
    for j in status.nonzero()[0]:
        print(f"Failed to copy '{srcs[j]}' -> '{dsts[j]}'")
    return copied, skipped
