"""Oracle (test infrastructure): CPU restatement of the CLAP curation arithmetic of
``data_modules/augment_data_with_CLAP.py`` (reference):

  * class-mean embeddings                         :116-121  (mean of the unit vectors, not re-normalised)
  * cosine similarity of every sample to a class  :139-151  (``F.cosine_similarity(x, mean, dim=1)``)
  * global descending sort of (class, sample, score) triples, stable   :151
  * greedy assignment: the first (best) occurrence of a sample wins    :182-193
  * similarity bin label ``"{upper}-{lower}"``     :162-169

Pinned by tests/golden/curation.npz, produced by executing those very lines of the reference
script on synthetic embeddings (tools/make_golden.py, g8_curation).
"""
from __future__ import annotations

from typing import Dict, List, Sequence, Tuple

import torch


def class_means(embeddings_by_class: Dict[int, List[torch.Tensor]]) -> Tuple[List[int], torch.Tensor]:
    keys = [k for k, v in embeddings_by_class.items() if len(v) > 0]
    return keys, torch.stack([torch.mean(torch.stack(embeddings_by_class[k]), dim=0) for k in keys])


def bin_label(score: float, num_bins: int) -> str:
    size = 100 // num_bins
    pct = int(round((max(min(score, 1.0), -1.0) + 1.0) * 50.0))
    idx = min(pct // size, num_bins - 1)
    return f"{(idx + 1) * size}-{idx * size}"


def assign(sample_embeddings: torch.Tensor, reference_embeddings: torch.Tensor, labels: Sequence[int], num_bins: int = 10):
    """-> list of (label, sample index, bin label, score) in copy order."""
    scores = []
    for emb, lab in zip(reference_embeddings, labels):
        sim = torch.nn.functional.cosine_similarity(sample_embeddings, emb, dim=1).tolist()
        scores.extend((lab, i, float(s)) for i, s in enumerate(sim))
    scores.sort(key=lambda x: x[2], reverse=True)
    seen, out = set(), []
    for lab, i, s in scores:
        if i in seen:
            continue
        seen.add(i)
        out.append((lab, i, bin_label(s, num_bins), s))
    return out
