"""Oracle (test infrastructure): numpy mirror of the counter-based dropout masks of
adt_str_amd/csrc/dropout.h, so the CPU restatement can apply exactly the masks the kernels
generate (the reference's own Philox draws are not reproducible)."""
from __future__ import annotations

import numpy as np
import torch


def mix32(x: np.ndarray) -> np.ndarray:
    x = x.astype(np.uint32)
    x ^= x >> np.uint32(16); x *= np.uint32(0x7FEB352D); x ^= x >> np.uint32(15); x *= np.uint32(0x846CA68B); x ^= x >> np.uint32(16)
    return x


def scale(shape, p: float, key: int) -> torch.Tensor:
    """mask / (1 - p) for a contiguous tensor of ``shape`` whose element index is its flat offset."""
    n = int(np.prod(shape))
    idx = np.arange(n, dtype=np.uint64)
    with np.errstate(over="ignore"):
        inner = mix32((idx >> np.uint64(32)).astype(np.uint32) ^ np.uint32(key))
        h = mix32(idx.astype(np.uint32) ^ inner)
    pp = min(p, 0.999999)
    thr = np.uint32(int(pp * 4294967296.0))
    keep = (h >= thr).astype(np.float32) * np.float32(1.0 / (1.0 - pp))
    return torch.from_numpy(keep.reshape(shape))
