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
    """mask / (1 - p) for a tensor of ``shape``: element index = row * L2 + col (rows = leading dimensions flattened, L2 = last
    dimension rounded up to even); two neighbouring elements take the 16-bit halves of one hash (dropout.h)."""
    shape = tuple(int(s) for s in shape)
    L = shape[-1]
    rows = int(np.prod(shape[:-1])) if len(shape) > 1 else 1
    L2 = L + (L & 1)
    idx = (np.arange(rows, dtype=np.uint64)[:, None] * np.uint64(L2) + np.arange(L, dtype=np.uint64)[None, :]).reshape(-1)
    pair = idx >> np.uint64(1)
    with np.errstate(over="ignore"):
        inner = mix32((pair >> np.uint64(32)).astype(np.uint32) ^ np.uint32(key))
        h = mix32(pair.astype(np.uint32) ^ inner)
    half = np.where((idx & np.uint64(1)).astype(bool), h >> np.uint32(16), h & np.uint32(0xFFFF))
    pp = min(p, 0.999999)
    thr = min(65535, max(1, int(pp * 65536.0 + 0.5)))
    keep = (half >= np.uint32(thr)).astype(np.float32) * np.float32(1.0 / (1.0 - pp))
    return torch.from_numpy(keep.reshape(shape))
