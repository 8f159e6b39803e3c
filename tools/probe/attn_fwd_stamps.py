#!/usr/bin/env python3
"""Cycle stamps of one wave of the pipelined attention forward over two steady-state tiles (experiment build:
make -C adt_str_amd/csrc EXTRA=-DADT_FWD_EXPERIMENT).  Stamp k of a tile: 0 tile start, 1 after phase A(2t), 2 after the barrier,
3 after the DMA issue, 4 after phase B(2t), 5 after the rescale check, 6 after phase A(2t+1), 7 after phase B(2t+1)."""
import math
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from adt_str_amd import kernels as K

dev = "cuda:0"
B, H, S = 64, 6, 986
d = H * 128
q = torch.randn((B * S, d), device=dev).bfloat16()
kv = torch.randn((B * S, 2 * d), device=dev).bfloat16()
for drop in (None, (0.1, 5)):
    for _ in range(5):
        K.attn_fwd(q, kv[:, :d], kv[:, d:], B, H, S, S, 1 / math.sqrt(128), False, None, drop=drop)
    torch.cuda.synchronize()
    print("dropout", drop is not None, flush=True)
    os.environ["ADT_FWD_STAMPS"] = "1"
    for _ in range(3):
        K.attn_fwd(q, kv[:, :d], kv[:, d:], B, H, S, S, 1 / math.sqrt(128), False, None, drop=drop)
    torch.cuda.synchronize()
    os.environ.pop("ADT_FWD_STAMPS")
