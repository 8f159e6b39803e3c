#!/usr/bin/env python3
"""Launch the encoder self-attention forward + backward (B=64, H=6, S=986, dropout 0.1) a few times; run under rocprofv3 --pmc."""
import math
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from adt_str_amd import kernels as K

dev = "cuda:0"
B, H, S = 64, 6, 986
d = H * 128
qkv = torch.randn((B * S, 3 * d), device=dev).bfloat16()
scale = 1 / math.sqrt(128)
drop = (0.1, 12345) if os.environ.get("ADT_ATTN_DROP", "1") == "1" else None
o, lse = K.attn_fwd(qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:], B, H, S, S, scale, drop=drop, save_bits=True)     # the product's path: keep bits -> one-kernel backward
do = torch.randn((B * S, d), device=dev).bfloat16()
dqkv = torch.empty_like(qkv)
# the backward path under the counters: as in the product (one kernel; with dropout fed by the forward's keep bits), or forced by
# ADT_ATTN_BWD=split / ADT_ATTN_NO_BITS=1
for _ in range(4):
    K.attn_fwd(qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:], B, H, S, S, scale, drop=drop, save_bits=True)
    K.attn_bwd(qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:], o, do, lse, dqkv[:, :d], dqkv[:, d:2 * d], dqkv[:, 2 * d:], B, H, S, S, scale, drop=drop)
torch.cuda.synchronize()
