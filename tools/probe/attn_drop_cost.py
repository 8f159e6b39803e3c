"""Encoder-shape self-attention (B = 64, H = 6, S = 986) forward and backward pair, with and without dropout, back to back."""
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from adt_str_amd import kernels as K

dev = "cuda:0"
B, H, S = 64, 6, 986
d = H * 128
qkv = torch.randn((B * S, 3 * d), device=dev).bfloat16()
do = torch.randn((B * S, d), device=dev).bfloat16()
dqkv = torch.empty_like(qkv)
scale = 1 / math.sqrt(128)


def timed(fn, n=40):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for drop in (None, (0.1, 12345)):
    o, lse = K.attn_fwd(qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:], B, H, S, S, scale, drop=drop)
    f = timed(lambda: K.attn_fwd(qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:], B, H, S, S, scale, drop=drop))
    bw = timed(lambda: K.attn_bwd(qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:], o, do, lse, dqkv[:, :d], dqkv[:, d:2 * d], dqkv[:, 2 * d:], B, H, S, S,
                                  scale, drop=drop))
    print(f"dropout {drop[0] if drop else 0}: forward {f:.1f} us, backward pair {bw:.1f} us, out checksum {float(o.float().abs().mean()):.6f}")
