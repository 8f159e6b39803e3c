#!/usr/bin/env python3
"""Split-bf16 ("bf16x3") GEMM against the exact f32-MFMA GEMM and an fp64 product: all four operand layouts, error levels, timings."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from adt_str_amd import kernels as K

dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)


def t(fn, n=10, warm=3):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for (M, N, Kd) in ((300, 200, 100), (1000, 768, 768), (4096, 3072, 768)):
    a = torch.randn((M, Kd), device=dev, generator=g)
    b = torch.randn((N, Kd), device=dev, generator=g) * 0.05
    ref = a.double() @ b.double().T
    scale = float(ref.abs().max())
    for name, kw, A, B in (("NT", {}, a, b), ("NN (b_kn)", dict(b_kn=True), a, b.T.contiguous()), ("TN", dict(trans=True), a.T.contiguous(), b.T.contiguous())):
        out = {}
        for mode in ("f32", "bf16x3"):
            K.set_f32_products(mode)
            out[mode] = K.gemm(A, B, **kw)
        e32 = float((out["f32"].double() - ref).abs().max()) / scale
        ex3 = float((out["bf16x3"].double() - ref).abs().max()) / scale
        ebf = float(((a.bfloat16().double() @ b.bfloat16().double().T) - ref).abs().max()) / scale
        print(f"{name:10s} M={M} N={N} K={Kd}: max|err|/max|ref|  f32 {e32:.2e}  bf16x3 {ex3:.2e}  (plain bf16 operands {ebf:.2e})", flush=True)
M, N, Kd = 63104, 3072, 768
a = torch.randn((M, Kd), device=dev, generator=g)
b = torch.randn((N, Kd), device=dev, generator=g) * 0.05
o = torch.empty((M, N), device=dev)
for mode in ("f32", "bf16x3"):
    K.set_f32_products(mode)
    ms = t(lambda: K.gemm(a, b, out=o))
    print(f"NT {M}x{N}x{Kd} {mode}: {ms:.3f} ms = {2.0 * M * N * Kd / ms / 1e9:.1f} TFLOP/s", flush=True)
    at, bt = a.T.contiguous(), torch.randn((M, N), device=dev, generator=g)
    w = torch.empty((Kd, N), device=dev)
    ms = t(lambda: K.gemm(at.T.contiguous() if False else a, bt, trans=True, out=w))
    print(f"TN K={M} M={Kd} N={N} {mode}: {ms:.3f} ms = {2.0 * M * N * Kd / ms / 1e9:.1f} TFLOP/s", flush=True)
    del at, bt, w
K.set_f32_products("f32")

# attention at the encoder shape: exact f32 products vs split-bf16
import math
B, H, S, dh = 64, 6, 986, 128
d = H * dh
qkv = torch.randn((B * S, 3 * d), device=dev, generator=g)
dout = torch.randn((B * S, d), device=dev, generator=g)
for mode in ("f32", "bf16x3"):
    K.set_f32_products(mode)
    o, lse = K.attn_fwd(qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:], B, H, S, S, 1.0 / math.sqrt(dh))
    dqkv = torch.empty_like(qkv)
    tf = t(lambda: K.attn_fwd(qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:], B, H, S, S, 1.0 / math.sqrt(dh), out=o), n=3, warm=1)
    tb = t(lambda: K.attn_bwd(qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:], o, dout, lse, dqkv[:, :d], dqkv[:, d:2 * d], dqkv[:, 2 * d:], B, H, S, S, 1.0 / math.sqrt(dh)), n=3, warm=1)
    fl = 4.0 * B * H * S * S * dh
    print(f"attention {B}x{H}x{S}^2 {mode}: fwd {tf:.2f} ms ({fl / tf / 1e9:.0f} TFLOP/s), bwd {tb:.2f} ms ({2.5 * fl / tb / 1e9:.0f} TFLOP/s algorithmic)", flush=True)
K.set_f32_products("f32")
