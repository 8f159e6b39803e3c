#!/usr/bin/env python3
"""Micro-benchmarks of the hot kernels at the training-step shapes (GPU box only).
Prints one line per kernel/shape: average ms (HIP events, 40 launches after 30 untimed ones) and TFLOP/s or GB/s."""
import math
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from adt_str_amd import kernels as K

dev = "cuda:0"


def timeit(fn, n=40, warm=30):      # the first ~30 launches after an idle gap are a boost -> throttle transient
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    M = 64 * 986
    only = sys.argv[1] if len(sys.argv) > 1 else ""
    print("== NT GEMM (forward / dgrad) ==  ADT_GEMM_TILE=%s" % os.environ.get("ADT_GEMM_TILE", "auto"))
    for (m, n, k) in [] if only == "attn" else [(M, 3072, 768), (M, 768, 3072), (M, 2304, 768), (M, 768, 768), (M, 768, 2304), (M, 1536, 768), (8192, 1400, 768),
                      (8192, 768, 1400), (8192, 3072, 768), (4096, 4096, 4096), (8192, 8192, 8192)]:
        a = torch.randn((m, k), device=dev).bfloat16()
        b = torch.randn((n, k), device=dev).bfloat16()
        ms = timeit(lambda: K.gemm(a, b))
        print(f"NT M={m} N={n} K={k}: {ms:.3f} ms  {2.0*m*n*k/ms/1e9:.1f} TFLOP/s")
    bias = torch.zeros(3072, device=dev)
    a = torch.randn((M, 768), device=dev).bfloat16()
    b = torch.randn((3072, 768), device=dev).bfloat16()
    u = torch.empty((M, 3072), device=dev, dtype=torch.bfloat16)
    ms = timeit(lambda: K.gemm(a, b, bias=bias, act=1, pre_act_out=u))
    print(f"NT FFN1 + bias + GELU + pre-act: {ms:.3f} ms  {2.0*M*3072*768/ms/1e9:.1f} TFLOP/s")
    res = torch.randn((M, 768), device=dev)
    x16 = torch.empty((M, 768), device=dev, dtype=torch.bfloat16)
    wo = torch.randn((768, 768), device=dev).bfloat16()
    b768 = torch.zeros(768, device=dev)
    ms = timeit(lambda: K.gemm(a, wo, bias=b768, residual=res, out_dtype=torch.float32, aux_bf16_out=x16, drop=(0.1, 3)))
    print(f"NT out-proj + bias + dropout + residual (fp32 + bf16 out): {ms:.3f} ms  {2.0*M*768*768/ms/1e9:.1f} TFLOP/s")
    h = torch.randn((M, 3072), device=dev).bfloat16()
    w2 = torch.randn((768, 3072), device=dev).bfloat16()
    ms = timeit(lambda: K.gemm(h, w2, bias=b768, residual=res, out_dtype=torch.float32, aux_bf16_out=x16, drop=(0.1, 3)))
    print(f"NT FFN2 + bias + dropout + residual (fp32 + bf16 out): {ms:.3f} ms  {2.0*M*768*3072/ms/1e9:.1f} TFLOP/s")
    du = torch.empty((M, 3072), device=dev, dtype=torch.bfloat16)
    ms = timeit(lambda: K.gemm(a, b, gelu_grad_of=u, drop=(0.1, 3), out=du))
    print(f"NT FFN dgrad through GELU + dropout: {ms:.3f} ms  {2.0*M*3072*768/ms/1e9:.1f} TFLOP/s")
    cs = torch.empty(3072, device=dev)
    ms = timeit(lambda: K.gemm(a, b, gelu_grad_of=u, drop=(0.1, 3), out=du, colsum_out=cs))
    print(f"NT FFN dgrad through GELU + dropout + colsum: {ms:.3f} ms  {2.0*M*3072*768/ms/1e9:.1f} TFLOP/s")
    ms = timeit(lambda: K.gemm(a, b, act_grad=u, out=du, colsum_out=cs))
    print(f"NT FFN dgrad x saved factor + colsum: {ms:.3f} ms  {2.0*M*3072*768/ms/1e9:.1f} TFLOP/s")
    ms = timeit(lambda: K.gemm(a, b, bias=bias, act=1, act_grad_out=u, drop=(0.1, 3)))
    print(f"NT FFN1 + bias + GELU + dropout + saved factor: {ms:.3f} ms  {2.0*M*3072*768/ms/1e9:.1f} TFLOP/s")
    ms = timeit(lambda: K.gemm(a, b, bias=bias, act=1, pre_act_out=u, drop=(0.1, 3)))
    print(f"NT FFN1 + bias + GELU + dropout + pre-act: {ms:.3f} ms  {2.0*M*3072*768/ms/1e9:.1f} TFLOP/s")
    ms = timeit(lambda: K.gemm(du, w2, residual=res, out_dtype=torch.float32))
    print(f"NT dx (K=3072) + residual fp32: {ms:.3f} ms  {2.0*M*3072*768/ms/1e9:.1f} TFLOP/s")
    if only == "nt":
        return
    print("== TN GEMM (wgrad) ==")
    for (kk, m, n) in ([] if only == "attn" else [(M, 768, 3072), (M, 3072, 768), (M, 2304, 768), (M, 768, 768), (M, 1536, 768), (8192, 1400, 768), (8192, 768, 3072), (8192, 2304, 768), (8192, 768, 768)]):
        a = torch.randn((kk, m), device=dev).bfloat16()
        b = torch.randn((kk, n), device=dev).bfloat16()
        out = torch.empty((m, n), device=dev)
        ms = timeit(lambda: K.gemm(a, b, trans=True, out=out))
        print(f"TN K={kk} M={m} N={n}: {ms:.3f} ms  {2.0*m*n*kk/ms/1e9:.1f} TFLOP/s")
    print("== attention (encoder self-attention shape) ==")
    B, H, S = 64, 6, 986
    d = H * 128
    qkv = torch.randn((B * S, 3 * d), device=dev).bfloat16()
    scale = 1 / math.sqrt(128)
    o, lse = K.attn_fwd(qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:], B, H, S, S, scale)
    ms = timeit(lambda: K.attn_fwd(qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:], B, H, S, S, scale))
    fl = 4.0 * B * H * S * S * 128
    print(f"attn fwd B={B} H={H} S={S}: {ms:.3f} ms  {fl/ms/1e9:.1f} TFLOP/s")
    do = torch.randn((B * S, d), device=dev).bfloat16()
    dqkv = torch.empty_like(qkv)
    ms = timeit(lambda: K.attn_bwd(qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:], o, do, lse, dqkv[:, :d], dqkv[:, d:2 * d], dqkv[:, 2 * d:],
                                   B, H, S, S, scale))
    print(f"attn bwd (dq + dkdv kernels, 7 matmuls): {ms:.3f} ms  {fl*3.5/ms/1e9:.1f} TFLOP/s executed, {fl*2.5/ms/1e9:.1f} algorithmic")
    for drop in ((0.1, 12345),):
        # what the training step launches: the forward leaves its keep decisions as bits (save_bits), the one-kernel backward reads them
        ms = timeit(lambda: K.attn_fwd(qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:], B, H, S, S, scale, drop=drop, save_bits=True))
        print(f"attn fwd dropout {drop[0]} (storing keep bits, the step's form): {ms:.3f} ms  {fl/ms/1e9:.1f} TFLOP/s")
        od, saved = K.attn_fwd(qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:], B, H, S, S, scale, drop=drop, save_bits=True)
        ms = timeit(lambda: K.attn_bwd(qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:], od, do, saved, dqkv[:, :d], dqkv[:, d:2 * d], dqkv[:, 2 * d:],
                                       B, H, S, S, scale, drop=drop))
        print(f"attn bwd dropout {drop[0]} (keep bits, the step's form): {ms:.3f} ms  {fl*2.5/ms/1e9:.1f} TFLOP/s algorithmic")
        ms = timeit(lambda: K.attn_bwd(qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:], od, do, saved.lse, dqkv[:, :d], dqkv[:, d:2 * d], dqkv[:, 2 * d:],
                                       B, H, S, S, scale, drop=drop))
        print(f"attn bwd dropout {drop[0]} (masks hashed again: the path without bits): {ms:.3f} ms  {fl*2.5/ms/1e9:.1f} TFLOP/s algorithmic")
    T = 128                                   # decoder shapes: cross-attention onto the memory, causal self-attention
    qc = torch.randn((B * T, d), device=dev).bfloat16()
    kvc = torch.randn((B * S, 2 * d), device=dev).bfloat16()
    oc, lsec = K.attn_fwd(qc, kvc[:, :d], kvc[:, d:], B, H, T, S, scale, drop=(0.1, 7), save_bits=True)
    doc = torch.randn((B * T, d), device=dev).bfloat16()
    dqc, dkvc = torch.empty_like(qc), torch.empty_like(kvc)
    ms = timeit(lambda: K.attn_fwd(qc, kvc[:, :d], kvc[:, d:], B, H, T, S, scale, drop=(0.1, 7)))
    print(f"cross-attn fwd T={T} S={S} dropout: {ms:.3f} ms")
    ms = timeit(lambda: K.attn_bwd(qc, kvc[:, :d], kvc[:, d:], oc, doc, lsec, dqc, dkvc[:, :d], dkvc[:, d:], B, H, T, S, scale, drop=(0.1, 7)))
    print(f"cross-attn bwd T={T} S={S} dropout (keep bits): {ms:.3f} ms")
    qs = torch.randn((B * T, 3 * d), device=dev).bfloat16()
    klen = torch.full((B,), T - 9, dtype=torch.int32, device=dev)
    os_, lses = K.attn_fwd(qs[:, :d], qs[:, d:2 * d], qs[:, 2 * d:], B, H, T, T, scale, causal=True, key_len=klen, drop=(0.1, 9))
    dqs = torch.empty_like(qs)
    ms = timeit(lambda: K.attn_bwd(qs[:, :d], qs[:, d:2 * d], qs[:, 2 * d:], os_, doc, lses, dqs[:, :d], dqs[:, d:2 * d], dqs[:, 2 * d:],
                                   B, H, T, T, scale, causal=True, key_len=klen, drop=(0.1, 9)))
    print(f"causal self-attn bwd T={T} dropout: {ms:.3f} ms")
    if only == "attn":
        return
    print("== row kernels ==")
    x = torch.randn((M, 768), device=dev)
    g, b_ = torch.ones(768, device=dev), torch.zeros(768, device=dev)
    ms = timeit(lambda: K.layernorm_fwd(x, g, b_))
    print(f"layernorm fwd [{M},768]: {ms:.3f} ms  {(M*768*(4+4+2))/ms/1e6:.0f} GB/s")
    y32, y16, mean, rstd = K.layernorm_fwd(x, g, b_)
    dg, db, dxs = (torch.empty(768, device=dev) for _ in range(3))
    ms = timeit(lambda: K.layernorm_bwd(x, x, g, mean, rstd, dg, db, dxs))
    print(f"layernorm bwd [{M},768]: {ms:.3f} ms  {(M*768*(4+4+4+2))/ms/1e6:.0f} GB/s")
    xb = torch.randn((M, 3072), device=dev).bfloat16()
    ms = timeit(lambda: K.colsum(xb))
    print(f"colsum [{M},3072] bf16: {ms:.3f} ms  {(M*3072*2)/ms/1e6:.0f} GB/s")


if __name__ == "__main__":
    main()
