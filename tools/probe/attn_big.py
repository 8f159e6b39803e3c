#!/usr/bin/env python3
"""The attention half of a stage-1 / stage-2 HTSAT layer (C = 192 / 384) in one launch (htsat_attn_big_kernel) against the three launches it
replaces, at 512 clips: agreement and timings."""
import math
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from adt_str_amd import _ffi
from adt_str_amd.clap_encoder import pack_attn_block_weights, pack_rowblock_weights, rowblock, window_bias_layout

dev = "cuda:0"


def t(fn, n=10, warm=3):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for (C, nh, R) in ((384, 16, 16), (192, 8, 32)):
    B = 512
    M = B * R * R
    g = torch.Generator(device=dev).manual_seed(C)
    x = torch.randn((M, C), device=dev, generator=g) * 1.2
    gamma, beta = 1 + 0.1 * torch.randn(C, device=dev, generator=g), 0.1 * torch.randn(C, device=dev, generator=g)
    wqkv = torch.randn((3 * C, C), device=dev, generator=g) / C ** 0.5
    bqkv = 0.2 * torch.randn(3 * C, device=dev, generator=g)
    wo = torch.randn((C, C), device=dev, generator=g) / C ** 0.5
    bo = 0.1 * torch.randn(C, device=dev, generator=g)
    bias = window_bias_layout(0.5 * torch.randn((nh, 64, 64), device=dev, generator=g))
    scale = 1.0 / math.sqrt(24.0)
    qkv_pk, wo_pk = pack_rowblock_weights(0, wqkv), pack_rowblock_weights(1, wo)
    wpk, qkvb = pack_attn_block_weights(wqkv, bqkv, wo, nh)
    qkv = torch.empty((M, 3 * C), dtype=torch.bfloat16, device=dev)
    ctx = torch.empty((M, C), dtype=torch.bfloat16, device=dev)

    def three(xx):
        rowblock(0, xx, qkv_pk, 3 * C // 32, bqkv, ln=(gamma, beta), out16=qkv)
        _ffi.call("adt_window_attn_fwd", qkv.data_ptr(), qkv.stride(0), ctx.data_ptr(), C, bias.data_ptr(), 1, B, R, C, nh, 0, scale, _ffi.current_stream())
        rowblock(1, xx, wo_pk, C // 32, bo, a16=ctx)

    def one(xx):
        _ffi.call("adt_htsat_attn_block", xx.data_ptr(), B, R, C, nh, 0, gamma.data_ptr(), beta.data_ptr(), 1e-5, wpk.data_ptr(), qkvb.data_ptr(), bo.data_ptr(),
                  bias.data_ptr(), 1, scale, _ffi.current_stream())
    xr, xf = x.clone(), x.clone()
    three(xr); one(xf)
    torch.cuda.synchronize()
    ur, uf = xr - x, xf - x
    print(f"C={C}: update max {float(ur.abs().max()):.3f}; one launch vs three: max|diff| {float((uf - ur).abs().max()):.3e} ({float((uf - ur).abs().max() / ur.abs().max()):.2e} of max), mean {float((uf - ur).abs().mean() / ur.abs().mean()):.2e} of mean", flush=True)
    xa, xb = x.clone(), x.clone()
    print(f"C={C}: three launches {t(lambda: three(xa)):.1f} us, one launch {t(lambda: one(xb)):.1f} us; again {t(lambda: three(xa)):.1f} / {t(lambda: one(xb)):.1f}", flush=True)
    del x, xr, xf, xa, xb, qkv, ctx
    torch.cuda.empty_cache()
