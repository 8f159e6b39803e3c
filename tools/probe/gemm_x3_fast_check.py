#!/usr/bin/env python3
"""The split-bf16 product on the persistent bf16 kernels (adt_gemm_bf16x3, [hi | lo] planes) against an fp64 product and against the tiled
split kernel (gemm_f32x3_kernel, ADT_X3_TILED=1): the three layouts the engine uses, the FFN epilogues with fp32 side arrays, timings."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from adt_str_amd import kernels as K

dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
K.set_f32_products("bf16x3")


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


def both(fn):
    os.environ.pop("ADT_X3_TILED", None)
    fast = fn()
    os.environ["ADT_X3_TILED"] = "1"
    tiled = fn()
    os.environ.pop("ADT_X3_TILED", None)
    return fast, tiled


bad = 0
for (M, N, Kd) in ((63104, 768, 768), (8192, 2304, 768), (32768, 3072, 768), (16384, 768, 3072)):
    a = torch.randn((M, Kd), device=dev, generator=g)
    b = torch.randn((N, Kd), device=dev, generator=g) * 0.05
    ref = (a.double() @ b.double().T)
    scale = float(ref.abs().max())
    sup = K._ffi.load().adt_gemm_bf16x3_supported(0, M, N, Kd), K._ffi.load().adt_gemm_bf16x3_supported(1, Kd, N, M)
    for name, kw, A, B in (("NT", {}, a, b), ("NN (b_kn)", dict(b_kn=True), a, b.T.contiguous())):
        fast, tiled = both(lambda: K.gemm(A, B, **kw))
        ef, et = float((fast.double() - ref).abs().max()) / scale, float((tiled.double() - ref).abs().max()) / scale
        print(f"{name:10s} M={M} N={N} K={Kd} supported={sup[0]}: max|err|/max|ref| fast {ef:.2e} tiled {et:.2e}  fast-vs-tiled {float((fast - tiled).abs().max()) / scale:.2e}", flush=True)
        bad += ef > 3e-5
    # weight gradient: C[Kd, N] = a^T b2, a [M, Kd], b2 [M, N]
    b2 = torch.randn((M, N), device=dev, generator=g) * 0.05
    refw = a.double().T @ b2.double()
    sw = float(refw.abs().max())
    fast, tiled = both(lambda: K.gemm(a, b2, trans=True))
    ef, et = float((fast.double() - refw).abs().max()) / sw, float((tiled.double() - refw).abs().max()) / sw
    print(f"TN         rows={M} M={Kd} N={N} supported={sup[1]}: max|err|/max|ref| fast {ef:.2e} tiled {et:.2e}", flush=True)
    bad += ef > 3e-5
    del a, b, b2, ref, refw, fast, tiled
    torch.cuda.empty_cache()

# epilogues: FFN-1 (bias + GELU + dropout + saved fp32 factor), its data gradient (x saved factor), residual forms
M, N, Kd = 16384, 3072, 768
a = torch.randn((M, Kd), device=dev, generator=g)
w1 = torch.randn((N, Kd), device=dev, generator=g) * 0.05
b1 = torch.randn(N, device=dev, generator=g)
site = K.drop_site(0.1, 3, 7)
def ffn1():
    u = torch.empty((M, N), device=dev)
    h = K.gemm(a, w1, bias=b1, act=1, act_grad_out=u, drop=site)
    return h, u
(hf, uf), (ht, ut) = both(ffn1)
print(f"FFN-1 form: h fast-vs-tiled {float((hf - ht).abs().max()):.2e} (max |h| {float(ht.abs().max()):.2f}), factor {float((uf - ut).abs().max()):.2e}; zero pattern differs at {int((((hf == 0) != (ht == 0)) & (ht.abs() > 1e-6)).sum())} elements above 1e-6", flush=True)
bad += float((hf - ht).abs().max()) > 1e-4 * float(ht.abs().max()) or float((uf - ut).abs().max()) > 1e-4
dy = torch.randn((M, Kd), device=dev, generator=g)
res = torch.randn((M, N), device=dev, generator=g)
def dgrad():
    return K.gemm(dy, w1, act_grad=ut)
gf, gt = both(dgrad)
print(f"dgrad x factor: fast-vs-tiled {float((gf - gt).abs().max()):.2e} (max {float(gt.abs().max()):.2f})", flush=True)
bad += float((gf - gt).abs().max()) > 1e-4 * float(gt.abs().max())
def resid():
    return K.gemm(a, w1, bias=b1, residual=res, drop=site)
rf, rt = both(resid)
print(f"bias + dropout + residual: fast-vs-tiled {float((rf - rt).abs().max()):.2e}", flush=True)
bad += float((rf - rt).abs().max()) > 1e-4 * float(rt.abs().max())
print("X3 FAST CHECK:", "ok" if not bad else f"{bad} FAILED", flush=True)

# timings at the encoder shapes
M = 63104
for (N, Kd) in ((3072, 768), (768, 3072), (2304, 768), (768, 768)):
    a = torch.randn((M, Kd), device=dev, generator=g)
    b = torch.randn((N, Kd), device=dev, generator=g) * 0.05
    o = torch.empty((M, N), device=dev)
    K._x3_activation_planes(a)
    tf = t(lambda: K.gemm(a, b, out=o))
    K.x3_register_weights([b])
    tfw = t(lambda: K.gemm(a, b, out=o))
    K._x3_weights.clear()
    os.environ["ADT_X3_TILED"] = "1"
    tt = t(lambda: K.gemm(a, b, out=o))
    os.environ.pop("ADT_X3_TILED")
    ts = t(lambda: K.split_planes(a))
    fl = 2.0 * M * N * Kd
    print(f"NT {M}x{N}x{Kd}: fast {tf:.3f} ms, with registered weight planes {tfw:.3f} ms ({fl / tfw / 1e9:.0f} TFLOP/s of fp32-equivalent work), tiled {tt:.3f} ms; split of A alone {ts:.3f} ms", flush=True)
    b2 = torch.randn((M, N), device=dev, generator=g)
    wout = torch.empty((Kd, N), device=dev)
    K._x3_activation_planes(b2)
    tf = t(lambda: K.gemm(a, b2, trans=True, out=wout))
    os.environ["ADT_X3_TILED"] = "1"
    tt = t(lambda: K.gemm(a, b2, trans=True, out=wout))
    os.environ.pop("ADT_X3_TILED")
    print(f"TN rows={M} {Kd}x{N}: fast {tf:.3f} ms ({fl / tf / 1e9:.0f} TFLOP/s), tiled {tt:.3f} ms", flush=True)
    del a, b, o, b2, wout
    torch.cuda.empty_cache()
