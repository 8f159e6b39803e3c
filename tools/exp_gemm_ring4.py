#!/usr/bin/env python3
"""Experiment (round 6; needs a -DADT_GEMM_RING build: make -C adt_str_amd/csrc EXTRA=-DADT_GEMM_RING): the 128 x 128 NT tile with a four-slot LDS ring (gemm_nt_ring4_kernel, three K-steps of 32 in flight) against the
two-stage kernel (gemm_nt_glds_kernel) on the launches that are latency-bound: the decoder's M = 8192 products, the CLAP tower's small
stages, project_to_mel.  ADT_GEMM_ENV_DYNAMIC=1: the library re-reads ADT_GEMM_RING4 on every call, so the kernels alternate in one process.
Both add the same 32-deep MFMA partial sums in the same order: every epilogue form must agree bit for bit."""
import os
import sys
os.environ["ADT_GEMM_ENV_DYNAMIC"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from adt_str_amd import kernels as K

dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)


def use(v):
    os.environ["ADT_GEMM_RING4"] = v


def timeit(fn, n=50, warm=20):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


bad = 0
site = K.drop_site(0.1, 2, 9)
shapes = [(8192, 768, 768), (8192, 768, 3072), (8192, 768, 2304), (8192, 1400, 768), (63104, 768, 128), (8192, 768, 1536), (32768, 384, 1536), (2048, 768, 768),
          (1000, 264, 96), (32768, 768, 768), (131072, 384, 384)]
for (M, N, Kd) in shapes:
    a = torch.randn((M, Kd), device=dev, generator=g).bfloat16()
    w = (torch.randn((N, Kd), device=dev, generator=g) * 0.05).bfloat16()
    bias = torch.randn(N, device=dev, generator=g)
    res = torch.randn((M, N), device=dev, generator=g)
    forms = {
        "bare": lambda: K.gemm(a, w),
        "bias+dropout+residual fp32": lambda: K.gemm(a, w, bias=bias, out_dtype=torch.float32, drop=site, residual=res),
        "bias+GELU": lambda: K.gemm(a, w, bias=bias, act=1),
    }
    line = []
    for name, fn in forms.items():
        use("0"); r0 = fn().clone(); t0 = min(timeit(fn), timeit(fn))
        use("1"); r1 = fn().clone(); t1 = min(timeit(fn), timeit(fn))
        same = torch.equal(r0, r1)
        bad += 0 if same else 1
        line.append(f"{name}: two-stage {t0:.1f} us, ring {t1:.1f} us ({t1 / t0:.2f}) {'same bits' if same else 'DIFFERENT max ' + str(float((r0.float() - r1.float()).abs().max()))}")
    print(f"{M}x{N}x{Kd}: " + "; ".join(line), flush=True)
    del a, w, res
os.environ.pop("ADT_GEMM_RING4", None)
print("RING4 AGREEMENT:", "all forms bit for bit" if bad == 0 else f"{bad} form(s) differ", flush=True)
