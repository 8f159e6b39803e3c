#!/usr/bin/env python3
"""The persistent NT GEMM against the vendor library (torch.matmul -> hipBLASLt / rocBLAS) on the training step's shapes, bare
bf16 products, back to back.  A yardstick for the K loop only: the library has no fused epilogues."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from adt_str_amd import kernels as K
dev = "cuda:0"
def timeit(fn, n=40):
    for _ in range(10): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for M, N, Kd in [(63104, 3072, 768), (63104, 768, 3072), (63104, 2304, 768), (63104, 768, 768), (8192, 3072, 768), (8192, 8192, 8192), (4096, 4096, 4096)]:
    a = torch.randn(M, Kd, device=dev).bfloat16(); w = torch.randn(N, Kd, device=dev).bfloat16()
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    t_mine = timeit(lambda: K.gemm(a, w, out=out))
    t_lib = timeit(lambda: torch.matmul(a, w.t(), out=out))
    fl = 2.0 * M * N * Kd / 1e9
    print(f"NT M={M} N={N} K={Kd}: this repo {t_mine:.3f} ms ({fl / t_mine:.0f} TF/s) | torch.matmul {t_lib:.3f} ms ({fl / t_lib:.0f} TF/s)", flush=True)
for Kd, M, N in [(63104, 768, 3072), (63104, 2304, 768), (63104, 768, 768)]:
    a = torch.randn(Kd, M, device=dev).bfloat16(); b = torch.randn(Kd, N, device=dev).bfloat16()
    out = torch.empty(M, N, device=dev, dtype=torch.float32); out16 = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    t_mine = timeit(lambda: K.gemm(a, b, trans=True, out=out))
    t_lib = timeit(lambda: torch.matmul(a.t(), b, out=out16))
    fl = 2.0 * M * N * Kd / 1e9
    print(f"TN K={Kd} M={M} N={N}: this repo (fp32 out) {t_mine:.3f} ms ({fl / t_mine:.0f} TF/s) | torch.matmul (bf16 out) {t_lib:.3f} ms ({fl / t_lib:.0f} TF/s)", flush=True)
