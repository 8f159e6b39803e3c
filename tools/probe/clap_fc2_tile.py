"""Stage-2 fc2 of the CLAP tower (M = 131072, N = 384, K = 1536, + bias + fp32 residual in place): persistent 256^2 kernel (N = 384 is 1.5
tile columns) against the 128^2 LDS-DMA kernel.  Run twice: default and ADT_GEMM_TILE=128."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from adt_str_amd import kernels as K

dev = "cuda:0"
for M, N, Kd in ((131072, 384, 1536), (32768, 768, 3072), (32768, 3072, 768), (32768, 768, 768), (131072, 384, 384)):
    h = (torch.randn(M, Kd, device=dev) * 0.5).bfloat16()
    w = (torch.randn(N, Kd, device=dev) * 0.02).bfloat16()
    b = torch.randn(N, device=dev)
    x = torch.randn(M, N, device=dev)
    for _ in range(3):
        K.gemm(h, w, bias=b, residual=x, out=x)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        K.gemm(h, w, bias=b, residual=x, out=x)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"tile={os.environ.get('ADT_GEMM_TILE', 'auto')} M={M} N={N} K={Kd}: {ms * 1e3:.1f} us  {2 * M * N * Kd / ms / 1e9:.0f} TFLOP/s")
