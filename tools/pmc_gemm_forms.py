#!/usr/bin/env python3
"""Launch one bare NT product a few times on the persistent NT kernel named by ADT_GEMM_NT (256 = half-tile phases, 2wg / ring = the round-6
ring kernel) -- run under `rocprofv3 --pmc ...` (tools/run_pmc_gemm_forms.sh) to compare the kernels' counters on one shape.
env: ADT_PMC_SHAPE=M,N,K (default 63104,768,3072), ADT_PMC_LAUNCHES (default 12)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("ADT_GEMM_TILE", "256")
import torch  # noqa: E402
from adt_str_amd import kernels as K  # noqa: E402

dev = "cuda:0"
M, N, Kd = (int(x) for x in os.environ.get("ADT_PMC_SHAPE", "63104,768,3072").split(","))
a = torch.randn((M, Kd), device=dev).bfloat16()
w = torch.randn((N, Kd), device=dev).bfloat16()
o = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
for _ in range(int(os.environ.get("ADT_PMC_LAUNCHES", "12"))):
    K.gemm(a, w, out=o)
torch.cuda.synchronize()
