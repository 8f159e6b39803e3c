#!/usr/bin/env python3
"""Experiment: the persistent NT kernel's tile order -- column groups of ADT_GEMM_GROUP_N tile columns (gemm.hip: all row panels of a group
before the next group) -- over the shapes the training step and the CLAP tower launch.  The variable is read once per process: run once per
value (tools: `for g in 1 2 3 4 6 8; do ADT_GEMM_GROUP_N=$g python tools/exp_gemm_group.py; done`)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from adt_str_amd import kernels as K

dev = "cuda:0"
SHAPES = [(63104, 3072, 768), (63104, 768, 3072), (63104, 2304, 768), (63104, 768, 768), (63104, 768, 2304), (63104, 6144, 768), (63104, 768, 6144),
          (8192, 2304, 768), (8192, 3072, 768), (8192, 1400, 768), (131072, 384, 1536), (32768, 768, 3072), (32768, 3072, 768), (32768, 2304, 768)]


def timeit(fn, n=30, warm=15):
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
    out = []
    for (M, N, Kd) in SHAPES:
        a = torch.randn((M, Kd), device=dev).bfloat16()
        w = torch.randn((N, Kd), device=dev).bfloat16()
        z = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
        out.append(f"{M}x{N}x{Kd} {timeit(lambda: K.gemm(a, w, out=z)) * 1e3:.0f}")
        del a, w, z
    print(f"group_n={os.environ.get('ADT_GEMM_GROUP_N', 'default')}: " + "; ".join(out) + "  (us, bare NT)", flush=True)


if __name__ == "__main__":
    main()
