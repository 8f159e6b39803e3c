#!/usr/bin/env python3
"""Experiment: the NT forms of an encoder layer on the 128^2 kernel (two workgroups per CU, whose epilogues and K loops overlap by
themselves) against the persistent 256^2 kernel (one workgroup per CU, epilogue and K loop in series).  ADT_GEMM_TILE is read once
per process: run once per value."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from adt_str_amd import kernels as K

dev = "cuda:0"
M = 63104


def timeit(fn, n=40, warm=25):
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
    a = torch.randn((M, 768), device=dev).bfloat16()
    a2 = torch.randn((M, 3072), device=dev).bfloat16()
    w1 = torch.randn((3072, 768), device=dev).bfloat16()
    w2 = torch.randn((768, 3072), device=dev).bfloat16()
    wq = torch.randn((2304, 768), device=dev).bfloat16()
    wo = torch.randn((768, 768), device=dev).bfloat16()
    b1, bo, bq = torch.zeros(3072, device=dev), torch.zeros(768, device=dev), torch.zeros(2304, device=dev)
    u = torch.empty((M, 3072), dtype=torch.bfloat16, device=dev)
    z = torch.empty((M, 3072), dtype=torch.bfloat16, device=dev)
    z2 = torch.empty((M, 768), dtype=torch.bfloat16, device=dev)
    res = torch.randn((M, 768), device=dev)
    site = K.drop_site(0.1, 1, 5)
    forms = {
        "FFN-1": lambda: K.gemm(a, w1, bias=b1, act=1, act_grad_out=u, drop=site),
        "bare N=3072 K=768": lambda: K.gemm(a, w1, out=z),
        "bare N=768 K=3072": lambda: K.gemm(a2, w2, out=z2),
        "QKV (bias)": lambda: K.gemm(a, wq, bias=bq),
        "out-proj": lambda: K.gemm(a, wo, bias=bo, out_dtype=torch.float32, drop=site, residual=res),
        "FFN-2": lambda: K.gemm(a2, w2, bias=bo, out_dtype=torch.float32, drop=site, residual=res),
    }
    for _ in range(2):
        print(f"ADT_GEMM_TILE={os.environ.get('ADT_GEMM_TILE', 'auto')}: " + "; ".join(f"{name} {timeit(fn):.3f} ms" for name, fn in forms.items()), flush=True)


if __name__ == "__main__":
    main()
