#!/usr/bin/env python3
"""Experiment: are the K = 768 GEMMs' epilogues bound by every CU of an XCD writing its tile at the same moment?  ADT_GEMM_STAGGER=<ticks>
starts every second workgroup of an XCD group late (gemm.hip), so that half of the XCD's CUs are in their K loop while the other half
stores.  Times the FFN-1 form, the bare product, the out-proj and FFN-2 forms at the encoder shape for a sweep of delays."""
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
    wo = torch.randn((768, 768), device=dev).bfloat16()
    b1, bo = torch.zeros(3072, device=dev), torch.zeros(768, device=dev)
    u = torch.empty((M, 3072), dtype=torch.bfloat16, device=dev)
    z = torch.empty((M, 3072), dtype=torch.bfloat16, device=dev)
    res = torch.randn((M, 768), device=dev)
    site = K.drop_site(0.1, 1, 5)
    forms = {
        "FFN-1 (bias + GELU + dropout + saved factor)": lambda: K.gemm(a, w1, bias=b1, act=1, act_grad_out=u, drop=site),
        "bare N=3072 K=768": lambda: K.gemm(a, w1, out=z),
        "out-proj (bias + dropout + residual, fp32 out)": lambda: K.gemm(a, wo, bias=bo, out_dtype=torch.float32, drop=site, residual=res),
        "FFN-2 (bias + dropout + residual, fp32 out)": lambda: K.gemm(a2, w2, bias=bo, out_dtype=torch.float32, drop=site, residual=res),
    }
    for _ in range(2):
        for ticks in (0, 5000, 10000, 20000, 30000, 45000, 0):
            os.environ["ADT_GEMM_STAGGER"] = str(ticks)
            print(f"stagger {ticks:6d}: " + "; ".join(f"{name.split(' (')[0]} {timeit(fn):.3f} ms" for name, fn in forms.items()), flush=True)
    os.environ.pop("ADT_GEMM_STAGGER")


if __name__ == "__main__":
    main()
