#!/usr/bin/env python3
"""Tile timeline of the persistent 256 x 256 NT GEMM (experiment build of gemm.hip, -DADT_GEMM_EXPERIMENT, loaded through ADT_LIB_PATH):
per tile of four workgroups of one XCD group the s_memtime ticks of its K loop, its epilogue and the wait that ends it -- for the bare
product and the FFN-1 form at the encoder shape, with and without ADT_GEMM_STAGGER.
    make -C adt_str_amd/csrc && hipcc ... -DADT_GEMM_EXPERIMENT -c gemm.hip ... -o adt_str_amd/libadt_hip_gemmexp.so
    ADT_LIB_PATH=$PWD/adt_str_amd/libadt_hip_gemmexp.so python tools/probe/gemm_tile_stamps.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from adt_str_amd import kernels as K

dev = "cuda:0"
M = 63104
a = torch.randn((M, 768), device=dev).bfloat16()
w1 = torch.randn((3072, 768), device=dev).bfloat16()
wo = torch.randn((768, 768), device=dev).bfloat16()
b1 = torch.zeros(3072, device=dev)
u = torch.empty((M, 3072), dtype=torch.bfloat16, device=dev)
z = torch.empty((M, 3072), dtype=torch.bfloat16, device=dev)
zo = torch.empty((M, 768), dtype=torch.bfloat16, device=dev)
site = K.drop_site(0.1, 1, 5)
forms = {"bare N=3072": lambda: K.gemm(a, w1, out=z),
         "FFN-1 form": lambda: K.gemm(a, w1, bias=b1, act=1, act_grad_out=u, drop=site),
         "bare N=768": lambda: K.gemm(a, wo, out=zo)}
for name, fn in forms.items():
    for ticks in (0, 15000):
        os.environ["ADT_GEMM_STAGGER"] = str(ticks)
        for _ in range(30):
            fn()
        torch.cuda.synchronize()
        print(f"== {name}, stagger {ticks}", file=sys.stderr, flush=True)
        os.environ["ADT_GEMM_STAMPS"] = "1"
        fn()
        os.environ.pop("ADT_GEMM_STAMPS")
        torch.cuda.synchronize()
