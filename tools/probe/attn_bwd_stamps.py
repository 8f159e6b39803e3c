#!/usr/bin/env python3
"""Experiment build only (make EXTRA=-DADT_FB_EXPERIMENT): cycle stamps of one wave's phases inside one slice of the fused attention backward."""
import ctypes as C
import math
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from adt_str_amd import _ffi, kernels as K

dev = "cuda:0"
B, H, S = 64, 6, 986
d = H * 128
g = torch.Generator().manual_seed(1)
q = torch.randn((B * S, d), generator=g).to(dev).bfloat16()
kv = torch.randn((B * S, 2 * d), generator=g).to(dev).bfloat16()
kk, v = kv[:, :d], kv[:, d:]
dout = torch.randn((B * S, d), generator=g).to(dev).bfloat16()
scale = 1.0 / math.sqrt(128)
os.environ["ADT_ATTN_BWD"] = "fused"
names = ["top->ph1", "ph1 (S, dP both blocks + hashes b0)", "ph2 (arith b0 | dQ of prev slice)", "xbarrier+write", "ph3 (dvdk b0 | arith b1)", "ph4 (dvdk b1)", "vmcnt wait", "barrier", "(empty)", "reduce step + own tile"]
for drop in (None, (0.1, 5)):
    o, lse = K.attn_fwd(q, kk, v, B, H, S, S, scale, False, None, drop=drop)
    dq, dkv = torch.zeros_like(q), torch.zeros_like(kv)
    os.environ["ADT_FB_DBG"] = "32"
    for _ in range(3):
        K.attn_bwd(q, kk, v, o, dout, lse, dq, dkv[:, :d], dkv[:, d:], B, H, S, S, scale, False, None, drop=drop)
    torch.cuda.synchronize()
    desc = K._attn_desc(B, H, S, S, q, kk, v, o, scale, False, None, -1e4, drop, 128)
    nb = _ffi.load().adt_attn_bwd_workspace_bytes(C.byref(desc))
    ws = K._workspace(nb, q.device)
    st = ws[nb - 128: nb].cpu().view(torch.int64)[:10].tolist()
    print("dropout", drop is not None, "cycles per phase (s_memtime):")
    for i in range(1, 10):
        print(f"  {names[i]:32s} {st[i] - st[i - 1]:7d}")
    print(f"  stamped part of the iteration    {st[9] - st[0]:7d}")
