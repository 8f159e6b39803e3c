#!/usr/bin/env python3
"""Experiment build only (make EXTRA=-DADT_FB_EXPERIMENT): cycle stamps of wave 0's (dQ role) and wave 4's (fan-in role) phases inside one
slice of the 8-wave one-kernel attention backward."""
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
os.environ["ADT_ATTN_BWD_WAVES"] = "8"
names = {1: "ph1 chain (S, dP)", 10: "ph2 arithmetic", 2: "ph2b dQ of the previous slice + publish (waves 0-3)", 3: "barrier A + dS^T write", 5: "ph3 dV / dK",
         6: "vmcnt / lgkmcnt wait", 7: "barrier B", 8: "flag bookkeeping", 11: "fan-in step (waves 4-7)"}
order = [0, 1, 10, 2, 3, 5, 6, 7, 8, 11]
for drop in (None, (0.1, 5)):
    o, saved = K.attn_fwd(q, kk, v, B, H, S, S, scale, False, None, drop=drop, save_bits=drop is not None)
    dq, dkv = torch.zeros_like(q), torch.zeros_like(kv)
    for dbg, who in ((32, "wave 0"), (64, "wave 4")):
        os.environ["ADT_FB_DBG"] = str(dbg)
        for _ in range(3):
            K.attn_bwd(q, kk, v, o, dout, saved, dq, dkv[:, :d], dkv[:, d:], B, H, S, S, scale, False, None, drop=drop)
        torch.cuda.synchronize()
        desc = K._attn_desc(B, H, S, S, q, kk, v, o, scale, False, None, -1e4, drop, 128)
        if drop is not None:
            desc.keep_bits = _ffi.dptr(saved.bits)
        nb = _ffi.load().adt_attn_bwd_workspace_bytes(C.byref(desc))
        ws = K._workspace(nb, q.device)
        st = ws[nb - 128: nb].cpu().view(torch.int64)[:16].tolist()
        print("dropout", drop is not None, who, "cycles per phase (s_memtime):")
        for a, b in zip(order[:-1], order[1:]):
            print(f"  {names[b]:52s} {st[b] - st[a]:7d}")
        print(f"  stamped part of the iteration    {st[11] - st[0]:7d}")
        print(f"  the whole item: prologue (K image, V, slice 0) {st[13] - st[12]}, {S // 32 + (1 if S % 32 else 0)} slices {st[14] - st[13]}, last dQ + dK / dV stores + fan-in tail {st[15] - st[14]}")
