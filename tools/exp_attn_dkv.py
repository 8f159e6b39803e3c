#!/usr/bin/env python3
"""A/B of the dK/dV kernels (ADT_ATTN_DKV = 2 producer / consumer pairs, 3 symmetric staggered, 4 symmetric, not staggered) at the
encoder and cross-attention shapes: per-kernel time of the backward pair from HIP events, dropout on and off, plus a check that
every variant returns the same gradients (3 and 4 bitwise; 2 to rounding: its sum over query blocks runs in another order)."""
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from adt_str_amd import kernels as K

dev = "cuda:0"


def timeit(fn, n=30, warm=20):
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
    scale = 1 / math.sqrt(128)
    for name, (B, H, Sq, Sk) in (("encoder self-attention", (64, 6, 986, 986)), ("cross-attention", (64, 6, 128, 986))):
        d = H * 128
        torch.manual_seed(0)
        q = torch.randn((B * Sq, d), device=dev).bfloat16()
        kv = torch.randn((B * Sk, 2 * d), device=dev).bfloat16()
        k, v = kv[:, :d], kv[:, d:]
        dout = torch.randn((B * Sq, d), device=dev).bfloat16()
        for drop in (None, (0.1, 12345)):
            o, lse = K.attn_fwd(q, k, v, B, H, Sq, Sk, scale, drop=drop)
            fwd = timeit(lambda: K.attn_fwd(q, k, v, B, H, Sq, Sk, scale, drop=drop))
            res = {}
            for var in ("2", "3", "4"):
                os.environ["ADT_ATTN_DKV"] = var
                dq, dkv = torch.zeros_like(q), torch.zeros_like(kv)
                fn = lambda: K.attn_bwd(q, k, v, o, dout, lse, dq, dkv[:, :d], dkv[:, d:], B, H, Sq, Sk, scale, drop=drop)
                fn()
                torch.cuda.synchronize()
                res[var] = (timeit(fn), dkv.clone())
            os.environ.pop("ADT_ATTN_DKV")
            flops_bwd = 10.0 * B * H * Sq * Sk * 128
            same34 = torch.equal(res["3"][1], res["4"][1])
            err23 = (res["3"][1].float() - res["2"][1].float()).abs().max().item() / res["2"][1].float().abs().max().item()
            print(f"{name} B={B} H={H} Sq={Sq} Sk={Sk} dropout={'on' if drop else 'off'}: fwd {fwd:.3f} ms | bwd pair: "
                  + " ".join(f"v{v_}={res[v_][0]:.3f} ms ({flops_bwd / res[v_][0] / 1e9:.0f} TF/s)" for v_ in ("2", "3", "4"))
                  + f" | v3==v4 bitwise: {same34}, max|v3-v2|/max = {err23:.2e}", flush=True)


if __name__ == "__main__":
    main()
