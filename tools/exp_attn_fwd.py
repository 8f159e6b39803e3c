#!/usr/bin/env python3
"""A/B of the attention forward: the software-pipelined kernel (attention_fwd.hip, default) against the first one (ADT_ATTN_FWD=1) --
agreement on a set of shapes (with / without dropout, masks), then timings at the training step's shapes."""
import math
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from adt_str_amd import kernels as K

dev = "cuda:0"


def rnd(shape, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(shape, generator=g).to(dev)


def run(mode, *args, **kw):
    if mode == "1":
        os.environ["ADT_ATTN_FWD"] = "1"
    else:
        os.environ.pop("ADT_ATTN_FWD", None)
    o, lse = K.attn_fwd(*args, **kw)
    torch.cuda.synchronize()
    return o, lse


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
    scale = 1.0 / math.sqrt(128)
    cases = [(2, 2, 128, 128, False, False), (1, 1, 32, 64, False, False), (2, 3, 77, 50, True, True), (2, 2, 257, 300, True, True),
             (1, 2, 449, 64, False, False), (3, 6, 128, 986, False, False), (2, 6, 986, 986, False, False), (1, 1, 5, 200, False, False),
             (4, 6, 128, 128, True, True), (1, 1, 40, 1, False, False), (2, 1, 70, 65, False, True)]
    if len(sys.argv) < 2 or sys.argv[1] != "time":
        for (B, H, Sq, Sk, causal, padded) in cases:
            d = H * 128
            q = rnd((B * Sq, d), 11).bfloat16()
            kv = rnd((B * Sk, 2 * d), 12).bfloat16()
            kk, v = kv[:, :d], kv[:, d:]
            key_len = torch.tensor([max(1, Sk - 7 * (i + 1)) for i in range(B)], dtype=torch.int32, device=dev) if padded else None
            for drop in (None, (0.1, 777)):
                o1, l1 = run("1", q, kk, v, B, H, Sq, Sk, scale, causal, key_len, drop=drop)
                o2, l2 = run("2", q, kk, v, B, H, Sq, Sk, scale, causal, key_len, drop=drop)
                o3, l3 = run("2", q, kk, v, B, H, Sq, Sk, scale, causal, key_len, drop=drop)
                os.environ["ADT_ATTN_FWD_WAVES"] = "4" if Sq > 128 else "8"          # the other workgroup size: same values, bit for bit
                o4, l4 = run("2", q, kk, v, B, H, Sq, Sk, scale, causal, key_len, drop=drop)
                os.environ.pop("ADT_ATTN_FWD_WAVES")
                eo = (o1.float() - o2.float()).abs().max().item()
                el = (l1 - l2).abs().max().item()
                rep = torch.equal(o2, o3) and torch.equal(l2, l3) and torch.equal(o2, o4) and torch.equal(l2, l4)
                bad = (not math.isfinite(eo)) or eo > 2e-2 * o1.float().abs().max().item() + 1e-6 or (not math.isfinite(el)) or el > 1e-3
                print(f"{'BAD ' if bad or not rep else 'ok  '} B{B} H{H} Sq{Sq} Sk{Sk} causal{int(causal)} pad{int(padded)} drop{drop is not None}: "
                      f"out {eo:.3e}/{o1.float().abs().max().item():.2e} lse {el:.3e} repeatable {rep}", flush=True)
    for name, (B, H, Sq, Sk, causal) in {"encoder": (64, 6, 986, 986, False), "cross": (64, 6, 128, 986, False), "causal": (64, 6, 128, 128, True)}.items():
        d = H * 128
        q = rnd((B * Sq, d), 1).bfloat16()
        kv = rnd((B * Sk, 2 * d), 2).bfloat16()
        kk, v = kv[:, :d], kv[:, d:]
        for drop in (None, (0.1, 5)):
            res = {}
            for mode in ("1", "2"):
                if mode == "1":
                    os.environ["ADT_ATTN_FWD"] = "1"
                else:
                    os.environ.pop("ADT_ATTN_FWD", None)
                res[mode] = timeit(lambda: K.attn_fwd(q, kk, v, B, H, Sq, Sk, scale, causal, None, drop=drop))
            os.environ["ADT_ATTN_FWD_WAVES"] = "4" if Sq > 128 else "8"
            other = timeit(lambda: K.attn_fwd(q, kk, v, B, H, Sq, Sk, scale, causal, None, drop=drop))
            os.environ.pop("ADT_ATTN_FWD_WAVES")
            fl = 4.0 * B * H * Sq * Sk * 128
            print(f"{name} dropout {drop is not None}: first {res['1']:.3f} ms, pipelined {res['2']:.3f} ms ({fl / res['2'] / 1e9:.0f} TFLOP/s algorithmic); "
                  f"with {'4' if Sq > 128 else '8'} waves per workgroup {other:.3f} ms", flush=True)


if __name__ == "__main__":
    main()
