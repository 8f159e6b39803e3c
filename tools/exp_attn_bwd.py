#!/usr/bin/env python3
"""A/B of the attention backward: the one-kernel path (attention_bwd_fused.hip) against the two-kernel path (ADT_ATTN_BWD=split), and with
dropout the one-kernel path fed the forward's keep bits (attn_fwd(save_bits="force"): the default of the training step) against both
-- agreement on a set of shapes (with / without dropout, masks; the bits path must give the hashing fused path's bits exactly), then timings at
the training step's shapes, forward with / without the bit stores included."""
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


def run(mode, q, kk, v, o, dout, lse, B, H, Sq, Sk, scale, causal, key_len, drop):
    os.environ["ADT_ATTN_BWD"] = mode
    d = q.shape[1]
    dq, dkv = torch.zeros_like(q), torch.zeros((kk.shape[0], 2 * d), dtype=q.dtype, device=q.device)
    K.attn_bwd(q, kk, v, o, dout, lse, dq, dkv[:, :d], dkv[:, d:], B, H, Sq, Sk, scale, causal, key_len, drop=drop)
    torch.cuda.synchronize()
    return dq, dkv[:, :d], dkv[:, d:]


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
             (1, 2, 449, 64, False, False), (3, 6, 128, 986, False, False), (2, 6, 986, 986, False, False), (1, 1, 1, 200, False, False),
             (4, 6, 128, 128, True, True)]
    if len(sys.argv) < 2 or sys.argv[1] != "time":
        for (B, H, Sq, Sk, causal, padded) in cases:
            d = H * 128
            q = rnd((B * Sq, d), 11).bfloat16()
            kv = rnd((B * Sk, 2 * d), 12).bfloat16()
            kk, v = kv[:, :d], kv[:, d:]
            key_len = torch.tensor([max(1, Sk - 7 * (i + 1)) for i in range(B)], dtype=torch.int32, device=dev) if padded else None
            dout = rnd((B * Sq, d), 13).bfloat16()
            for drop in (None, (0.1, 777)):
                o, lse = K.attn_fwd(q, kk, v, B, H, Sq, Sk, scale, causal, key_len, drop=drop)
                a = run("split", q, kk, v, o, dout, lse, B, H, Sq, Sk, scale, causal, key_len, drop)
                f = run("fused", q, kk, v, o, dout, lse, B, H, Sq, Sk, scale, causal, key_len, drop)
                f2 = run("fused", q, kk, v, o, dout, lse, B, H, Sq, Sk, scale, causal, key_len, drop)
                bits_ok = True
                if drop is not None and Sq > 1:
                    ob, saved = K.attn_fwd(q, kk, v, B, H, Sq, Sk, scale, causal, key_len, drop=drop, save_bits="force")
                    assert isinstance(saved, K.AttnSaved)
                    os.environ.pop("ADT_ATTN_BWD", None)          # the default path: bits -> one kernel
                    dqb, dkvb = torch.zeros_like(q), torch.zeros((kk.shape[0], 2 * d), dtype=q.dtype, device=q.device)
                    K.attn_bwd(q, kk, v, ob, dout, saved, dqb, dkvb[:, :d], dkvb[:, d:], B, H, Sq, Sk, scale, causal, key_len, drop=drop)
                    torch.cuda.synchronize()
                    bits_ok = torch.equal(ob, o) and torch.equal(saved.lse, lse) and all(torch.equal(x, y) for x, y in zip(f, (dqb, dkvb[:, :d], dkvb[:, d:])))
                    if not bits_ok:
                        print("BAD  keep-bits path differs from the hashing one-kernel path", flush=True)
                if os.environ.get("ADT_FB_XCD"):          # experiment build: every hand-off crosses XCDs (one ticket counter for the grid)
                    os.environ["ADT_FB_DBG"] = "16"
                    f3 = run("fused", q, kk, v, o, dout, lse, B, H, Sq, Sk, scale, causal, key_len, drop)
                    os.environ.pop("ADT_FB_DBG")
                    if not all(torch.equal(x, y) for x, y in zip(f, f3)):
                        print("BAD  cross-XCD run differs", flush=True)
                errs = [((x.float() - y.float()).abs().max().item(), y.float().abs().max().item()) for x, y in zip(f, a)]
                rep = all(torch.equal(x, y) for x, y in zip(f, f2))
                bad = any((not math.isfinite(e)) or e > 2e-2 * m + 1e-6 for e, m in errs) or not bits_ok
                print(f"{'BAD ' if bad or not rep else 'ok  '} B{B} H{H} Sq{Sq} Sk{Sk} causal{int(causal)} pad{int(padded)} drop{drop is not None}: "
                      + " ".join(f"{n} {e:.3e}/{m:.2e}" for n, (e, m) in zip(("dq", "dk", "dv"), errs)) + f" repeatable {rep}", flush=True)
    for name, (B, H, Sq, Sk, causal) in {"encoder": (64, 6, 986, 986, False), "cross": (64, 6, 128, 986, False), "causal": (64, 6, 128, 128, True)}.items():
        d = H * 128
        q = rnd((B * Sq, d), 1).bfloat16()
        kv = rnd((B * Sk, 2 * d), 2).bfloat16()
        kk, v = kv[:, :d], kv[:, d:]
        dout = rnd((B * Sq, d), 3).bfloat16()
        for drop in (None, (0.1, 5)):
            o, lse = K.attn_fwd(q, kk, v, B, H, Sq, Sk, scale, causal, None, drop=drop)
            dq, dkv = torch.zeros_like(q), torch.zeros_like(kv)
            dk, dv = dkv[:, :d], dkv[:, d:]
            res = {}
            for mode in ("split", "fused"):
                os.environ["ADT_ATTN_BWD"] = mode
                res[mode] = timeit(lambda: K.attn_bwd(q, kk, v, o, dout, lse, dq, dk, dv, B, H, Sq, Sk, scale, causal, None, drop=drop))
            if mode == "fused" and name == "encoder" and os.environ.get("ADT_FB_SWEEP"):
                for dbg in (1, 2, 3, 4, 7, 8, 15):
                    os.environ["ADT_FB_DBG"] = str(dbg)
                    t = timeit(lambda: K.attn_bwd(q, kk, v, o, dout, lse, dq, dk, dv, B, H, Sq, Sk, scale, causal, None, drop=drop))
                    print(f"   dbg {dbg}: {t:.3f} ms", flush=True)
                os.environ.pop("ADT_FB_DBG")
            fl = 10.0 * B * H * Sq * Sk * 128
            extra = ""
            if drop is not None:
                os.environ.pop("ADT_ATTN_BWD", None)
                _, saved = K.attn_fwd(q, kk, v, B, H, Sq, Sk, scale, causal, None, drop=drop, save_bits="force")
                tb = timeit(lambda: K.attn_bwd(q, kk, v, o, dout, saved, dq, dk, dv, B, H, Sq, Sk, scale, causal, None, drop=drop))
                tf0 = timeit(lambda: K.attn_fwd(q, kk, v, B, H, Sq, Sk, scale, causal, None, drop=drop, out=o))
                tf1 = timeit(lambda: K.attn_fwd(q, kk, v, B, H, Sq, Sk, scale, causal, None, drop=drop, out=o, save_bits="force"))
                extra = (f"; one kernel + keep bits {tb:.3f} ms ({fl / tb / 1e9:.0f} TFLOP/s); forward {tf0:.3f} ms, storing the bits {tf1:.3f} ms;"
                         f" forward + backward: split {tf0 + res['split']:.3f}, bits {tf1 + tb:.3f} ms")
            print(f"{name} dropout {drop is not None}: split {res['split']:.3f} ms, fused {res['fused']:.3f} ms ({fl / res['fused'] / 1e9:.0f} TFLOP/s algorithmic){extra}", flush=True)


if __name__ == "__main__":
    main()
