#!/usr/bin/env python3
"""The attention forward (attention_fwd.hip) at both workgroup sizes (ADT_ATTN_FWD_WAVES=4 / 8): bitwise agreement with each other and
from run to run on a set of shapes (with / without dropout, masks) and closeness to a plain fp32 softmax(Q K^T) V, then timings at the
training step's shapes.  (profiles/r04/attn_fwd_paths.txt still holds the A/B against the first, unpipelined kernel it replaced.)"""
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


def run(waves, *args, **kw):
    os.environ["ADT_ATTN_FWD_WAVES"] = str(waves)
    o, lse = K.attn_fwd(*args, **kw)
    torch.cuda.synchronize()
    os.environ.pop("ADT_ATTN_FWD_WAVES")
    return o, lse


def reference(q, kk, v, B, H, Sq, Sk, scale, causal, key_len, keep=None, p=0.0):
    """fp32 softmax(Q K^T + masks) V; ``keep`` (bool [B, H, Sq, Sk]): dropout on the probabilities with THAT mask, kept ones / (1 - p)."""
    d = 128
    qh = q.float().view(B, Sq, H, d).transpose(1, 2)
    kh = kk.float().reshape(B, Sk, H, d).transpose(1, 2)
    vh = v.float().reshape(B, Sk, H, d).transpose(1, 2)
    s = qh @ kh.transpose(-1, -2) * scale
    if causal:
        s = s + torch.triu(torch.full((Sq, Sk), -1e4, device=dev), diagonal=1)
    if key_len is not None:
        s = s + (torch.arange(Sk, device=dev)[None, :] >= key_len[:, None]).float()[:, None, None, :] * -1e4
    pr = torch.softmax(s, -1)
    if keep is not None:
        pr = pr * keep.float() / (1.0 - p)
    return (pr @ vh).transpose(1, 2).reshape(B * Sq, H * d)


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
            ref = reference(q, kk, v, B, H, Sq, Sk, scale, causal, key_len)
            for drop in (None, (0.1, 777)):
                o4, l4 = run(4, q, kk, v, B, H, Sq, Sk, scale, causal, key_len, drop=drop)
                o8, l8 = run(8, q, kk, v, B, H, Sq, Sk, scale, causal, key_len, drop=drop)
                o8b, l8b = run(8, q, kk, v, B, H, Sq, Sk, scale, causal, key_len, drop=drop)
                rep = torch.equal(o8, o8b) and torch.equal(l8, l8b) and torch.equal(o4, o8) and torch.equal(l4, l8)
                rf = ref
                if drop is not None and Sq > 1:          # the same-mask reference: the mask is the one the kernel reports (its keep bits, which
                    ob, saved = K.attn_fwd(q, kk, v, B, H, Sq, Sk, scale, causal, key_len, drop=drop, save_bits="force")   # tests/test_dropout_gpu.py pins to the oracle's)
                    rep = rep and torch.equal(ob, o8)
                    rf = reference(q, kk, v, B, H, Sq, Sk, scale, causal, key_len, K.keep_bits_to_mask(saved.bits, B, H, Sq, Sk), drop[0])
                eo = (o8.float() - rf).abs().max().item()
                bad = (not rep) or not eo <= 2e-2 * rf.abs().max().item() + 1e-6 or not bool(torch.isfinite(o8.float()).all())
                print(f"{'BAD ' if bad else 'ok  '} B{B} H{H} Sq{Sq} Sk{Sk} causal{int(causal)} pad{int(padded)} drop{drop is not None}: "
                      f"out vs fp32{' (same mask)' if drop is not None else ''} {eo:.3e}/{rf.abs().max().item():.2e}, 4 = 8 waves and repeatable {rep}", flush=True)
    for name, (B, H, Sq, Sk, causal) in {"encoder": (64, 6, 986, 986, False), "cross": (64, 6, 128, 986, False), "causal": (64, 6, 128, 128, True)}.items():
        d = H * 128
        q = rnd((B * Sq, d), 1).bfloat16()
        kv = rnd((B * Sk, 2 * d), 2).bfloat16()
        kk, v = kv[:, :d], kv[:, d:]
        for drop in (None, (0.1, 5)):
            t = {}
            for waves in (4, 8):
                os.environ["ADT_ATTN_FWD_WAVES"] = str(waves)
                t[waves] = timeit(lambda: K.attn_fwd(q, kk, v, B, H, Sq, Sk, scale, causal, None, drop=drop))
            os.environ.pop("ADT_ATTN_FWD_WAVES")
            fl = 4.0 * B * H * Sq * Sk * 128
            best = t[8] if Sq > 128 else t[4]
            print(f"{name} dropout {drop is not None}: 4 waves x 2 workgroups {t[4]:.3f} ms, 8 waves persistent {t[8]:.3f} ms "
                  f"(product choice {best:.3f} ms = {fl / best / 1e9:.0f} TFLOP/s algorithmic)", flush=True)


if __name__ == "__main__":
    main()
