#!/usr/bin/env python3
"""The 8-wave one-kernel attention backward (ADT_ATTN_BWD_WAVES=8) against the 4-wave one: same inputs, same keep bits; largest
difference per gradient (0: both forms sum the same products in the same orders), then timings of both at the encoder / cross-attention /
decoder shapes."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from adt_str_amd import kernels as K

dev = "cuda:0"


def run(B, H, Sq, Sk, causal, drop, waves, seed=0, pad=False, time_it=False):
    g = torch.Generator(device=dev).manual_seed(seed)
    d = H * 128
    q = (torch.randn((B * Sq, d), device=dev, generator=g) * 0.5).bfloat16()
    kv = (torch.randn((B * Sk, 2 * d), device=dev, generator=g) * 0.5).bfloat16()
    dout = (torch.randn((B * Sq, d), device=dev, generator=g) * 0.5).bfloat16()
    kk, v = kv[:, :d], kv[:, d:]
    key_len = torch.tensor([Sk - 7 * (i % 3) for i in range(B)], dtype=torch.int32, device=dev) if pad else None
    site = K.drop_site(0.1, 3, 9) if drop else None
    scale = 128 ** -0.5
    o, saved = K.attn_fwd(q, kk, v, B, H, Sq, Sk, scale, causal, key_len, drop=site, save_bits="force" if drop else False)
    os.environ["ADT_ATTN_BWD"] = "fused"
    os.environ["ADT_ATTN_BWD_WAVES"] = str(waves)
    dq, dkv = torch.zeros_like(q), torch.zeros_like(kv)
    fn = lambda: K.attn_bwd(q, kk, v, o, dout, saved, dq, dkv[:, :d], dkv[:, d:], B, H, Sq, Sk, scale, causal, key_len, drop=site)
    fn()
    torch.cuda.synchronize()
    ms = None
    if time_it:
        for _ in range(10):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 30
    return dq.float(), dkv.float(), ms


def main():
    os.environ["ADT_ATTN_BWD_CHECK"] = "1"
    shapes = [(2, 2, 128, 128, False), (1, 1, 32, 64, False), (2, 3, 77, 50, True), (2, 2, 257, 300, True), (1, 2, 449, 64, False), (3, 6, 128, 986, False),
              (2, 6, 986, 986, False), (4, 6, 128, 128, True)]
    for (B, H, Sq, Sk, causal) in shapes:
        for drop in (False, True):
            pad = causal
            a = run(B, H, Sq, Sk, causal, drop, 4, pad=pad)
            b = run(B, H, Sq, Sk, causal, drop, 8, pad=pad)
            c = run(B, H, Sq, Sk, causal, drop, 8, pad=pad)
            print(f"B{B} H{H} Sq{Sq} Sk{Sk} causal{int(causal)} drop{int(drop)}: max |d dq| {float((a[0] - b[0]).abs().max()):.3e} of {float(a[0].abs().max()):.2e}, "
                  f"|d dkv| {float((a[1] - b[1]).abs().max()):.3e} of {float(a[1].abs().max()):.2e}; repeatable {bool(torch.equal(b[0], c[0]) and torch.equal(b[1], c[1]))}", flush=True)
    for name, (B, H, Sq, Sk, causal) in (("encoder", (64, 6, 986, 986, False)), ("cross", (64, 6, 128, 986, False)), ("causal", (64, 6, 128, 128, True))):
        for drop in (False, True):
            t4 = run(B, H, Sq, Sk, causal, drop, 4, time_it=True)[2]
            t8 = run(B, H, Sq, Sk, causal, drop, 8, time_it=True)[2]
            print(f"{name} dropout {drop}: 4 waves {t4:.3f} ms, 8 waves {t8:.3f} ms", flush=True)


if __name__ == "__main__":
    main()
