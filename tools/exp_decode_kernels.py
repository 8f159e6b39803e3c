#!/usr/bin/env python3
"""The decode step's two kernel kinds alone: single-query attention over 986 memory keys / a 1000-entry cache, and the M = 8
projections, against the tiled kernels (ADT_ATTN_NO_DECODE=1 / ADT_GEMM_NO_SKINNY=1 in the environment of a second run)."""
import math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from adt_str_amd import kernels as K
dev = "cuda:0"
def timeit(fn, n=50):
    """GPU time per launch: the launches are captured into a HIP graph and replayed (eager loops of 5 us kernels time the host)."""
    for _ in range(3): fn()
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (5 * n) * 1e3
B, H, d = 8, 6, 768
scale = 1 / math.sqrt(128)
for Sk, klen in ((986, None), (1000, 50), (1000, 1000)):
    kv = torch.randn(B * Sk, 3 * d, device=dev).bfloat16()
    q = torch.randn(B, 3 * d, device=dev).bfloat16()[:, :d]
    kl = None if klen is None else torch.full((B,), klen, dtype=torch.int32, device=dev)
    t = timeit(lambda: K.attn_fwd(q, kv[:, d:2 * d], kv[:, 2 * d:], B, H, 1, Sk, scale, key_len=kl))
    print(f"attention, 1 query x {Sk} keys (key_len {klen}), B={B}: {t:.1f} us")
for M in (8, 32, 64):
    for N, Kd in ((2304, 768), (768, 768), (3072, 768), (768, 3072), (1400, 768)):
        a = torch.randn(M, Kd, device=dev).bfloat16(); w = torch.randn(N, Kd, device=dev).bfloat16(); bias = torch.randn(N, device=dev)
        t = timeit(lambda: K.gemm(a, w, bias=bias))
        print(f"GEMM M={M} N={N} K={Kd}: {t:.1f} us")
