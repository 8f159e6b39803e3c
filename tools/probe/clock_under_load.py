#!/usr/bin/env python3
"""What shader clock and board power does the chip hold under this repo's kernels?  A thread reads the card's hwmon node in sysfs (sclk, average power) while the
main thread runs a bare 256^2-tile NT GEMM, the FFN-1 form, the attention forward / backward and whole training steps, each for ~2 s.
The MFMA peak the roofline is priced against (2.5 PFLOP/s bf16) assumes 2.4 GHz; what a kernel can reach scales with the clock it gets."""
import json
import math
import os
import re
import subprocess
import sys
import threading
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from adt_str_amd import kernels as K

dev = "cuda:0"
samples = []
stop = False


def poll():
    """In-process sysfs reads (bench.ClockPoll's nodes): no rocm-smi child processes (those are python scripts: under a profiler they are
    the exec-after-GPU-init hop the pool forbids)."""
    import bench
    freq, power = bench.ClockPoll.find_nodes(0, bench.ClockPoll.pci_address(0))
    while not stop:
        try:
            mhz = bench.ClockPoll._read_mhz(freq)
            watts = bench.ClockPoll._read(power) / 1e6 if power else -1.0
            samples.append((time.time(), int(round(mhz)), watts))
        except Exception:          # noqa: BLE001
            samples.append((time.time(), -1, -1.0))
        time.sleep(0.05)


def load(name, fn, seconds=2.5):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.time()
    n = 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    while time.time() - t0 < seconds:
        for _ in range(20):
            fn()
        n += 20
        torch.cuda.synchronize()
    e1.record()
    torch.cuda.synchronize()
    t1 = time.time()
    s = [(c, p) for (t, c, p) in samples if t0 + 0.7 <= t <= t1 and c > 0]
    clk = sorted(c for c, _ in s)
    pw = sorted(p for _, p in s)
    print(f"{name:44s} {e0.elapsed_time(e1) / n:8.3f} ms per launch; sclk MHz min / median / max {clk[0] if clk else -1} / {clk[len(clk) // 2] if clk else -1} / {clk[-1] if clk else -1}"
          f"; power W median {pw[len(pw) // 2] if pw else -1:.0f} ({len(s)} samples)", flush=True)


def main():
    global stop
    th = threading.Thread(target=poll, daemon=True)
    th.start()
    time.sleep(1.0)
    idle = [c for (_, c, _) in samples if c > 0]
    print("idle sclk MHz:", idle[-3:] if idle else "rocm-smi gave nothing: " + str(samples[-1:]), flush=True)
    M = 63104
    a = torch.randn((M, 768), device=dev).bfloat16()
    w = torch.randn((3072, 768), device=dev).bfloat16()
    out = torch.empty((M, 3072), dtype=torch.bfloat16, device=dev)
    load("bare NT GEMM M63104 N3072 K768", lambda: K.gemm(a, w, out=out))
    bias = torch.zeros(3072, device=dev)
    u = torch.empty((M, 3072), dtype=torch.bfloat16, device=dev)
    site = K.drop_site(0.1, 1, 5)
    load("FFN-1 form (bias GELU dropout factor)", lambda: K.gemm(a, w, bias=bias, act=1, act_grad_out=u, drop=site))
    big = torch.randn((8192, 8192), device=dev).bfloat16()
    o2 = torch.empty((8192, 8192), dtype=torch.bfloat16, device=dev)
    load("bare NT GEMM 8192^3", lambda: K.gemm(big, big, out=o2))
    load("torch.matmul 8192^3 (hipBLASLt)", lambda: torch.matmul(big, big.t(), out=o2))
    B, H, S = 64, 6, 986
    d = H * 128
    q = torch.randn((B * S, d), device=dev).bfloat16()
    kv = torch.randn((B * S, 2 * d), device=dev).bfloat16()
    sc = 1 / math.sqrt(128)
    load("attention forward, dropout 0.1", lambda: K.attn_fwd(q, kv[:, :d], kv[:, d:], B, H, S, S, sc, False, None, drop=(0.1, 5)))
    o, lse = K.attn_fwd(q, kv[:, :d], kv[:, d:], B, H, S, S, sc, False, None, drop=(0.1, 5))
    dq, dkv = torch.zeros_like(q), torch.zeros_like(kv)
    load("attention backward, dropout 0.1", lambda: K.attn_bwd(q, kv[:, :d], kv[:, d:], o, q, lse, dq, dkv[:, :d], dkv[:, d:], B, H, S, S, sc, False, None, drop=(0.1, 5)))
    x = torch.randn((M, 768), device=dev)
    load("elementwise copy 194 MB (HBM-bound)", lambda: x.clone())
    stop = True


if __name__ == "__main__":
    main()
