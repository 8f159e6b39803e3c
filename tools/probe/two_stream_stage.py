#!/usr/bin/env python3
"""Does the HTSAT tower gain from running two half-batches on two HIP streams?  The one-workgroup-per-CU kernels of stages 1-2 run their
workgroups in lock step -- every CU loads its token rows at the same time, computes at the same time, stores at the same time -- so the HBM
phases and the compute phases of a launch do not overlap.  Two streams with different kernels in flight de-phase the CUs.
Stage 2 (C = 384, six layers) and stage 0 (C = 96, two layers) at 512 clips: one stream over all clips vs two streams over 256 clips each
(the second one a half layer behind), and scaling of one launch with the number of workgroup rounds."""
import math
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from adt_str_amd import _ffi
from adt_str_amd.clap_encoder import pack_attn_block_weights, pack_rowblock_weights, rowblock, window_bias_layout

dev = "cuda:0"
B = 512


def layer_params(C, nh, g):
    gamma, beta = 1 + 0.1 * torch.randn(C, device=dev, generator=g), 0.1 * torch.randn(C, device=dev, generator=g)
    wqkv = torch.randn((3 * C, C), device=dev, generator=g) / C ** 0.5
    bqkv = 0.2 * torch.randn(3 * C, device=dev, generator=g)
    wo = torch.randn((C, C), device=dev, generator=g) / C ** 0.5
    bo = 0.1 * torch.randn(C, device=dev, generator=g)
    bias = window_bias_layout(0.5 * torch.randn((nh, 64, 64), device=dev, generator=g))
    wpk, qkvb = pack_attn_block_weights(wqkv, bqkv, wo, nh)
    w1 = torch.randn((4 * C, C), device=dev, generator=g) / C ** 0.5
    w2 = torch.randn((C, 4 * C), device=dev, generator=g) / (4 * C) ** 0.5
    b1, b2 = 0.1 * torch.randn(4 * C, device=dev, generator=g), 0.1 * torch.randn(C, device=dev, generator=g)
    return dict(ln=(gamma, beta), wpk=wpk, qkvb=qkvb, bo=bo, bias=bias, mlp_pk=pack_rowblock_weights(2, w1, w2), b1=b1, b2=b2)


def attn(x, nb, R, C, nh, P):
    _ffi.call("adt_htsat_attn_block", x.data_ptr(), nb, R, C, nh, 0, P["ln"][0].data_ptr(), P["ln"][1].data_ptr(), 1e-5, P["wpk"].data_ptr(),
              P["qkvb"].data_ptr(), P["bo"].data_ptr(), P["bias"].data_ptr(), 1, 1.0 / math.sqrt(24.0), _ffi.current_stream())


def mlp(x, C, P):
    rowblock(2, x, P["mlp_pk"], C // 8, P["b1"], ln=P["ln"], eps=1e-5, bias2=P["b2"])


def timed(fn, n=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

# CU-masked streams (hipExtStreamCreateWithCUMask): each partition of the batch on its own share of the CUs, so the partitions run out of phase
import ctypes
hip = ctypes.CDLL("libamdhip64.so")


def masked_stream(bits):
    words = (ctypes.c_uint32 * 8)(*[sum(1 << (i - 32 * w) for i in bits if 32 * w <= i < 32 * w + 32) for w in range(8)])
    h = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(h), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(h.value)


MASKS = {"2 x 128 CUs, contiguous bits": [masked_stream(range(128 * i, 128 * i + 128)) for i in range(2)],
         "4 x 64 CUs, contiguous bits": [masked_stream(range(64 * i, 64 * i + 64)) for i in range(4)]}
t0 = timed(lambda: torch.cuda._sleep(10000), n=3, warm=1)
print(f"torch.cuda._sleep(10000) = {t0:.1f} us")
for (C, nh, R, n_layers) in ((384, 16, 16, 6), (192, 8, 32, 2), (96, 4, 64, 2)):
    g = torch.Generator(device=dev).manual_seed(C)
    M = B * R * R
    x = torch.randn((M, C), device=dev, generator=g)
    P = layer_params(C, nh, g)
    # one launch, 1 ... 4 rounds of workgroups (128 clips = one round at C = 384: 32 768 tokens / 128 per workgroup = 256 workgroups)
    for nb in (B // 4, B // 2, B):
        xs = x[: nb * R * R]
        print(f"C={C}: {nb} clips: attention half {timed(lambda: attn(xs, nb, R, C, nh, P)):.1f} us, MLP half {timed(lambda: mlp(xs, C, P)):.1f} us", flush=True)

    def one_stream():
        for _ in range(n_layers):
            attn(x, B, R, C, nh, P)
            mlp(x, C, P)

    def two_streams(parts=2):
        ev = torch.cuda.Event()
        ev.record()
        done = []
        per = B // parts
        for i, s in enumerate((s1, s2)[:parts]):
            s.wait_event(ev)
            with torch.cuda.stream(s):
                xs = x[i * per * R * R:(i + 1) * per * R * R]
                for _ in range(n_layers):
                    attn(xs, per, R, C, nh, P)
                    mlp(xs, C, P)
                e = torch.cuda.Event()
                e.record()
                done.append(e)
        for e in done:
            torch.cuda.current_stream().wait_event(e)

    def parts_on(streams, delay_us=0):
        ev = torch.cuda.Event()
        ev.record()
        done = []
        per = B // len(streams)
        for i, s in enumerate(streams):
            s.wait_event(ev)
            with torch.cuda.stream(s):
                if delay_us and i:
                    torch.cuda._sleep(int(delay_us * i * 100))            # (wall-clock ticks: 100 per us)
                xs = x[i * per * R * R:(i + 1) * per * R * R]
                for _ in range(n_layers):
                    attn(xs, per, R, C, nh, P)
                    mlp(xs, C, P)
                e = torch.cuda.Event()
                e.record()
                done.append(e)
        for e in done:
            torch.cuda.current_stream().wait_event(e)

    for name, streams in MASKS.items():
        if "contiguous" in name:
            for d in (10, 20, 30, 45, 60):
                print(f"C={C}: {name}: partition i starts {d} us x i late: {timed(lambda: parts_on(streams, d)):.1f} us, again {timed(lambda: parts_on(streams, d)):.1f} us", flush=True)
        with torch.cuda.stream(streams[0]):
            xs = x[: (B // len(streams)) * R * R]
            ta = timed(lambda: attn(xs, B // len(streams), R, C, nh, P))
            tm = timed(lambda: mlp(xs, C, P))
        print(f"C={C}: {name}: one partition alone on its CUs: attention half {ta:.1f} us, MLP half {tm:.1f} us; all partitions, {n_layers} layers: "
              f"{timed(lambda: parts_on(streams)):.1f} us, again {timed(lambda: parts_on(streams)):.1f} us", flush=True)
    a, b = timed(one_stream), timed(two_streams)
    a2, b2 = timed(one_stream), timed(two_streams)
    print(f"C={C}: {n_layers} layers at {B} clips: one stream {a:.1f} us, two streams of {B // 2} clips {b:.1f} us; again {a2:.1f} / {b2:.1f}", flush=True)
    del x
    torch.cuda.empty_cache()
