#!/usr/bin/env python3
"""Race hunt (GPU box): the kernels are deterministic, so any run-to-run difference of the outputs of the hand-synchronised kernels
(attention forward / backward at the encoder and decoder shapes, persistent NT / TN GEMMs in the step's forms, the fused HTSAT
tower) is a synchronisation bug.
Usage: python tools/stress_repro.py [iterations]"""
import math
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from adt_str_amd import kernels as K

dev = "cuda:0"
n_it = int(sys.argv[1]) if len(sys.argv) > 1 else 100
bad = 0


def check(name, fn):
    global bad
    ref = [t.clone() for t in fn()]
    for it in range(n_it):
        out = fn()
        for a, b in zip(ref, out):
            if not torch.equal(a, b):
                bad += 1
                print(f"MISMATCH {name} iteration {it}: max diff {(a.float() - b.float()).abs().max().item()}")
                return
    print(f"{name}: {n_it} identical runs")


for (B, H, Sq, Sk, causal) in [(64, 6, 986, 986, False), (64, 6, 128, 986, False), (64, 6, 128, 128, True), (3, 2, 449, 200, False)]:
    d = H * 128
    g = torch.Generator().manual_seed(Sq + Sk)
    q = torch.randn((B * Sq, d), generator=g).to(dev).bfloat16()
    kv = torch.randn((B * Sk, 2 * d), generator=g).to(dev).bfloat16()
    do = torch.randn((B * Sq, d), generator=g).to(dev).bfloat16()
    scale, drop = 1 / math.sqrt(128), (0.1, 77)
    o, lse = K.attn_fwd(q, kv[:, :d], kv[:, d:], B, H, Sq, Sk, scale, causal, drop=drop)
    dq, dkv, bg = torch.empty_like(q), torch.empty_like(kv), torch.empty(3 * d, device=dev)

    def fwd():
        return K.attn_fwd(q, kv[:, :d], kv[:, d:], B, H, Sq, Sk, scale, causal, drop=drop)

    def bwd():
        K.attn_bwd(q, kv[:, :d], kv[:, d:], o, do, lse, dq, dkv[:, :d], dkv[:, d:], B, H, Sq, Sk, scale, causal, drop=drop, bias_grad=bg)
        return dq, dkv, bg

    check(f"attn fwd {B}x{H}x{Sq}x{Sk}", fwd)
    check(f"attn bwd {B}x{H}x{Sq}x{Sk}", bwd)
    # the training step's backward: keep bits from the forward -> the 8-wave one-kernel form; and the same form without dropout
    ob, saved = K.attn_fwd(q, kv[:, :d], kv[:, d:], B, H, Sq, Sk, scale, causal, drop=drop, save_bits="force")
    o0, lse0 = K.attn_fwd(q, kv[:, :d], kv[:, d:], B, H, Sq, Sk, scale, causal)

    def bwd_bits():
        K.attn_bwd(q, kv[:, :d], kv[:, d:], ob, do, saved, dq, dkv[:, :d], dkv[:, d:], B, H, Sq, Sk, scale, causal, drop=drop, bias_grad=bg)
        return dq, dkv, bg

    def bwd_plain():
        K.attn_bwd(q, kv[:, :d], kv[:, d:], o0, do, lse0, dq, dkv[:, :d], dkv[:, d:], B, H, Sq, Sk, scale, causal, bias_grad=bg)
        return dq, dkv, bg

    check(f"attn bwd, keep bits {B}x{H}x{Sq}x{Sk}", bwd_bits)
    check(f"attn bwd, no dropout {B}x{H}x{Sq}x{Sk}", bwd_plain)

M = 64 * 986
a = torch.randn((M, 768), device=dev).bfloat16()
w = torch.randn((3072, 768), device=dev).bfloat16()
bias = torch.randn(3072, device=dev)
u = torch.empty((M, 3072), dtype=torch.bfloat16, device=dev)
cs = torch.empty(3072, device=dev)
check("NT 256^2 FFN1 + GELU + dropout + column sums", lambda: (K.gemm(a, w, bias=bias, act=1, pre_act_out=u, drop=(0.1, 5), colsum_out=cs), u, cs))
x = torch.randn((M, 3072), device=dev).bfloat16()
gw = torch.empty((768, 3072), device=dev)
check("TN 256^2 split-K wgrad", lambda: (K.gemm(a, x, trans=True, out=gw),))
# the forms the training step launches (compile-time epilogue masks; interior tiles leave their last stores in flight)
u2 = torch.empty((M, 3072), dtype=torch.bfloat16, device=dev)
check("NT 256^2 FFN1 + GELU + dropout + saved factor (roofline form)", lambda: (K.gemm(a, w, bias=bias, act=1, act_grad_out=u2, drop=(0.1, 9)), u2))
res = torch.randn((M, 768), device=dev)
w2 = torch.randn((768, 3072), device=dev).bfloat16()
b2 = torch.randn(768, device=dev)
check("NT 256^2 FFN2 + bias + dropout + residual (fp32 out)", lambda: (K.gemm(x, w2, bias=b2, residual=res, out_dtype=torch.float32, drop=(0.1, 11)),))
wq = torch.randn((2304, 768), device=dev).bfloat16()
check("NT 256^2 QKV (column-group tile order, 9 tile columns)", lambda: (K.gemm(a, wq, bias=torch.zeros(2304, device=dev)),))

# the split-bf16 parity arm's large products on the persistent kernels (round 6: virtual-K-tile operand map, planes split once)
K.set_f32_products("bf16x3")
a32 = torch.randn((M, 768), device=dev)
w32 = torch.randn((3072, 768), device=dev) * 0.05
x32 = torch.randn((M, 3072), device=dev) * 0.05
check("bf16x3 NT on the persistent kernel (planes split once)", lambda: (K.gemm(a32, w32),))
check("bf16x3 TN on the persistent kernel, split K", lambda: (K.gemm(a32, x32, trans=True),))
K.set_f32_products("f32")
del a32, w32, x32

# fused HTSAT tower (K15) + K9: one forward of 64 clips, repeated (round 6: stages 1-2 run the one-workgroup-per-CU kernels with the sub-chunk
# rings -- htsat_attn_big_kernel, htsat_mlp_kernel<384, 1, 1>), then of 512 clips (four workgroups per CU in sequence)
import numpy as np
from adt_str_amd.clap_encoder import ClapWrapper, random_init_clap_model
wrap = ClapWrapper("random-init", dev, 48000, clap_model=random_init_clap_model(0))
rng = np.random.default_rng(3)
clips = [torch.from_numpy((rng.standard_normal(int(n)) * 0.2).astype(np.float32)).to(dev) for n in rng.integers(4800, 96001, 64)]
flags = torch.zeros(64, dtype=torch.bool)
flags[5] = True
n_it = max(10, n_it // 5)
check("CLAP features + fused HTSAT forward, 64 clips", lambda: (wrap.get_audio_features(clips, is_longer=flags),))
clips512 = [torch.from_numpy((rng.standard_normal(int(n)) * 0.2).astype(np.float32)).to(dev) for n in rng.integers(4800, 96001, 512)]
flags512 = torch.zeros(512, dtype=torch.bool)
flags512[77] = True
n_it = max(8, n_it // 2)
check("CLAP features + fused HTSAT forward, 512 clips", lambda: (wrap.get_audio_features(clips512, is_longer=flags512),))
print("FAILED" if bad else "all reproducible")
sys.exit(1 if bad else 0)
