#!/usr/bin/env python3
"""The C = 384 row-block launches of the HTSAT tower's third stage alone (512 clips: M = 131072 tokens): LN -> q|k|v (mode 0), attention
output projection + residual (mode 1), LN -> fc1 -> GELU (mode 4).  A/B harness for builds of htsat_fused.hip (ADT_LIB_PATH)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from adt_str_amd.clap_encoder import pack_rowblock_weights, rowblock

dev = "cuda:0"
M, C = 131072, 384
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn((M, C), device=dev, generator=g)
ln = (torch.randn(C, device=dev, generator=g) * 0.1 + 1.0, torch.randn(C, device=dev, generator=g) * 0.1)
wqkv = (torch.randn((3 * C, C), device=dev, generator=g) * 0.05)
wo = (torch.randn((C, C), device=dev, generator=g) * 0.05)
w1 = (torch.randn((4 * C, C), device=dev, generator=g) * 0.05)
w2 = (torch.randn((C, 4 * C), device=dev, generator=g) * 0.05)
qkv_pk, wo_pk, fc1_pk = pack_rowblock_weights(0, wqkv), pack_rowblock_weights(1, wo), pack_rowblock_weights(0, w1)
mlp_pk = pack_rowblock_weights(2, w1, w2)
w2b = w2.bfloat16()
b2 = torch.zeros(C, device=dev)
from adt_str_amd import kernels as K
bq, bo, b1 = torch.randn(3 * C, device=dev, generator=g), torch.randn(C, device=dev, generator=g), torch.randn(4 * C, device=dev, generator=g)
qkv = torch.empty((M, 3 * C), dtype=torch.bfloat16, device=dev)
h = torch.empty((M, 4 * C), dtype=torch.bfloat16, device=dev)
ctx = torch.randn((M, C), device=dev, generator=g).bfloat16()


def t(fn, n=20, warm=5):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


forms = {
    "LN->qkv (mode 0)": (lambda: rowblock(0, x, qkv_pk, 3 * C // 32, bq, ln=ln, out16=qkv), 2.0 * M * C * 3 * C),
    "out-proj + x (mode 1)": (lambda: rowblock(1, x, wo_pk, C // 32, bo * 0, a16=ctx), 2.0 * M * C * C),
    "LN->fc1->GELU (mode 4)": (lambda: rowblock(4, x, fc1_pk, 4 * C // 32, b1, ln=ln, out16=h), 2.0 * M * C * 4 * C),
    "fc2 GEMM + x": (lambda: K.gemm(h, w2b, bias=b2, residual=x, out=x2), 2.0 * M * C * 4 * C),
    "whole MLP (mode 2)": (lambda: rowblock(2, x3, mlp_pk, C // 8, b1, ln=ln, bias2=b2), 4.0 * M * C * 4 * C),
}
x2 = torch.empty_like(x)
x3 = x.clone()
# agreement of the one-launch MLP with LN -> fc1 -> GELU + the fc2 GEMM (bf16 hidden activation in both)
rowblock(4, x, fc1_pk, 4 * C // 32, b1, ln=ln, out16=h)
K.gemm(h, w2b, bias=b2, residual=x, out=x2)
rowblock(2, x3, mlp_pk, C // 8, b1, ln=ln, bias2=b2)
print(f"MLP one launch vs two: max|diff| {float((x3 - x2).abs().max()):.3e} (max |out - x| {float((x2 - x).abs().max()):.2f})", flush=True)
x3 = x.clone() * 0.0
for rep in range(2):
    print("; ".join(f"{name} {t(fn):.1f} us ({fl / t(fn) / 1e6:.0f} TFLOP/s)" for name, (fn, fl) in forms.items()), flush=True)
print("checksums", float(qkv.float().abs().mean()), float(h.float().abs().mean()), flush=True)
