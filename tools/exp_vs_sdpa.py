#!/usr/bin/env python3
"""The attention kernels against the vendor path (torch.nn.functional.scaled_dot_product_attention -> flash / efficient kernels
of this PyTorch-ROCm build) on the encoder's shape, forward and forward + backward, without dropout.  A yardstick only."""
import os, sys, math, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from adt_str_amd import kernels as K
dev = "cuda:0"
B, H, S, D = 64, 6, 986, 128
def timeit(fn, n=20):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
qkv = (torch.randn(B * S, 3 * H * D, device=dev) * 0.5).bfloat16()
d = H * D
q, k, v = qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:]
scale = 1.0 / math.sqrt(D)
o, lse = K.attn_fwd(q, k, v, B, H, S, S, scale, head_dim=D)
do = torch.randn_like(o)
dqkv = torch.empty_like(qkv)
t_f = timeit(lambda: K.attn_fwd(q, k, v, B, H, S, S, scale, head_dim=D))
t_b = timeit(lambda: K.attn_bwd(q, k, v, o, do, lse, dqkv[:, :d], dqkv[:, d:2 * d], dqkv[:, 2 * d:], B, H, S, S, scale, head_dim=D))
fl = 4.0 * B * H * S * S * D / 1e9
print(f"this repo: fwd {t_f:.3f} ms ({fl / t_f:.0f} TF/s), bwd {t_b:.3f} ms ({2.5 * fl / t_b:.0f} TF/s algorithmic)", flush=True)
q4 = qkv[:, :d].reshape(B, S, H, D).transpose(1, 2).contiguous().requires_grad_(True)
k4 = qkv[:, d:2 * d].reshape(B, S, H, D).transpose(1, 2).contiguous().requires_grad_(True)
v4 = qkv[:, 2 * d:].reshape(B, S, H, D).transpose(1, 2).contiguous().requires_grad_(True)
from torch.nn.attention import SDPBackend, sdpa_kernel
for name, be in (("flash", SDPBackend.FLASH_ATTENTION), ("efficient", SDPBackend.EFFICIENT_ATTENTION)):
    try:
        with sdpa_kernel(be):
            t_f = timeit(lambda: F.scaled_dot_product_attention(q4, k4, v4))
            out = F.scaled_dot_product_attention(q4, k4, v4)
            g = torch.randn_like(out)
            def fb():
                o_ = F.scaled_dot_product_attention(q4, k4, v4)
                o_.backward(g)
            t_fb = timeit(fb)
        print(f"torch SDPA {name}: fwd {t_f:.3f} ms ({fl / t_f:.0f} TF/s), fwd+bwd {t_fb:.3f} ms -> bwd ~{t_fb - t_f:.3f} ms", flush=True)
    except Exception as e:
        print(f"torch SDPA {name}: not available ({type(e).__name__}: {str(e)[:120]})", flush=True)
