"""window_attn4 at the three stage shapes of the CLAP tower (512 clips), back to back."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from adt_str_amd import _ffi

dev = "cuda:0"
B = 512
for R, C, nh in ((32, 192, 8), (16, 384, 16), (8, 768, 32)):
    M = B * R * R
    qkv = (torch.randn(M, 3 * C, device=dev)).bfloat16()
    ctx = torch.empty(M, C, dtype=torch.bfloat16, device=dev)
    bias = torch.randn(nh, 64, 64, device=dev)
    def run():
        _ffi.call("adt_window_attn_fwd", qkv.data_ptr(), 3 * C, ctx.data_ptr(), C, bias.data_ptr(), 1, B, R, C, nh, 0, 24 ** -0.5, 0)
    for _ in range(3):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"lds={os.environ.get('ADT_WA4_LDS', 'default')} R={R} C={C}: {ms * 1e3:.1f} us  {(M * 4 * C * 2) / ms / 1e9:.2f} TB/s")
