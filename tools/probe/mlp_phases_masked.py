#!/usr/bin/env python3
"""Phase stamps of the C = 384 MLP kernel (a -DADT_MLP_PHASES build through ADT_LIB_PATH, ADT_MLP_PRINT=1) on the whole chip and on a CU-masked
stream with half of the CUs: do the row-load / epilogue phases shrink IN CYCLES when fewer CUs compete (bandwidth-bound phases), or only in
time (clock)?"""
import ctypes
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from adt_str_amd.clap_encoder import pack_rowblock_weights, rowblock

dev = "cuda:0"
C = 384
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn((131072, C), device=dev, generator=g)
ln = (torch.randn(C, device=dev, generator=g) * 0.1 + 1.0, torch.randn(C, device=dev, generator=g) * 0.1)
w1, w2 = torch.randn((4 * C, C), device=dev, generator=g) * 0.05, torch.randn((C, 4 * C), device=dev, generator=g) * 0.05
pk = pack_rowblock_weights(2, w1, w2)
b1, b2 = torch.randn(4 * C, device=dev, generator=g), torch.zeros(C, device=dev)
hip = ctypes.CDLL("libamdhip64.so")


def masked_stream(bits):
    words = (ctypes.c_uint32 * 8)(*[sum(1 << (i - 32 * w) for i in bits if 32 * w <= i < 32 * w + 32) for w in range(8)])
    h = ctypes.c_void_p()
    assert hip.hipExtStreamCreateWithCUMask(ctypes.byref(h), 8, words) == 0
    return torch.cuda.ExternalStream(h.value)


for name, s, rows in (("whole chip, 512 clips", torch.cuda.current_stream(), 131072), ("bits 0-127, 256 clips", masked_stream(range(128)), 65536),
                      ("even bits, 256 clips", masked_stream(range(0, 256, 2)), 65536), ("bits 0-63, 256 clips", masked_stream(range(64)), 65536)):
    with torch.cuda.stream(s):
        for _ in range(3):
            print(f"== {name}", file=sys.stderr, flush=True)
            rowblock(2, x[:rows], pk, C // 8, b1, ln=ln, eps=1e-5, bias2=b2)
    torch.cuda.synchronize()
