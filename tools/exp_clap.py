#!/usr/bin/env python3
"""A/B of the HTSAT tower with and without the fused row-block kernels (ADT_HTSAT_FUSED), 512 clips: ms per forward."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from adt_str_amd.clap_encoder import ClapWrapper, random_init_clap_model

dev = "cuda:0"
wrap = ClapWrapper("random-init", dev, 48000, clap_model=random_init_clap_model(0))
rng = np.random.default_rng(7)
clips = [torch.from_numpy((rng.standard_normal(int(n)) * 0.2).astype(np.float32)).to(dev) for n in rng.integers(4800, 96001, 512)]
mel = wrap.features.mel(clips)
flags = torch.zeros(512, dtype=torch.bool)
flags[3] = True
for fused, mlp in (("0", "2"), ("1", "2"), ("1", "3"), ("0", "2"), ("1", "2"), ("1", "3")):
    os.environ["ADT_HTSAT_FUSED"] = fused
    os.environ["ADT_HTSAT_MLP"] = mlp
    for _ in range(3):
        wrap.encoder.forward(mel, flags)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(8):
        wrap.encoder.forward(mel, flags)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 8
    print(f"ADT_HTSAT_FUSED={fused} MLP kernel {mlp}: {ms:.2f} ms per 512 clips = {512 / ms * 1e3:.0f} embeds/s (tower only), {2 * 5.91e9 * 512 / ms / 1e9:.0f} TFLOP/s", flush=True)
