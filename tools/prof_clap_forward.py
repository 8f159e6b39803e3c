#!/usr/bin/env python3
"""N forwards of HtsatEncoder.forward on one fixed mel batch (512 clips), for the per-forward PMC totals of tools/run_pmc_clap_forward.sh."""
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
flags = torch.zeros(512, dtype=torch.bool)
flags[3] = True
mel = wrap.features.mel(clips)
for _ in range(int(sys.argv[1])):
    wrap.encoder.forward(mel, flags)
torch.cuda.synchronize()
