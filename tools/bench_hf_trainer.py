#!/usr/bin/env python3
"""Step time of the HF-Trainer-style path (ADTTrainer.compute_loss hook -> autograd bridge -> clip_grad_norm_ -> torch AdamW)
next to the native flat-buffer loop, on the model / batch shape of bench.py's train workload.  GPU box only."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from adt_str_amd.masks import create_mask_plain
from adt_str_amd.network import ADTModel, ADTModelConfig

dev = torch.device("cuda:0")
B, L, T = 64, 160000, 128
torch.manual_seed(0)
cfg = ADTModelConfig(input_sec=10.0, time_res=0.01, win_length=2048, sample_rate=16000, dropout=0.1, plain=True, **bench.SETTING1)
model = ADTModel(cfg).to(dev).train()
rng = np.random.default_rng(0)
tok, tl = bench.synthetic_tokens(rng, B, T)
tokens, lens = torch.from_numpy(tok).to(dev), torch.from_numpy(tl).to(dev)
wavs = torch.randn(B, L, device=dev) * 0.1
opt = torch.optim.AdamW(model.parameters(), lr=1e-4, weight_decay=1e-5)


def hf_step():
    tgt_input, labels = tokens[:, :-1], tokens[:, 1:]
    _, pad = create_mask_plain(tgt_input.size(1), lens, dev)
    loss = model(src=wavs, tgt=tgt_input, tgt_mask=None, tgt_padding_mask=pad, labels=labels)
    loss.backward()
    torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
    opt.step()
    opt.zero_grad(set_to_none=True)
    return loss


ONLY = os.environ.get("ADT_HF_ARM")      # profiling aid: run ONE arm ("fused", "ddp-engine", "ddp-own") so that a kernel trace holds nothing else
if not ONLY:
    for _ in range(3):
        hf_step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10):
        hf_step()
    torch.cuda.synchronize()
    print(f"HF-style step (autograd bridge + clip_grad_norm_ + torch AdamW): {(time.perf_counter() - t0) * 100:.2f} ms/step")

# the same loop with the optimizer ADTTrainer.create_optimizer builds (adt_str_amd/optim.py: torch.optim.AdamW's contract on the fused kernel)
from adt_str_amd.optim import FusedAdamW
from adt_str_amd.trainer import no_decay_names
skip = set(no_decay_names(model))
named = list(model.named_parameters())
opt = FusedAdamW([{"params": [p for n, p in named if n not in skip], "weight_decay": 1e-5}, {"params": [p for n, p in named if n in skip], "weight_decay": 0.0}],
                 lr=1e-4, engine=model.engine)
if not ONLY or ONLY == "fused":
    for _ in range(3):
        hf_step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10):
        hf_step()
    torch.cuda.synchronize()
    print(f"HF-style step with FusedAdamW (what train.py's Trainer now builds): {(time.perf_counter() - t0) * 100:.2f} ms/step")
del opt

if not ONLY:
    from adt_str_amd.trainer import FlatTrainer
    tr = FlatTrainer(model, lr=1e-4, weight_decay=1e-5, max_grad_norm=1.0, total_steps=10000, warmup_ratio=0.1)
    for _ in range(3):
        tr.train_step(wavs, tokens, lens)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10):
        tr.train_step(wavs, tokens, lens)
    torch.cuda.synchronize()
    print(f"native FlatTrainer step (same batch, no mixer): {(time.perf_counter() - t0) * 100:.2f} ms/step")


# ---- under a launcher (`torchrun --nproc-per-node 1 tools/bench_hf_trainer.py`): the reference's multi-GPU launch shape -- HF-style step on a
# DistributedDataParallel wrapper over RCCL -- with the engine-driven per-segment reduction (trainer.install_engine_reduction) and with
# DDP's own bucketed all-reduce behind the bridge (what the path did before round 4).  At world size 1 neither sends anything; what is
# timed is the plumbing each arm adds to the step.
if "RANK" in os.environ and "WORLD_SIZE" in os.environ:
    import torch.distributed as dist
    from torch.nn.parallel import DistributedDataParallel as DDP
    from adt_str_amd.trainer import forward_engine_reduced, install_engine_reduction
    dist.init_process_group("nccl", device_id=dev)
    for arm in ("engine-driven reduction", "DDP's own all-reduce"):
        if ONLY and ONLY != ("ddp-engine" if arm.startswith("engine") else "ddp-own"):
            continue
        torch.manual_seed(0)
        m2 = ADTModel(cfg).to(dev).train()
        ddp = DDP(m2, device_ids=[dev.index], broadcast_buffers=False)
        if arm.startswith("engine"):
            install_engine_reduction(ddp, 1)
        named2 = list(m2.named_parameters())
        opt2 = FusedAdamW([{"params": [p for n, p in named2 if n not in skip], "weight_decay": 1e-5}, {"params": [p for n, p in named2 if n in skip], "weight_decay": 0.0}],
                          lr=1e-4, engine=m2.engine)

        def ddp_step():
            tgt_input, labels = tokens[:, :-1], tokens[:, 1:]
            _, pad = create_mask_plain(tgt_input.size(1), lens, dev)
            if arm.startswith("engine"):          # what ADTTrainer.compute_loss calls: the engine reduces, DDP's reducer (and its bucket copies) stays out of it
                loss = forward_engine_reduced(ddp, src=wavs, tgt=tgt_input, tgt_mask=None, tgt_padding_mask=pad, labels=labels)
            else:
                loss = ddp(src=wavs, tgt=tgt_input, tgt_mask=None, tgt_padding_mask=pad, labels=labels)
            loss.backward()
            torch.nn.utils.clip_grad_norm_(m2.parameters(), 1.0)
            opt2.step()
            opt2.zero_grad(set_to_none=True)

        for _ in range(3):
            ddp_step()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10):
            ddp_step()
        torch.cuda.synchronize()
        print(f"HF-style step on DDP over RCCL, world size {dist.get_world_size()}, {arm}: {(time.perf_counter() - t0) * 100:.2f} ms/step", flush=True)
        del opt2, ddp, m2
    dist.destroy_process_group()
