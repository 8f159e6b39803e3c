"""Training step for the ADT network on one GPU per process.

Mirrors what HF ``Trainer`` does for the reference (``train.py:163-250``:
AdamW ``adamw_torch``, ``weight_decay`` 1e-5, ``max_grad_norm`` 1.0, warm-up +
cosine schedule, data-parallel gradient averaging) with an MI355X-first layout:

  * parameters, gradients and both Adam moments are single flat fp32 buffers
    (288 GB of HBM: no reason to keep 132 small tensors) -- one fused
    clip + AdamW launch, no per-parameter loop;
  * data parallelism = one ``all_reduce`` per finished segment of the flat gradient
    buffer (decoder, then each encoder layer), issued from the backward pass as
    soon as the segment is final so RCCL's xGMI traffic overlaps the remaining
    backward kernels; constants (PE tables, Hann window, mel filterbank) are never
    broadcast (the reference's DDP re-broadcasts them every forward);
  * the global gradient norm and the clip factor stay on the device (no ``.item()``
    per step; the reference syncs the host every step through ``logging_steps: 1``).
"""
from __future__ import annotations

import math
from typing import Optional

import torch
import torch.distributed as dist

from . import kernels as K
from .network import ADTModel


def cosine_with_warmup(step: int, total_steps: int, warmup_steps: int, min_ratio: float = 0.0) -> float:
    """LR multiplier of HF ``get_cosine_schedule_with_warmup`` (and its ``min_lr`` variant)."""
    if step < warmup_steps:
        return step / max(1, warmup_steps)
    prog = (step - warmup_steps) / max(1, total_steps - warmup_steps)
    return min_ratio + (1.0 - min_ratio) * 0.5 * (1.0 + math.cos(math.pi * min(prog, 1.0)))


class FlatTrainer:
    def __init__(self, model: ADTModel, lr: float = 1e-4, weight_decay: float = 1e-5, betas=(0.9, 0.999), eps: float = 1e-8,
                 max_grad_norm: float = 1.0, total_steps: int = 1000, warmup_ratio: float = 0.1, min_lr_ratio: float = 0.0,
                 process_group=None):
        self.model, self.eng = model, model.engine
        self.lr, self.wd, self.betas, self.eps, self.max_norm = lr, weight_decay, betas, eps, max_grad_norm
        self.total_steps, self.warmup = total_steps, int(total_steps * warmup_ratio)
        self.min_lr_ratio = min_lr_ratio
        self.step_no = 0
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_available() and dist.is_initialized() else 1
        named = self.eng.named
        dev = next(iter(named.values())).device
        n = sum(p.numel() for p in named.values())
        # flatten: every parameter becomes a view of one buffer (state-dict keys and shapes are untouched)
        self.pflat = torch.empty(n, dtype=torch.float32, device=dev)
        off = 0
        for p in named.values():
            self.pflat[off:off + p.numel()].copy_(p.data.reshape(-1))
            p.data = self.pflat[off:off + p.numel()].view_as(p)
            off += p.numel()
        self.m = torch.zeros_like(self.pflat)
        self.v = torch.zeros_like(self.pflat)
        self.gflat, _ = self.eng.grad_buffers()
        self.norm = torch.zeros(2, dtype=torch.float32, device=dev)
        self._works = []
        if self.world > 1:
            self.broadcast_parameters()
            self.eng.grad_ready_hook = self._reduce_segment
        self.eng.refresh_weights(force=True)

    # ---- data parallel -----------------------------------------------------------------
    def broadcast_parameters(self):
        """Rank 0's parameters to everyone, once (DDP constructor semantics); buffers are constants and stay local."""
        dist.broadcast(self.pflat, src=dist.get_global_rank(self.pg, 0) if self.pg is not None else 0, group=self.pg)

    def _reduce_segment(self, lo: int, hi: int):
        seg = self.gflat[lo:hi]
        self._works.append(dist.all_reduce(seg, op=dist.ReduceOp.SUM, group=self.pg, async_op=True))

    def _finish_reduce(self):
        for w in self._works:
            w.wait()
        self._works.clear()

    # ---- one optimisation step ----------------------------------------------------------
    def current_lr(self) -> float:
        return self.lr * cosine_with_warmup(self.step_no, self.total_steps, self.warmup, self.min_lr_ratio)

    def train_step(self, wavs, tokens, token_lengths):
        """``ADTTrainer.compute_loss`` (train.py:40-78) + backward + clip + AdamW.  Returns the loss (device scalar)."""
        self.model.train()
        tgt_in, labels = tokens[:, :-1], tokens[:, 1:]
        T = tgt_in.shape[1]
        pad = torch.arange(T, device=tokens.device).unsqueeze(0) >= token_lengths.to(tokens.device).unsqueeze(1)
        out = self.eng.loss_and_grads(wavs, tgt_in, pad, labels, want_grads=True)
        if self.world > 1:
            self._finish_reduce()
            self.gflat.mul_(1.0 / self.world)
        K.grad_norm(self.gflat, self.max_norm, out=self.norm)
        lr = self.current_lr()
        self.step_no += 1
        K.adamw_step(self.pflat, self.gflat, self.m, self.v, self.step_no, lr, self.betas[0], self.betas[1], self.eps, self.wd,
                     self.norm)
        self.eng.refresh_weights(force=True)
        return out["loss"]
