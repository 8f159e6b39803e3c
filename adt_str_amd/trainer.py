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


class GradReducer:
    """Data-parallel gradient averaging over one flat buffer.

    ``segment_ready(lo, hi)`` is called by the backward pass as soon as ``gflat[lo:hi]`` is final and
    starts an asynchronous all-reduce of that slice (RCCL on the GPUs, gloo in the CPU tests);
    ``finish()`` waits for all of them and turns the sums into means.  Segments must tile the buffer
    (the network's ``flat_range`` segments do); ``covered`` lets callers assert that."""

    def __init__(self, gflat: torch.Tensor, process_group=None):
        self.gflat, self.pg = gflat, process_group
        self.world = dist.get_world_size(process_group)
        self._works, self.covered = [], 0

    def segment_ready(self, lo: int, hi: int):
        self._works.append(dist.all_reduce(self.gflat[lo:hi], op=dist.ReduceOp.SUM, group=self.pg, async_op=True))
        self.covered += hi - lo

    def finish(self):
        for w in self._works:
            w.wait()
        self._works.clear()
        if self.covered != self.gflat.numel():
            raise RuntimeError(f"gradient segments covered {self.covered} of {self.gflat.numel()} elements")
        self.covered = 0
        self.gflat.mul_(1.0 / self.world)


def backward_segments(engine):
    """The flat ranges in the order the backward pass completes them (network._Engine._backward)."""
    segs = [engine.flat_range("decoder.")]
    for L in reversed(engine.enc):
        segs.append(engine.flat_range(L["p"] + "."))
    for pre in ("encoder.dense_layer.", "encoder.layer_norm.", "project_to_mel."):
        segs.append(engine.flat_range(pre))
    return segs


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
        self.reducer = None
        if self.world > 1:
            self.broadcast_parameters()
            self.reducer = GradReducer(self.gflat, self.pg)
            self.eng.grad_ready_hook = self.reducer.segment_ready
        self.eng.refresh_weights(force=True)

    # ---- data parallel -----------------------------------------------------------------
    def broadcast_parameters(self):
        """Rank 0's parameters to everyone, once (DDP constructor semantics); buffers are constants and stay local."""
        dist.broadcast(self.pflat, src=dist.get_global_rank(self.pg, 0) if self.pg is not None else 0, group=self.pg)

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
        if self.reducer is not None:
            self.reducer.finish()
        K.grad_norm(self.gflat, self.max_norm, out=self.norm)
        lr = self.current_lr()
        self.step_no += 1
        K.adamw_step(self.pflat, self.gflat, self.m, self.v, self.step_no, lr, self.betas[0], self.betas[1], self.eps, self.wd,
                     self.norm)
        self.eng.refresh_weights(force=True)
        return out["loss"]


def run_native_training(model: ADTModel, dataset, cfg: dict):
    """Epoch loop over a ``LakhDataset`` with ``FlatTrainer`` (the native counterpart of ``Trainer.train()``)."""
    import random
    t, lg = cfg["training"], cfg["logging"]
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    bs = t["batch_size"]
    steps_per_epoch = len(dataset) // (bs * world)
    total = steps_per_epoch * (t["num_epochs"] or 1)
    min_ratio = (t["min_learning_rate"] / t["learning_rate"]) if t.get("min_learning_rate") else 0.0
    tr = FlatTrainer(model, lr=t["learning_rate"], weight_decay=t["weight_decay"], max_grad_norm=t["max_grad_norm"],
                     total_steps=total, warmup_ratio=t["warmup_ratio"], min_lr_ratio=min_ratio)
    order = list(range(len(dataset)))
    for epoch in range(t["num_epochs"] or 1):
        random.Random(cfg["experiment"]["seed"] + epoch).shuffle(order)          # same permutation on every rank
        shard = order[rank::world]
        for s in range(steps_per_epoch):
            batch = dataset.collate([dataset[i] for i in shard[s * bs:(s + 1) * bs]])
            loss = tr.train_step(batch["wavs"], batch["tokens"].to(batch["wavs"].device), batch["token_lengths"])
            if rank == 0 and lg.get("logging_steps") and (tr.step_no % lg["logging_steps"] == 0):
                print(f"epoch {epoch} step {tr.step_no}/{total} loss {loss.item():.4f} lr {tr.current_lr():.3e}", flush=True)
    return tr
