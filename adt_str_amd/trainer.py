"""Training loop for the ADT network, one GPU per process.

Mirrors what HF ``Trainer`` does for the reference (``train.py:163-250``: AdamW ``adamw_torch`` with HF's decay /
no-decay parameter groups, ``weight_decay`` 1e-5, ``max_grad_norm`` 1.0, warm-up + cosine schedule, data-parallel
gradient averaging, ``save_steps`` / ``save_total_limit`` / ``resume_from_checkpoint`` ``:179-190,228-232,319-323``)
with an MI355X-first layout:

  * parameters, gradients and both Adam moments are single flat fp32 buffers (288 GB of HBM: no reason to keep
    132 small tensors) -- one fused clip + AdamW launch, no per-parameter loop;
  * data parallelism = one ``all_reduce`` per finished segment of the flat gradient buffer (decoder, then each
    encoder layer), issued from the backward pass as soon as the segment is final so RCCL's xGMI traffic overlaps
    the remaining backward kernels; constants (PE tables, Hann window, mel filterbank) are never broadcast (the
    reference's DDP re-broadcasts them every forward);
  * the global gradient norm and the clip factor stay on the device, the loss is read back asynchronously (the
    reference syncs the host every step through ``logging_steps``);
  * the host half of a batch (note mapping, tokenising, mixer planning) runs one batch ahead on a background
    thread (``data.Prefetcher``), where the reference uses up to 16 DataLoader worker processes (``train.py:235-238``).
"""
from __future__ import annotations

import collections
import glob
import json
import math
import os
import random
import re
import shutil
import threading
import time
import warnings
from typing import Callable, Optional

import torch
import torch.distributed as dist
import torch.nn as nn

from . import kernels as K
from .data import Prefetcher


# ----------------------------------------------------------------------------- schedules (HF optimization.py)
def lr_multiplier(kind: str, step: int, total_steps: int, warmup_steps: int, min_ratio: float = 0.0) -> float:
    """LR multiplier HF's ``LambdaLR`` applies at optimizer step ``step`` (0-based count of steps already taken):
    ``cosine`` = ``get_cosine_schedule_with_warmup``; ``cosine_warmup_with_min_lr`` = ``get_cosine_with_min_lr_schedule_with_warmup_lr_rate``
    (what ``create_training_arguments`` selects when ``min_learning_rate`` is set, train.py:239-244); ``linear``, ``constant``,
    ``constant_with_warmup`` as in HF."""
    if kind == "cosine":
        if step < warmup_steps:
            return float(step) / float(max(1, warmup_steps))
        prog = float(step - warmup_steps) / float(max(1, total_steps - warmup_steps))
        return max(0.0, 0.5 * (1.0 + math.cos(math.pi * 0.5 * 2.0 * prog)))
    if kind == "cosine_warmup_with_min_lr":
        s, w, t = float(step), float(warmup_steps), float(total_steps)
        if s < w:
            return (s + 1.0) / max(1.0, w)
        prog = (s - w + 1.0) / max(1.0, t - w)
        factor = 0.5 * (1.0 + math.cos(math.pi * 0.5 * 2.0 * prog))
        return max(0, factor * (1 - min_ratio) + min_ratio)
    if kind == "linear":
        if step < warmup_steps:
            return float(step) / float(max(1, warmup_steps))
        return max(0.0, float(total_steps - step) / float(max(1, total_steps - warmup_steps)))
    if kind == "constant":
        return 1.0
    if kind == "constant_with_warmup":
        return float(step) / float(max(1.0, warmup_steps)) if step < warmup_steps else 1.0
    raise ValueError(f"unsupported lr_scheduler_type {kind!r}")


def cosine_with_warmup(step: int, total_steps: int, warmup_steps: int, min_ratio: float = 0.0) -> float:
    """Kept name: ``lr_multiplier`` of the scheduler the reference's config selects."""
    return lr_multiplier("cosine_warmup_with_min_lr" if min_ratio > 0 else "cosine", step, total_steps, warmup_steps, min_ratio)


_NO_DECAY_PATTERNS = [re.compile(p) for p in (r"bias", r"layernorm", r"rmsnorm", r"(?:^|\.)norm(?:$|\.)", r"_norm(?:$|\.)")]


def no_decay_names(model: nn.Module):
    """Parameter names HF ``Trainer.get_decay_parameter_names`` leaves OUT of the weight-decay group: parameters owned by an
    ``nn.LayerNorm`` and names matching its forbidden patterns (any ``bias``, ``norm`` path components)."""
    ln_owned = {id(p) for mod in model.modules() if isinstance(mod, nn.LayerNorm) for p in mod.parameters(recurse=False)}
    return [name for name, p in model.named_parameters()
            if id(p) in ln_owned or any(pat.search(name.lower()) for pat in _NO_DECAY_PATTERNS)]


def no_decay_ranges(named_parameters, skip) -> torch.Tensor:
    """Sorted, merged flat ranges [lo, hi) (int64 [n, 2]) of the parameters in ``skip``, for the flat buffer laid out in
    ``named_parameters`` order."""
    skip, out, off = set(skip), [], 0
    for name, p in named_parameters:
        n = p.numel()
        if name in skip:
            if off % 4 or n % 4:
                raise ValueError(f"{name}: flat range [{off}, {off + n}) is not 4-aligned")
            if out and out[-1][1] == off:
                out[-1][1] = off + n
            else:
                out.append([off, off + n])
        off += n
    return torch.tensor(out, dtype=torch.int64).reshape(-1, 2)


# ----------------------------------------------------------------------------- data parallel
def init_distributed(backend: Optional[str] = None):
    """One process per GPU: under ``torchrun`` / ``accelerate launch`` (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the
    environment) bind this process to its GPU and create the process group -- ``nccl`` IS RCCL on ROCm -- BEFORE any other
    GPU work.  Returns (rank, local_rank, world); a no-op for a single process."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank, local = int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
    if world <= 1:
        return 0, 0, 1
    if os.environ.get("ADT_SHARE_GPU") == "1":           # debug / one-GPU test boxes only: every rank on GPU 0, gloo transport
        local, backend = 0, backend or "gloo"
        os.environ["LOCAL_RANK"] = "0"                   # accelerate / DDP take device_ids from it: cuda:1 does not exist on such a box
    if not dist.is_initialized():
        if torch.cuda.is_available():
            torch.cuda.set_device(local)
            be = backend or "nccl"
            dist.init_process_group(be, **({"device_id": torch.device("cuda", local)} if be == "nccl" else {}))
        else:
            dist.init_process_group(backend or "gloo")
    elif torch.cuda.is_available():
        torch.cuda.set_device(local)
    return rank, local, world


class GradReducer:
    """Data-parallel gradient averaging over one flat buffer.

    ``segment_ready(lo, hi)`` is called by the backward pass as soon as ``gflat[lo:hi]`` is final and
    starts an asynchronous all-reduce of that slice (RCCL on the GPUs, gloo in the CPU tests);
    ``finish()`` waits for all of them and turns the sums into means.  Segments must tile the buffer
    (the network's ``flat_range`` segments do); ``covered`` lets callers assert that.

    ``enabled = False`` turns a pass into DDP's ``no_sync()``: nothing is sent (gradient accumulation reduces only on the
    last micro-step).  ``addend``: a flat buffer added to each segment right before it is sent (the locally accumulated
    gradients of the earlier micro-steps), so the overlap with the backward pass survives accumulation.
    ``compress="bf16"`` sends bf16 copies of the segments (138 MB instead of 276 MB per step at setting-1; the mean is taken
    BEFORE the cast, as ``torch.distributed``'s ``bf16_compress_hook`` does) and widens them back in ``finish()``.
    ``timing=True`` brackets the waits of ``finish()`` with events on the compute stream: the time that stream sat behind
    the collectives (``exposed_wait_ms``) without a host synchronisation per step."""

    def __init__(self, gflat: torch.Tensor, process_group=None, compress: Optional[str] = None, timing: bool = False):
        if compress not in (None, "", "none", "bf16"):
            raise ValueError(f"compress must be None or 'bf16', not {compress!r}")
        self.gflat, self.pg = gflat, process_group
        self.world = dist.get_world_size(process_group)
        self._works, self.covered = [], 0
        self._segs = []
        # RCCL averages inside the collective; gloo (CPU tests, shared-GPU debug runs) only sums, the mean is then one more pass
        self.avg_in_collective = dist.get_backend(process_group) == "nccl"
        # One rank: the mean IS the sum, and RCCL runs an in-place single-rank SUM as nothing at all, whereas AVG there is a pre-multiply
        # copy of every segment (oneRankReduce<FuncPreMulSum>: 8 launches, 0.41 ms of HBM-bound kernel time per step under the backward
        # pass -- the whole +0.6 ms of `torchrun --nproc-per-node 1` over the plain run, profiles/r05/rccl_one_rank_ab.txt).
        self.op_avg = dist.ReduceOp.AVG if self.world > 1 else dist.ReduceOp.SUM
        self.enabled = True
        self.addend: Optional[torch.Tensor] = None
        self.compress = "bf16" if compress == "bf16" else None
        self.wire = torch.empty(gflat.numel(), dtype=torch.bfloat16, device=gflat.device) if self.compress else None
        self.timing = bool(timing) and gflat.is_cuda
        self._events = []
        self.bytes_last_step = 0
        self.steps = 0

    def segment_ready(self, lo: int, hi: int):
        if not self.enabled:
            return
        seg = self.gflat[lo:hi]
        if self.addend is not None:
            seg.add_(self.addend[lo:hi])
        if self.compress:
            buf = self.wire[lo:hi]
            if self.avg_in_collective:
                buf.copy_(seg)
                op = self.op_avg
            else:
                torch.mul(seg, 1.0 / self.world, out=seg)          # mean before the cast: the bf16 sum cannot overflow its range
                buf.copy_(seg)
                op = dist.ReduceOp.SUM
        else:
            buf, op = seg, (self.op_avg if self.avg_in_collective else dist.ReduceOp.SUM)
        self._works.append(dist.all_reduce(buf, op=op, group=self.pg, async_op=True))
        self._segs.append((lo, hi))
        self.covered += hi - lo

    def finish(self):
        if not self.enabled:
            return
        ev = None
        if self.timing:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        for w in self._works:
            w.wait()
        if ev is not None:
            ev[1].record()
            self._events.append(ev)
        self._works.clear()
        if self.covered != self.gflat.numel():
            raise RuntimeError(f"gradient segments covered {self.covered} of {self.gflat.numel()} elements")
        self.bytes_last_step = self.covered * (2 if self.compress else 4)
        self.covered = 0
        self.steps += 1
        if self.compress:
            for lo, hi in self._segs:
                self.gflat[lo:hi].copy_(self.wire[lo:hi])
        elif not self.avg_in_collective:
            self.gflat.mul_(1.0 / self.world)
        self._segs.clear()

    def comm_stats(self, reset: bool = True) -> dict:
        """``bytes_per_step`` (what one rank hands to the collective library per optimisation step) and, with ``timing``, the mean
        ``exposed_wait_ms`` over the steps since the last call (synchronises the device: call it outside timed regions)."""
        out = {"bytes_per_step": int(self.bytes_last_step), "wire_dtype": "bf16" if self.compress else "f32",
               "reduce_op": "avg" if self.avg_in_collective else "sum+scale", "exposed_wait_ms": None}
        if self._events:
            torch.cuda.synchronize()
            ms = [a.elapsed_time(b) for a, b in self._events]
            out["exposed_wait_ms"] = sum(ms) / len(ms)
            out["exposed_wait_ms_max"] = max(ms)
            out["steps_timed"] = len(ms)
            if reset:
                self._events.clear()
        return out


def install_engine_reduction(ddp_model, accumulation_steps: int = 1, timing: bool = False):
    """HF Trainer + ``DistributedDataParallel`` (the reference's ``accelerate launch train.py <yaml>``, README.md:53-57, train.py:305-319): let
    the engine's backward pass drive the gradient all-reduce.  DDP's buckets are filled when autograd delivers the gradients, and the
    autograd bridge delivers all 132 at once AFTER the whole hand-written backward pass -- 276 MB of all-reduce fully exposed every step.
    Here a ``GradReducer`` is hooked to the engine's gradient segments exactly as in the native loop (decoder block first, then each encoder
    layer: all but the last ~31 MB travel under the remaining backward kernels), the bridge returns the averaged gradients, and DDP gets a
    comm hook that passes a bucket through when the engine has already averaged this pass (and runs the ordinary all-reduce otherwise:
    inside ``no_sync()`` nothing is sent by either, and with gradient accumulation the engine leaves the reduction to DDP, which reduces the
    accumulated sum on the last micro-step).  Idempotent -- call it before every pass (``ADTTrainer.compute_loss`` does): it re-attaches the
    reducer when the model's engine object has been replaced; returns the reducer (None when not applicable)."""
    from torch.distributed.algorithms.ddp_comm_hooks import default_hooks
    from torch.nn.parallel import DistributedDataParallel as DDP
    if not isinstance(ddp_model, DDP) or not hasattr(ddp_model.module, "engine"):
        return None
    pg = ddp_model.process_group

    def attach(eng):
        """Hook a GradReducer to THIS engine object (``ADTModel._apply`` -- .to() / .float() -- and ``set_precision`` build a new one: the
        reducer, the sync predicate and the pass counters live on the engine, so a rebuilt engine starts without them)."""
        if getattr(eng, "_hf_hook_installed", False):
            return
        eng._hf_hook_installed = True
        if accumulation_steps <= 1:
            gflat, _ = eng.grad_buffers()
            red = GradReducer(gflat, pg, timing=timing)
            eng.hf_reducer = red
            eng.hf_sync = lambda: bool(getattr(eng, "hf_force_sync", False) or ddp_model.require_backward_grad_sync)
            eng.grad_ready_hook = red.segment_ready
        eng.hf_hook_stats = ddp_model._adt_hook_stats

    if not hasattr(ddp_model, "_adt_hook_stats"):
        # one comm hook per DDP wrapper; it looks the engine up on every call, so it never tests a replaced engine's frozen counters (which
        # would pass every bucket through unreduced and let the ranks diverge silently)
        stats = {"passed_through": 0, "reduced_by_ddp": 0}
        ddp_model._adt_hook_stats = stats

        def hook(state, bucket):
            eng = ddp_model.module.engine
            attach(eng)                                               # (a rebuilt engine: engine-driven from its next pass on)
            if getattr(eng, "hf_reducer", None) is not None and eng.reduced_generation == eng.generation:   # this pass's gradients arrived averaged
                stats["passed_through"] += 1
                fut = torch.futures.Future()
                fut.set_result(bucket.buffer())
                return fut
            stats["reduced_by_ddp"] += 1
            return default_hooks.allreduce_hook(pg, bucket)

        ddp_model.register_comm_hook(None, hook)
    eng = ddp_model.module.engine
    attach(eng)
    return eng.hf_reducer


def forward_engine_reduced(ddp_model, *args, **kwargs):
    """One forward pass through the DDP wrapper on the engine-driven path WITHOUT DDP's bucket traffic.  The engine has averaged the
    gradients by the time autograd hands them over, so DDP's reducer has nothing left to do -- but left in place it still copies every
    gradient into its buckets and back (264 device copies of 276 MB in all per step at setting-1: +1.1 ms, `profiles/r05/hf_ddp_copies.txt`)
    before the comm hook passes the buckets through.  DDP decides in its forward whether the coming backward is a synchronising one
    (`prepare_for_backward` is skipped inside ``no_sync()``), so the forward runs inside ``no_sync()`` with the engine told to reduce
    anyway.  Falls back to the plain call whenever the engine is not the one that reduces this pass (no reducer: gradient accumulation or a
    model without an engine; the caller's own ``no_sync()``), where DDP's own machinery is what is wanted."""
    from torch.nn.parallel import DistributedDataParallel as DDP
    eng = getattr(getattr(ddp_model, "module", None), "engine", None)
    if not isinstance(ddp_model, DDP) or eng is None or getattr(eng, "hf_reducer", None) is None or not ddp_model.require_backward_grad_sync:
        return ddp_model(*args, **kwargs)
    # Inside no_sync() DDP skips prepare_for_backward AND clears require_forward_param_sync, so module buffers are never re-broadcast
    # after the first pass, and static_graph's bookkeeping never sees a synchronising backward.  Both are harmless only under the
    # conditions train.py sets (ddp_broadcast_buffers=False; this model's buffers are constants anyway; no static graph) -- anything
    # else keeps DDP's own path, said once.
    n_buffers = sum(1 for _ in ddp_model.module.buffers())
    if (getattr(ddp_model, "broadcast_buffers", False) and n_buffers > 0) or getattr(ddp_model, "static_graph", False):
        if not getattr(ddp_model, "_adt_fast_path_note", False):
            ddp_model._adt_fast_path_note = True
            warnings.warn("adt_str_amd: DDP was built with broadcast_buffers=True (and the module has buffers) or static_graph=True; the "
                          "copy-free engine-reduced forward is not taken (DDP's own bucket path runs: correct, ~1 ms per step slower). "
                          "Pass ddp_broadcast_buffers=False / find_unused_parameters=False as train.py does.")
        return ddp_model(*args, **kwargs)
    eng.hf_force_sync = True
    try:
        with ddp_model.no_sync():
            return ddp_model(*args, **kwargs)
    finally:
        eng.hf_force_sync = False


def backward_segments(engine):
    """The flat ranges in the order the backward pass completes them (network._Engine._backward)."""
    segs = [engine.flat_range("decoder.")]
    for L in reversed(engine.enc):
        segs.append(engine.flat_range(L["p"] + "."))
    for pre in ("encoder.dense_layer.", "encoder.layer_norm.", "project_to_mel."):
        segs.append(engine.flat_range(pre))
    return segs


class FlatTrainer:
    """One optimisation step = ``grad_accum`` micro-steps of ``ADTTrainer.compute_loss`` (train.py:40-78) + backward, then
    (data-parallel mean) + global-norm clip + AdamW on the flat buffers, with HF's schedule and decay groups."""

    def __init__(self, model, lr: float = 1e-4, weight_decay: float = 1e-5, betas=(0.9, 0.999), eps: float = 1e-8,
                 max_grad_norm: float = 1.0, total_steps: int = 1000, warmup_ratio: float = 0.1, min_lr_ratio: float = 0.0,
                 process_group=None, scheduler: Optional[str] = None, grad_accum: int = 1, seed: Optional[int] = None,
                 grad_compress: Optional[str] = None, comm_timing: bool = False):
        self.model, self.eng = model, model.engine
        self.lr, self.wd, self.betas, self.eps, self.max_norm = lr, weight_decay, betas, eps, max_grad_norm
        self.total_steps = total_steps
        self.warmup = math.ceil(total_steps * warmup_ratio) if warmup_ratio < 1 else int(warmup_ratio)    # TrainingArguments.get_warmup_steps
        self.min_lr_ratio = min_lr_ratio
        self.scheduler = scheduler or ("cosine_warmup_with_min_lr" if min_lr_ratio > 0 else "cosine")
        self.grad_accum, self._micro = max(1, int(grad_accum)), 0
        self.step_no = 0
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_available() and dist.is_initialized() else 1
        self.rank = dist.get_rank(process_group) if self.world > 1 else 0
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
        self.gacc = torch.zeros_like(self.pflat) if self.grad_accum > 1 else None
        self.norm = torch.zeros(2, dtype=torch.float32, device=dev)
        self.nodecay = no_decay_ranges(named.items(), no_decay_names(model)).to(dev)
        if seed is not None:
            self.eng.seed_dropout(seed, self.rank)
        self.reducer = None
        # an explicit process group is honoured even at world size 1 (the collective path then runs end to end on one GPU:
        # how tests and ``torchrun --nproc-per-node 1 bench.py`` exercise the RCCL branch); otherwise only when there are peers
        if dist.is_available() and dist.is_initialized() and (self.world > 1 or process_group is not None):
            self.broadcast_parameters()
            self.reducer = GradReducer(self.gflat, self.pg, compress=grad_compress or os.environ.get("ADT_GRAD_COMPRESS"),
                                       timing=comm_timing)
            self.eng.grad_ready_hook = self.reducer.segment_ready
        self.eng.refresh_weights(force=True)

    # ---- data parallel -----------------------------------------------------------------
    def broadcast_parameters(self):
        """Rank 0's parameters to everyone, once (DDP constructor semantics); buffers are constants and stay local."""
        dist.broadcast(self.pflat, src=dist.get_global_rank(self.pg, 0) if self.pg is not None else 0, group=self.pg)

    # ---- one optimisation step ----------------------------------------------------------
    def current_lr(self) -> float:
        return self.lr * lr_multiplier(self.scheduler, self.step_no, self.total_steps, self.warmup, self.min_lr_ratio)

    def micro_step(self, wavs, tokens, token_lengths):
        """Forward + backward of one micro-batch; every ``grad_accum``-th call also applies the update.  Returns the loss
        (device scalar)."""
        self.model.train()
        tgt_in, labels = tokens[:, :-1], tokens[:, 1:]
        T = tgt_in.shape[1]
        pad = torch.arange(T, device=tokens.device).unsqueeze(0) >= token_lengths.to(tokens.device).unsqueeze(1)
        last = self._micro + 1 >= self.grad_accum
        if self.reducer is not None:
            # DDP ``no_sync()`` semantics: only the LAST micro-step of an accumulation window talks to the other ranks; its
            # segments carry the locally accumulated sum of the earlier ones (added per segment just before it is sent)
            self.reducer.enabled = last
            self.reducer.addend = self.gacc if (last and self.grad_accum > 1) else None
        out = self.eng.loss_and_grads(wavs, tgt_in, pad, labels, want_grads=True)
        if self.reducer is not None:
            self.reducer.finish()
        self._micro += 1
        if self.grad_accum > 1:
            if not last:
                self.gacc.add_(self.gflat)
                return out["loss"]
            if self.reducer is None:
                self.gflat.add_(self.gacc)
            self.gflat.mul_(1.0 / self.grad_accum)                         # HF scales each micro-batch loss by 1 / accumulation steps
            self.gacc.zero_()
        self._micro = 0
        K.grad_norm(self.gflat, self.max_norm, out=self.norm)
        lr = self.current_lr()
        self.step_no += 1
        K.adamw_step(self.pflat, self.gflat, self.m, self.v, self.step_no, lr, self.betas[0], self.betas[1], self.eps, self.wd,
                     self.norm, nodecay=self.nodecay)
        self.eng.refresh_weights(force=True)
        K.check_attn_bwd()          # a give-up of the one-kernel attention backward (incomplete dQ) in an earlier step ends the run here
        return out["loss"]

    def train_step(self, wavs, tokens, token_lengths):
        """One micro-step (= one optimisation step when ``grad_accum == 1``)."""
        return self.micro_step(wavs, tokens, token_lengths)

    # ---- checkpoint state ------------------------------------------------------------------
    def state_dict(self) -> dict:
        return dict(self.state_dict_meta(), m=self.m.detach().cpu(), v=self.v.detach().cpu())

    def state_dict_meta(self) -> dict:
        # ``drop_steps``: training passes drawn so far -- the rank-independent part of the dropout counter (``drop_seed`` itself is
        # seeded per rank, so rank 0's value must not be handed to the other ranks on resume)
        return {"step_no": self.step_no, "drop_seed": self.eng.drop_seed,
                "drop_steps": self.eng.drop_seed - self.eng.drop_base, "total_steps": self.total_steps, "scheduler": self.scheduler,
                "grad_accum": self.grad_accum, "warmup": self.warmup}

    def load_state_dict(self, sd: dict):
        self.m.copy_(sd["m"]); self.v.copy_(sd["v"])
        self.step_no = int(sd["step_no"])
        if "drop_steps" in sd:
            self.eng.drop_seed = self.eng.drop_base + int(sd["drop_steps"])      # this rank's own sequence, resumed
        else:                                                                    # checkpoints written before drop_steps existed
            self.eng.drop_seed = int(sd["drop_seed"])
        for key, mine in (("scheduler", self.scheduler), ("grad_accum", self.grad_accum)):
            if key in sd and sd[key] != mine:
                raise ValueError(f"checkpoint was written with {key}={sd[key]!r}, this run has {mine!r}: the learning-rate curve and the "
                                 "data order would silently differ from the interrupted run")
        if "total_steps" in sd and sd["total_steps"] != self.total_steps:        # a longer / shorter schedule can be deliberate
            warnings.warn(f"checkpoint was written for {sd['total_steps']} optimisation steps, this run plans {self.total_steps}: "
                          "the learning-rate curve differs from the interrupted run's")
        self._micro = 0
        if self.gacc is not None:
            self.gacc.zero_()
        self.eng.refresh_weights(force=True)


# ----------------------------------------------------------------------------- checkpoints (HF layout: checkpoint-<step>/)
def output_path(cfg: dict) -> str:
    """``logging.output_dir / experiment.run_name`` -- where the reference puts checkpoints and the final model
    (train.py:171-176), so that two experiments sharing an ``output_dir`` never resume from each other."""
    lg, ex = cfg.get("logging", {}) or {}, cfg.get("experiment", {}) or {}
    return os.path.join(lg.get("output_dir") or "./outputs", str(ex.get("run_name") or "default"))


def _checkpoint_dirs(output_dir: str):
    found = []
    for d in glob.glob(os.path.join(output_dir, "checkpoint-*")):
        m = re.fullmatch(r"checkpoint-(\d+)", os.path.basename(d))
        if m and os.path.exists(os.path.join(d, "trainer_state.pt")):
            found.append((int(m.group(1)), d))
    return [d for _, d in sorted(found)]


def latest_checkpoint(output_dir: str) -> Optional[str]:
    dirs = _checkpoint_dirs(output_dir)
    return dirs[-1] if dirs else None


def _write_model_files(state: dict, cfg, directory: str):
    from safetensors.torch import save_file
    os.makedirs(directory, exist_ok=True)
    save_file(state, os.path.join(directory, "model.safetensors"))
    if cfg is not None and hasattr(cfg, "save_pretrained"):
        cfg.save_pretrained(directory)


def save_model(model: nn.Module, directory: str):
    """``trainer.save_model()`` (train.py:323): ``model.safetensors`` with the reference's state-dict keys (+ ``config.json`` when
    the model has an HF config) -- what ``build_model.py`` loads."""
    _write_model_files({k: v.detach().to("cpu").contiguous().clone() for k, v in model.state_dict().items()},
                       getattr(model, "config", None), directory)


INCOMPLETE_SENTINEL = ".adt_incomplete"
FOREIGN_CHECKPOINT_FILES = ("trainer_state.json", "optimizer.pt", "scheduler.pt", "training_args.bin", "pytorch_model.bin")


def _is_own_incomplete(d: str) -> bool:
    """A directory THIS loop created and never finished: it still carries the sentinel ``save_checkpoint`` drops at makedirs time
    (removed when ``trainer_state.pt`` is written) and holds nothing an HF / reference checkpoint would (``trainer_state.json``,
    ``optimizer.pt`` ...).  Anything else in ``output_dir`` -- the HF-Trainer path writes its ``checkpoint-N`` directories into the
    same place -- is not ours to delete."""
    if not os.path.exists(os.path.join(d, INCOMPLETE_SENTINEL)) or os.path.exists(os.path.join(d, "trainer_state.pt")):
        return False
    return not any(os.path.exists(os.path.join(d, f)) for f in FOREIGN_CHECKPOINT_FILES)


def _prune_checkpoints(output_dir: str, keep: Optional[int], just_written: str):
    """``save_total_limit``: remove the oldest NATIVE checkpoints beyond ``keep`` -- never the one just written, whatever stale
    higher-numbered directories a previous run left behind, and never a directory this loop did not provably create (HF Trainer /
    reference checkpoints share ``output_dir``: they have ``trainer_state.json``, not ``.pt``).  Without ``keep`` nothing is removed."""
    if not keep:
        return
    just = os.path.abspath(just_written)
    # directories a killed run of THIS loop left without their marker are invisible to _checkpoint_dirs and would never be rotated out:
    # remove the ones OLDER (lower step) than the checkpoint just completed (a higher-numbered one may be the next checkpoint, which
    # other ranks are already writing their RNG files into while this one's background write finishes)
    mj = re.fullmatch(r"checkpoint-(\d+)", os.path.basename(just))
    for d in glob.glob(os.path.join(output_dir, "checkpoint-*")):
        m = re.fullmatch(r"checkpoint-(\d+)", os.path.basename(d))
        if m and mj and int(m.group(1)) < int(mj.group(1)) and _is_own_incomplete(d):
            shutil.rmtree(d, ignore_errors=True)
    others = [d for d in _checkpoint_dirs(output_dir) if os.path.abspath(d) != just]
    others.sort(key=lambda d: os.path.getmtime(os.path.join(d, "trainer_state.pt")))
    for old in others[:max(0, len(others) - (keep - 1))]:
        shutil.rmtree(old, ignore_errors=True)


class CheckpointWriter:
    """Writes checkpoints off the training thread.  ``save_checkpoint`` takes a DEVICE-side snapshot (three flat clones, stream
    ordered, ~0.1 ms) and hands it over; this thread copies it to pinned host memory on its own HIP stream (so the 0.8 GB of
    device -> host traffic never sits in front of a training kernel) and writes the files.  One write in flight: a second
    request first waits for the previous one (back-pressure instead of unbounded host memory).  ``close()`` joins; an error
    in the writer surfaces on the next ``submit`` / ``close``."""

    def __init__(self):
        self._thread: Optional[threading.Thread] = None
        self._error: Optional[BaseException] = None
        self._stream = None

    def _run(self, job):
        try:
            job()
        except BaseException as e:                       # surfaced on the training thread
            self._error = e

    def wait(self):
        if self._thread is not None:
            self._thread.join()
            self._thread = None
        if self._error is not None:
            e, self._error = self._error, None
            raise RuntimeError("background checkpoint write failed") from e

    def submit(self, job: Callable[[], None]):
        self.wait()
        self._thread = threading.Thread(target=self._run, args=(job,), name="adt-checkpoint-writer", daemon=True)
        self._thread.start()

    def side_stream(self, device):
        if self._stream is None:
            self._stream = torch.cuda.Stream(device=device)
        return self._stream

    close = wait


def _device_snapshot(model: nn.Module, trainer):
    """(state-dict snapshot on the device, optimizer moments on the device): parameters as views of ONE clone of the flat
    buffer, buffers (constants) cloned one by one."""
    snap = {}
    pflat = getattr(trainer, "pflat", None)
    if pflat is not None:
        flat = pflat.detach().clone()
        off = 0
        for name, p in trainer.eng.named.items():
            snap[name] = flat[off:off + p.numel()].view(p.shape)
            off += p.numel()
    for k, v in model.state_dict().items():
        if k not in snap:
            snap[k] = v.detach().clone()
    return snap, trainer.m.detach().clone(), trainer.v.detach().clone()


def save_checkpoint(output_dir: str, model: nn.Module, trainer, progress: dict, rng_state, rank: int = 0, world: int = 1,
                    keep: Optional[int] = None, writer: Optional[CheckpointWriter] = None) -> str:
    """Rank 0 writes weights + optimizer state + progress; every rank writes its own host RNG state (the data stream of a rank
    is a function of it).  ``keep``: ``save_total_limit`` -- older checkpoints are removed (by rank 0, after its own files are
    complete; ``trainer_state.pt`` is written last and marks a checkpoint as complete).  With a ``writer`` the device -> host
    copy and the file writes happen on its thread; without one, here."""
    d = os.path.join(output_dir, f"checkpoint-{trainer.step_no}")
    os.makedirs(d, exist_ok=True)
    if not os.path.exists(os.path.join(d, "trainer_state.pt")):
        open(os.path.join(d, INCOMPLETE_SENTINEL), "a").close()   # "this loop made it": every rank, BEFORE its RNG file (rank 0 removes it with the marker)
    if rng_state is not None:
        rp = os.path.join(d, f"rng_state_{rank}.pth")
        torch.save({"python": rng_state[0], "torch": rng_state[1]}, rp + ".tmp")
        os.replace(rp + ".tmp", rp)                         # a file that exists is complete: rank 0 waits for all of them before the marker
    if rank != 0:
        return d
    rng_files = world if rng_state is not None else 0
    cfg = getattr(model, "config", None)
    flat = getattr(trainer, "pflat", None)
    if writer is None or flat is None or not flat.is_cuda or not hasattr(trainer, "state_dict_meta"):
        state = {k: v.detach().to("cpu").contiguous().clone() for k, v in model.state_dict().items()}
        _finish_checkpoint(d, state, cfg, trainer.state_dict(), progress, world, output_dir, keep, rng_files)
        return d
    meta = trainer.state_dict_meta()
    snap, m_dev, v_dev = _device_snapshot(model, trainer)
    ready = torch.cuda.Event()
    ready.record()                                          # the clones are stream ordered behind the step that produced them
    side = writer.side_stream(trainer.pflat.device)
    progress = dict(progress)

    def job():
        with torch.cuda.stream(side):
            side.wait_event(ready)
            host = {k: torch.empty(v.shape, dtype=v.dtype, pin_memory=True).copy_(v, non_blocking=True) for k, v in snap.items()}
            m_h = torch.empty(m_dev.shape, dtype=m_dev.dtype, pin_memory=True).copy_(m_dev, non_blocking=True)
            v_h = torch.empty(v_dev.shape, dtype=v_dev.dtype, pin_memory=True).copy_(v_dev, non_blocking=True)
            side.synchronize()
        _finish_checkpoint(d, {k: t.contiguous() for k, t in host.items()}, cfg, dict(meta, m=m_h, v=v_h), progress, world, output_dir, keep, rng_files)

    writer.submit(job)
    return d


RNG_FILE_WAIT_S = 120.0


def _finish_checkpoint(d, state, cfg, tstate, progress, world, output_dir, keep, rng_files=0):
    _write_model_files(state, cfg, d)
    # ``trainer_state.pt`` marks the checkpoint complete, so it is only written once every rank's host RNG file is there (the ranks write
    # them on their own, without a barrier: a rank that died or lags must not leave a "complete" checkpoint that resumes on another data stream)
    deadline = time.monotonic() + RNG_FILE_WAIT_S
    while True:
        absent = [r for r in range(rng_files) if not os.path.exists(os.path.join(d, f"rng_state_{r}.pth"))]
        if not absent:
            break
        if time.monotonic() > deadline:
            raise RuntimeError(f"checkpoint {d}: rng_state files of ranks {absent} did not appear within {RNG_FILE_WAIT_S:.0f} s; the checkpoint "
                               "is left without trainer_state.pt (incomplete) and will not be resumed from")
        time.sleep(0.02)
    tmp = os.path.join(d, "trainer_state.pt.tmp")
    torch.save({"trainer": tstate, "progress": dict(progress), "world": world, "rng_files": rng_files}, tmp)
    os.replace(tmp, os.path.join(d, "trainer_state.pt"))
    try:
        os.remove(os.path.join(d, INCOMPLETE_SENTINEL))
    except FileNotFoundError:
        pass
    _prune_checkpoints(output_dir, keep, d)


def load_checkpoint(directory: str, model: nn.Module, trainer, rank: int = 0, world: int = 1, allow_missing_rng: Optional[bool] = None) -> dict:
    """Restore weights (in place: the flat parameter buffer keeps its views), optimizer state and this rank's host RNG state.
    Returns the saved progress dict.  A checkpoint that recorded RNG files for every rank refuses to resume a rank whose file is gone
    (the run would silently leave the interrupted run's data stream) unless ``allow_missing_rng`` / ``ADT_ALLOW_MISSING_RNG=1``."""
    from safetensors.torch import load_file
    st = torch.load(os.path.join(directory, "trainer_state.pt"), map_location="cpu", weights_only=False)
    if st.get("world", 1) != world:
        raise ValueError(f"checkpoint was written by {st.get('world', 1)} ranks, this run has {world}: the data sharding would differ")
    weights = load_file(os.path.join(directory, "model.safetensors"))
    own = dict(model.state_dict())
    with torch.no_grad():
        for k, v in weights.items():
            if k in own:
                own[k].copy_(v)
        missing = [k for k, _ in model.named_parameters() if k not in weights]
    if missing:
        raise KeyError(f"checkpoint {directory} lacks parameters: {missing[:4]}...")
    trainer.load_state_dict(st["trainer"])
    rp = os.path.join(directory, f"rng_state_{rank}.pth")
    if os.path.exists(rp):
        r = torch.load(rp, map_location="cpu", weights_only=False)
        random.setstate(r["python"])
        torch.set_rng_state(r["torch"])
    else:
        if allow_missing_rng is None:
            allow_missing_rng = os.environ.get("ADT_ALLOW_MISSING_RNG") == "1"
        if st.get("rng_files", 0) > rank and not allow_missing_rng:
            raise FileNotFoundError(f"{rp} is missing although the checkpoint recorded one for each of its {st['rng_files']} ranks: rank {rank} "
                                    "would resume on a different data stream (set ADT_ALLOW_MISSING_RNG=1 to accept that)")
        warnings.warn(f"{rp} is missing: rank {rank} resumes with a fresh host RNG state, so its data stream (random velocities, "
                      "timbre / mix-up / FX draws) will not repeat the interrupted run's")
    return st["progress"]


class _LossLog:
    """``logging_steps`` without a host stall: the loss is copied to pinned memory asynchronously and printed once it has landed.  The line
    also carries the step time and the whole-job clips/s since the previously logged step (the reference logs through HF's
    ``logging_steps: 1``, train.py:144-152; throughput is this build's addition), measured between timing events on the step's stream on
    the GPU (host clock for CPU tensors) -- never by synchronising."""

    def __init__(self, every: int, rank: int, total: int, sink: Callable[[str], None] = print, clips_per_step: int = 0):
        self.every, self.rank, self.total, self.sink, self.clips_per_step = every, rank, total, sink, clips_per_step
        self.pending = collections.deque()
        self.history = []                       # (step, loss) as printed
        self.rates = []                         # (step, ms per step, clips/s) as printed
        self._prev = None                       # (step, event or host seconds) of the previously logged step

    def push(self, step: int, epoch: int, loss: torch.Tensor, lr: float):
        if not self.every or self.rank != 0 or step % self.every:
            return
        if loss.is_cuda:
            host = torch.empty((), dtype=torch.float32, pin_memory=True)
            host.copy_(loss.detach().reshape(()), non_blocking=True)
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            mark = ev
        else:
            host, ev, mark = loss.detach().reshape(()).float().clone(), None, time.perf_counter()
        self.pending.append((step, epoch, host, lr, ev, self._prev, mark))
        self._prev = (step, mark)
        self.drain(False)

    def drain(self, block: bool = True):
        while self.pending and (block or self.pending[0][4] is None or self.pending[0][4].query()):
            step, epoch, host, lr, ev, prev, mark = self.pending.popleft()
            if ev is not None:
                ev.synchronize()
            self.history.append((step, float(host)))
            rate = ""
            if prev is not None and step > prev[0] and type(prev[1]) is type(mark):
                ms = (prev[1].elapsed_time(mark) if ev is not None else (mark - prev[1]) * 1e3) / (step - prev[0])
                if ms > 0:
                    cps = self.clips_per_step / (ms * 1e-3)
                    self.rates.append((step, ms, cps))
                    rate = f" step_ms {ms:.2f} clips/s {cps:.1f}"
            self.sink(f"epoch {epoch} step {step}/{self.total} loss {float(host):.4f} lr {lr:.3e}{rate}")


def run_native_training(model, dataset, cfg: dict, trainer_factory: Optional[Callable] = None, prefetch_depth: int = 2):
    """Epoch loop over a ``NoteChunkDataset`` with ``FlatTrainer`` -- the native counterpart of ``Trainer.train()`` +
    ``trainer.save_model()`` (train.py:319-323).  Call ``init_distributed()`` first in a multi-process launch: every rank walks
    the same per-epoch permutation and takes its stride of it (``DistributedSampler`` semantics).  Honours
    ``gradient_accumulation_steps``, ``save_every_n_steps``, ``max_checkpoints``, ``resume_from_checkpoint`` / ``auto_resume``."""
    t, lg, ck = cfg["training"], cfg["logging"], cfg.get("checkpoint", {}) or {}
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    bs, accum = t["batch_size"], max(1, int(t.get("gradient_accumulation_steps") or 1))
    epochs = t["num_epochs"] or 1
    micro_per_epoch = (len(dataset) // (bs * world * accum)) * accum
    steps_per_epoch = micro_per_epoch // accum
    total = steps_per_epoch * epochs
    if total == 0:
        raise ValueError(f"dataset of {len(dataset)} chunks is smaller than one global batch ({bs} x {world} ranks x {accum})")
    lr_type = t.get("lr_scheduler_type") or "cosine"
    min_lr = float(t.get("min_learning_rate") or 0.0)
    # the reference swaps in the min-LR cosine only for ``cosine`` with a positive floor (train.py:203-218); any other type keeps HF's own curve
    min_ratio = (min_lr / t["learning_rate"]) if (lr_type == "cosine" and min_lr > 0) else 0.0
    sched = "cosine_warmup_with_min_lr" if min_ratio > 0 else lr_type
    seed = cfg["experiment"]["seed"]
    factory = trainer_factory or FlatTrainer
    tr = factory(model, lr=t["learning_rate"], weight_decay=t["weight_decay"], max_grad_norm=t["max_grad_norm"], total_steps=total,
                 warmup_ratio=t["warmup_ratio"], min_lr_ratio=min_ratio, scheduler=sched, grad_accum=accum, seed=seed)
    out_dir = output_path(cfg)                              # output_dir / run_name, as the reference (train.py:171-176)
    writer = None if os.environ.get("ADT_SYNC_CHECKPOINTS") == "1" else CheckpointWriter()
    save_every, keep = lg.get("save_every_n_steps"), ck.get("max_checkpoints")
    start_epoch, start_micro = 0, 0
    resume = ck.get("resume_from_checkpoint")
    if resume is True or (not resume and ck.get("auto_resume")):
        resume = latest_checkpoint(out_dir)            # the reference globs a name HF never writes (train.py:186); this finds checkpoint-<step>
    if resume:
        prog = load_checkpoint(resume, model, tr, rank, world)
        start_epoch, start_micro = int(prog["epoch"]), int(prog["micro"])
        if rank == 0:
            print(f"resumed from {resume}: step {tr.step_no}, epoch {start_epoch}, micro-batch {start_micro}", flush=True)
    log = _LossLog(lg.get("logging_steps") or 0, rank, total, lambda s: print(s, flush=True), clips_per_step=bs * accum * world)
    order = list(range(len(dataset)))

    def run_epoch(epoch: int, first_micro: int):
        perm = list(order)
        random.Random(seed + epoch).shuffle(perm)                          # same permutation on every rank
        shard = perm[rank::world]
        pf = Prefetcher(lambda s: dataset.host_batch(shard[s * bs:(s + 1) * bs], snapshot_rng=bool(save_every)), micro_per_epoch,
                        start=first_micro, depth=prefetch_depth)
        try:
            for s, hb in zip(range(first_micro, micro_per_epoch), pf):
                batch = dataset.batcher.upload(hb, device_tokens=True)
                loss = tr.micro_step(batch["wavs"], batch["tokens"], batch["token_lengths"])
                if (s + 1) % accum:
                    continue
                log.push(tr.step_no, epoch, loss, tr.current_lr())
                if save_every and tr.step_no % save_every == 0:
                    nxt = (epoch, s + 1) if s + 1 < micro_per_epoch else (epoch + 1, 0)
                    save_checkpoint(out_dir, model, tr, {"epoch": nxt[0], "micro": nxt[1]}, hb.rng_state, rank, world, keep, writer)
        finally:
            pf.close()

    try:
        for epoch in range(start_epoch, epochs):
            run_epoch(epoch, start_micro)
            start_micro = 0
    finally:
        if writer is not None:
            writer.close()                                 # the last checkpoint is on disk before anyone can ask for it
    log.drain(True)
    if rank == 0:
        save_model(model, out_dir)
    if world > 1:
        dist.barrier()
    tr.loss_history = log.history
    tr.rate_history = log.rates
    return tr
