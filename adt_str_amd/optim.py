"""``torch.optim.AdamW`` for the HF-Trainer path, on ONE flat buffer.

The reference trains through ``transformers.Trainer`` (train.py:305-319), whose default optimizer is ``torch.optim.AdamW`` over two
parameter groups (decay / no decay, ``Trainer.get_decay_parameter_names``).  On this model that is 132 tensors and, as multi-tensor
("foreach") kernels, ~1.2 ms per step; the native loop's fused kernel (``adt_adamw_step``: parameters, gradients and both moments as flat
fp32 buffers, the no-decay set as sorted ranges) does the same arithmetic in 0.34 ms.  ``FusedAdamW`` gives the HF path that kernel
without leaving the ``torch.optim.Optimizer`` contract: it takes the Trainer's own parameter groups, re-points every parameter at a view of
one flat buffer (names, shapes and ``state_dict`` keys untouched), keeps ``state[p] = {step, exp_avg, exp_avg_sq}`` as views of the flat
moments (so ``state_dict`` / ``load_state_dict`` and HF checkpoints work), reads the learning rate the scheduler writes into the groups,
and leaves gradient clipping to the Trainer (``accelerator.clip_grad_norm_`` scales ``p.grad`` in place before ``step``).

The gradients: the autograd bridge (network._ADTLossFn.backward) hands every parameter a view of ONE fresh buffer in parameter order;
accumulation, clipping and DDP's copy-back all work in place on those views, so ``step`` normally finds them still laid out as one flat
buffer (checked by address: autograd keeps ``grad.detach()``, which has no ``_base``) and launches on it directly.  Anything else (a
gradient that is foreign or re-allocated) is gathered into a scratch buffer first -- correct, one extra pass.
"""
from __future__ import annotations

from typing import Iterable, Optional

import torch

from . import kernels as K


class FusedAdamW(torch.optim.Optimizer):
    def __init__(self, param_groups: Iterable[dict], lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 1e-2,
                 engine=None):
        defaults = dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay)
        super().__init__(param_groups, defaults)
        b0, e0 = self.param_groups[0]["betas"], self.param_groups[0]["eps"]
        wds = sorted({float(g["weight_decay"]) for g in self.param_groups})
        if any(tuple(g["betas"]) != tuple(b0) or g["eps"] != e0 for g in self.param_groups) or len([w for w in wds if w != 0.0]) > 1:
            raise ValueError("FusedAdamW: the groups must share betas / eps and use at most one non-zero weight_decay (HF Trainer's two groups do)")
        self._wd = max(wds)
        self._engine = engine
        params = [p for g in self.param_groups for p in g["params"]]
        if engine is not None:                                            # the engine's own order: its gradient buffer is laid out that way
            order = {id(p): i for i, p in enumerate(engine.named.values())}
            if set(order) != {id(p) for p in params}:
                raise ValueError("FusedAdamW: the parameter groups must cover exactly the model's parameters")
            params.sort(key=lambda p: order[id(p)])
        if any(p.dtype != torch.float32 or not p.is_cuda for p in params):
            raise ValueError("FusedAdamW: fp32 parameters on the GPU")
        self._params = params
        dev = params[0].device
        n = sum(p.numel() for p in params)
        self._flat = torch.empty(n, dtype=torch.float32, device=dev)
        self._off, off = [], 0
        for p in params:                                                  # flatten: every parameter becomes a view of one buffer
            self._flat[off:off + p.numel()].copy_(p.data.reshape(-1))
            p.data = self._flat[off:off + p.numel()].view_as(p)
            self._off.append(off)
            off += p.numel()
        self._m, self._v = torch.zeros_like(self._flat), torch.zeros_like(self._flat)
        self._gather: Optional[torch.Tensor] = None
        nodecay = {id(p) for g in self.param_groups if float(g["weight_decay"]) == 0.0 for p in g["params"]} if self._wd != 0.0 else set()
        ranges = []
        for p, o in zip(params, self._off):
            if id(p) in nodecay:
                if o % 4 or p.numel() % 4:
                    raise ValueError("FusedAdamW: a no-decay parameter's flat range is not 4-aligned")
                if ranges and ranges[-1][1] == o:
                    ranges[-1][1] = o + p.numel()
                else:
                    ranges.append([o, o + p.numel()])
        self._nodecay = torch.tensor(ranges, dtype=torch.int64).reshape(-1, 2).to(dev) if ranges else None
        self._step = 0
        self._bind_state()
        if engine is not None:
            engine._versions = None                                      # the bf16 operands are re-derived from the new storage

    def _bind_state(self):
        step = torch.tensor(float(self._step))
        for p, o in zip(self._params, self._off):
            self.state[p] = {"step": step.clone(), "exp_avg": self._m[o:o + p.numel()].view_as(p), "exp_avg_sq": self._v[o:o + p.numel()].view_as(p)}

    def _flat_grad(self) -> torch.Tensor:
        """The gradients as one flat buffer in parameter order.  The autograd bridge hands out views of one fresh buffer, but autograd
        stores ``grad.detach()`` (no ``_base``), so the buffer is recognised by ADDRESS: when every ``p.grad`` is a contiguous fp32 tensor
        in the same storage at ``first + 4 * offset`` the flat tensor is rebuilt over that storage and the kernel runs on it directly (no
        copy, no second buffer: ``self._gather`` stays None).  Anything else is gathered (missing gradients are an error: torch would
        skip the update, so a model with unused parameters should use torch's optimizer)."""
        g0 = self._params[0].grad
        n = self._flat.numel()
        if g0 is not None and g0.dtype == torch.float32 and g0.is_contiguous() and g0.device == self._flat.device:
            st, ptr = g0.untyped_storage(), g0.data_ptr()
            if st.nbytes() >= 4 * (g0.storage_offset() + n):
                sp, ok = st.data_ptr(), True
                for p, o in zip(self._params, self._off):
                    g = p.grad
                    if (g is None or g.dtype != torch.float32 or g.data_ptr() != ptr + 4 * o or not g.is_contiguous()
                            or g.untyped_storage().data_ptr() != sp):
                        ok = False
                        break
                if ok:
                    return torch.empty(0, dtype=torch.float32, device=g0.device).set_(st, g0.storage_offset(), (n,), (1,))
        if any(p.grad is None for p in self._params):
            raise RuntimeError("FusedAdamW: every parameter needs a gradient (the ADT engine produces all of them in one backward pass)")
        if self._gather is None:
            self._gather = torch.empty_like(self._flat)
        torch._foreach_copy_([self._gather[o:o + p.numel()].view_as(p) for p, o in zip(self._params, self._off)], [p.grad for p in self._params])
        return self._gather

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        g = self._flat_grad()
        self._step += 1
        grp = self.param_groups[0]
        K.adamw_step(self._flat, g, self._m, self._v, self._step, float(grp["lr"]), beta1=grp["betas"][0], beta2=grp["betas"][1], eps=grp["eps"],
                     weight_decay=self._wd, nodecay=self._nodecay)
        for st in self.state.values():
            st["step"] += 1
        if self._engine is not None:
            self._engine._versions = None                                # parameters changed behind torch's version counters
        return loss

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)                               # fills self.state with NEW tensors: copy them into the flat moments
        steps = set()
        for p, o in zip(self._params, self._off):
            st = self.state.get(p)
            if st:
                self._m[o:o + p.numel()].copy_(st["exp_avg"].reshape(-1))
                self._v[o:o + p.numel()].copy_(st["exp_avg_sq"].reshape(-1))
                steps.add(int(float(st["step"])))
        if len(steps) > 1:
            raise ValueError("FusedAdamW: the loaded state has different step counts per parameter")
        self._step = steps.pop() if steps else 0
        self._bind_state()
