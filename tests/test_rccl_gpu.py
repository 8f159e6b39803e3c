"""The N > 1 code paths on the REAL collective library -- torch's ``nccl`` backend IS RCCL on ROCm -- at world size 1, which is
what a one-GPU box can host: ``GradReducer`` (the ``ReduceOp.AVG`` branch that only exists on RCCL) over the real flat gradient
buffer, a ``FlatTrainer`` step forced through the reducer against the plain step (bitwise), gradient accumulation with the
reduce on the last micro-step only, bf16-compressed segments, and the event-bracketed wait measurement.  Runs in a spawned
child so that the process group never leaks into the test runner."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    try:
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        from tests.test_ddp_gpu import _batch, _make
        from adt_str_amd.trainer import FlatTrainer, GradReducer, backward_segments
        res = {"backend": dist.get_backend()}

        # (1) the reducer alone on the real flat buffer: AVG inside the collective, values unchanged at world size 1
        model = _make(seed=5)
        eng = model.engine
        gflat, _ = eng.grad_buffers()
        gflat.copy_(torch.randn(gflat.numel(), device="cuda"))
        want = gflat.clone()
        red = GradReducer(gflat, dist.group.WORLD, timing=True)
        assert red.avg_in_collective
        for lo, hi in backward_segments(eng):
            red.segment_ready(lo, hi)
        red.finish()
        torch.cuda.synchronize()
        res["reducer_equal"] = bool(torch.equal(gflat, want))
        st = red.comm_stats()
        res["bytes"], res["wait_ms"] = st["bytes_per_step"], st["exposed_wait_ms"]
        res["n"] = gflat.numel()

        # (2) three optimisation steps with dropout: plain trainer vs. the same trainer forced through RCCL -- bitwise
        def run(pg, accum=1, compress=None, steps=3):
            m = _make(seed=11)
            m.config.dropout = 0.1
            tr = FlatTrainer(m, lr=1e-3, weight_decay=1e-2, max_grad_norm=1.0, total_steps=10, warmup_ratio=0.0, process_group=pg,
                             grad_accum=accum, seed=42, grad_compress=compress)
            losses = []
            for s in range(steps * accum):
                wav, tok, tl = _batch(seed=20 + s)
                losses.append(float(tr.micro_step(wav, tok, tl)))
            torch.cuda.synchronize()
            return tr, losses
        plain, l0 = run(None)
        forced, l1 = run(dist.group.WORLD)
        res["plain_has_no_reducer"] = plain.reducer is None
        res["forced_has_reducer"] = forced.reducer is not None and forced.reducer.steps == 3
        res["step_bitwise"] = bool(torch.equal(plain.pflat, forced.pflat)) and l0 == l1 and bool(torch.equal(plain.m, forced.m))
        # (3) accumulation: the collective runs once per optimisation step, on the last micro-step
        pa, la = run(None, accum=2)
        fa, lb = run(dist.group.WORLD, accum=2)
        res["accum_reduces_once_per_step"] = fa.reducer.steps == 3 and fa.step_no == 3
        res["accum_bitwise"] = bool(torch.equal(pa.pflat, fa.pflat)) and la == lb
        # (4) bf16 wire format: half the bytes, parameters within a bf16 rounding of the gradient's effect (Adam's normalised
        #     step is lr-bounded, so compare the updates)
        fc, _ = run(dist.group.WORLD, compress="bf16")
        res["compress_bytes"] = fc.reducer.comm_stats()["bytes_per_step"]
        res["compress_max_dev"] = float((fc.pflat - plain.pflat).abs().max())
        # (5) the HF path under DistributedDataParallel (world size 1 on RCCL): the engine's backward pass drives the all-reduce, DDP's
        #     buckets pass the gradients through; the gradients autograd ends up with are bitwise the plain bridge's
        from torch.nn.parallel import DistributedDataParallel as DDP
        from adt_str_amd.masks import create_mask_plain
        from adt_str_amd.trainer import forward_engine_reduced, install_engine_reduction
        wav, tok, tl = _batch(seed=31)
        _, pad = create_mask_plain(tok.shape[1] - 1, tl, wav.device)

        def grads(wrap, fast=False):
            m = _make(seed=13).train()
            net = DDP(m, device_ids=[0], broadcast_buffers=False) if wrap else m
            red = install_engine_reduction(net) if wrap else None
            kw = dict(src=wav, tgt=tok[:, :-1], tgt_mask=None, tgt_padding_mask=pad, labels=tok[:, 1:])
            (forward_engine_reduced(net, **kw) if fast else net(**kw)).backward()
            torch.cuda.synchronize()
            return torch.cat([p.grad.reshape(-1) for p in m.engine.named.values()]), red, m.engine

        g_plain, _, _ = grads(False)
        g_ddp, red_hf, eng_hf = grads(True)
        res["hf_ddp_bitwise"] = bool(torch.equal(g_plain, g_ddp))
        res["hf_reducer_ran"] = red_hf is not None and red_hf.steps == 1 and red_hf.bytes_last_step == 4 * g_ddp.numel()
        res["hf_ddp_passed_through"] = eng_hf.hf_hook_stats["passed_through"] >= 1 and eng_hf.hf_hook_stats["reduced_by_ddp"] == 0
        # ... and what ADTTrainer.compute_loss calls (the forward inside DDP.no_sync(), the engine told to reduce): the same bits, the
        # reducer ran, DDP's comm hook was not even asked (no bucket copies)
        g_fast, red_fast, eng_fast = grads(True, fast=True)
        res["hf_fast_bitwise"] = bool(torch.equal(g_plain, g_fast))
        res["hf_fast_reducer_ran"] = red_fast is not None and red_fast.steps == 1 and red_fast.bytes_last_step == 4 * g_fast.numel()
        res["hf_fast_hook_silent"] = eng_fast.hf_hook_stats["passed_through"] == 0 and eng_fast.hf_hook_stats["reduced_by_ddp"] == 0
        q.put(("ok", res))
    except Exception as e:                                 # pragma: no cover
        import traceback
        q.put(("error", repr(e) + "\n" + traceback.format_exc()))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def test_reducer_and_trainer_on_rccl_world_size_one():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_worker, args=(_free_port(), q))
    p.start()
    status, res = q.get(timeout=600)
    p.join(timeout=120)
    assert status == "ok", res
    assert res["backend"] == "nccl"
    assert res["reducer_equal"] and res["bytes"] == 4 * res["n"] and res["wait_ms"] is not None and res["wait_ms"] >= 0.0
    assert res["plain_has_no_reducer"] and res["forced_has_reducer"]
    assert res["step_bitwise"], "a step through the RCCL reducer at world size 1 must equal the plain step bit for bit"
    assert res["accum_reduces_once_per_step"] and res["accum_bitwise"]
    assert res["compress_bytes"] == 2 * res["n"]
    assert res["compress_max_dev"] < 3 * 1e-3 * 3, res["compress_max_dev"]       # <= lr per step per element, three steps
    assert res["hf_ddp_bitwise"] and res["hf_reducer_ran"] and res["hf_ddp_passed_through"]
    assert res["hf_fast_bitwise"] and res["hf_fast_reducer_ran"] and res["hf_fast_hook_silent"]
