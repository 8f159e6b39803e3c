"""Data-parallel path with world_size 2 on the gloo backend (CPU): the flat gradient buffer is
tiled exactly by the backward segments, each segment is all-reduced once, the result is the mean
over ranks, and rank 0's parameters reach every rank."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from adt_str_amd.network import ADTModel, ADTModelConfig
        from adt_str_amd.trainer import GradReducer, backward_segments
        torch.manual_seed(rank)                       # different initial weights per rank
        model = ADTModel(ADTModelConfig(input_sec=0.5, time_res=0.01, win_length=2048, sample_rate=16000, enc_layers=2, dec_layers=1,
                                        nhead=1, d_query=128, dropout=0.0, tgt_vocab_size=1400, plain=True, n_mels=128))
        eng = model.engine
        gflat, G = eng.grad_buffers()
        n = gflat.numel()
        segs = backward_segments(eng)
        # the segments tile [0, n) without overlap
        marks = torch.zeros(n, dtype=torch.int32)
        for lo, hi in segs:
            marks[lo:hi] += 1
        assert bool((marks == 1).all()), "backward segments must cover every gradient element exactly once"
        # rank-dependent gradients -> mean
        base = torch.arange(n, dtype=torch.float32) % 1000
        gflat.copy_(base * (rank + 1))
        red = GradReducer(gflat)
        for lo, hi in segs:
            red.segment_ready(lo, hi)
        red.finish()
        expect = base * (sum(r + 1 for r in range(world)) / world)
        assert torch.allclose(gflat, expect)
        # views stay aliased to the flat buffer
        name = "decoder.generator.bias"
        lo, hi = eng.flat_range(name)
        assert torch.equal(G[name], gflat[lo:hi])
        # a missing segment is detected
        red2 = GradReducer(gflat)
        red2.segment_ready(*segs[0])
        try:
            red2.finish()
            ok = False
        except RuntimeError:
            ok = True
        assert ok
        # DDP no_sync(): a disabled pass sends nothing and finish() leaves the local gradients alone
        gflat.copy_(base * (rank + 1))
        red3 = GradReducer(gflat)
        red3.enabled = False
        for lo, hi in segs:
            red3.segment_ready(lo, hi)
        red3.finish()
        assert torch.equal(gflat, base * (rank + 1)) and red3.steps == 0
        # ... and the last micro-step of an accumulation window carries the local sum of the earlier ones, segment by segment
        acc = torch.full_like(gflat, 3.0 * (rank + 1))
        red3.enabled, red3.addend = True, acc
        for lo, hi in segs:
            red3.segment_ready(lo, hi)
        red3.finish()
        assert torch.allclose(gflat, (base + 3.0) * (sum(r + 1 for r in range(world)) / world))
        assert red3.comm_stats()["bytes_per_step"] == 4 * n
        # bf16-compressed segments: half the bytes, the mean within bf16 rounding of the fp32 path (|x| <= 1500 here: ulp 8)
        gflat.copy_(base * (rank + 1))
        red4 = GradReducer(gflat, compress="bf16")
        for lo, hi in segs:
            red4.segment_ready(lo, hi)
        red4.finish()
        st = red4.comm_stats()
        assert st["bytes_per_step"] == 2 * n and st["wire_dtype"] == "bf16"
        err = (gflat - expect).abs()
        assert float(err.max()) <= 8.0 and bool((err <= expect.abs() * 2.0 ** -7 + 1e-6).all()), float(err.max())
        both = [torch.zeros_like(gflat) for _ in range(world)]
        dist.all_gather(both, gflat)
        assert torch.equal(both[0], both[1]), "compressed reduction must still leave every rank with the same gradients"
        # parameter broadcast from rank 0 (what FlatTrainer does once at construction)
        pflat = torch.cat([p.data.reshape(-1) for p in model.parameters()])
        dist.broadcast(pflat, src=0)
        ref = [pflat.clone() for _ in range(world)]
        dist.all_gather(ref, pflat)
        assert all(torch.equal(ref[0], r) for r in ref)
        q.put((rank, "ok"))
    except Exception as e:                              # pragma: no cover
        q.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


def test_gradient_reduction_world_size_2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, "ok"), (1, "ok")], res


def test_lr_schedule_matches_hf_cosine():
    from transformers.optimization import get_cosine_schedule_with_warmup
    from adt_str_amd.trainer import cosine_with_warmup
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.SGD([p], lr=1.0)
    sch = get_cosine_schedule_with_warmup(opt, num_warmup_steps=10, num_training_steps=100)
    for step in range(100):
        assert abs(sch.get_last_lr()[0] - cosine_with_warmup(step, 100, 10)) < 1e-7, step
        opt.step(); sch.step()


class _StubWrapper:
    """Stands in for ClapWrapper in the sharding test: a deterministic 'embedding' of each clip, on the CPU."""
    device = torch.device("cpu")

    def get_audio_features(self, audios):
        return torch.stack([torch.tensor([float(a.numel()), float(a.abs().sum()), float(a[0, 0])] + [0.0] * 509) for a in audios])


def _curation_worker(rank, world, port, q, files):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import importlib
        mod = importlib.import_module("data_modules.augment_data_with_CLAP")
        got = mod._embed(_StubWrapper(), files, 2, 48000)
        want = torch.cat([_StubWrapper().get_audio_features([mod.normalize(mod.load_audio(f, 48000))]) for f in files])
        assert got.shape == want.shape and torch.equal(got, want), "gathered embeddings must be in file order on every rank"
        q.put((rank, "ok"))
    except Exception as e:                              # pragma: no cover
        q.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


def test_curation_embedding_shards_and_gathers_world_size_2(tmp_path):
    """CLAP curation under torchrun: strided file shards per rank, one all_gather, file order restored."""
    import numpy as np
    from adt_str_amd.audio_io import write_wav
    rng = np.random.default_rng(0)
    files = []
    for i in range(7):                                  # odd count: the last rank's shard is one short
        path = str(tmp_path / f"s{i}.wav")
        write_wav(path, (rng.standard_normal(600 + 50 * i) * 0.3).astype(np.float32), 48000)
        files.append(path)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_curation_worker, args=(r, 2, port, q, files)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, "ok"), (1, "ok")], res


def _hf_overlap_worker(rank, world, port, q, accumulation):
    """The reference's launch (HF Trainer + DistributedDataParallel around the autograd bridge) on two gloo ranks, with the engine's
    kernels replaced by a stand-in that produces rank-dependent gradients segment by segment in the backward pass's own order: everything
    else is the product's code (network._ADTLossFn, trainer.install_engine_reduction, GradReducer, DDP and its comm hook)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import time
        from torch.nn.parallel import DistributedDataParallel as DDP
        from adt_str_amd import trainer as T
        from adt_str_amd.network import ADTModel, ADTModelConfig
        torch.manual_seed(0)
        model = ADTModel(ADTModelConfig(input_sec=0.5, time_res=0.01, win_length=2048, sample_rate=16000, enc_layers=2, dec_layers=1,
                                        nhead=1, d_query=128, dropout=0.0, tgt_vocab_size=1400, plain=True, n_mels=128))
        events = []                                             # ("collective", t) / ("backward_end", t)
        calls = {"n": 0}

        def stand_in(eng):
            """Replace this engine object's kernels by the stand-in (a rebuilt engine -- .to() / set_precision -- needs it again)."""
            gflat, _ = eng.grad_buffers()

            def fake_loss_and_grads(src, tgt, pad, labels, want_grads=True, return_logits=False):
                calls["n"] += 1
                eng.generation += 1
                gflat.copy_(base * (rank + 1) * calls["n"])      # this rank's local gradient of this pass
                for pre in ("decoder.",) + tuple(L["p"] + "." for L in reversed(eng.enc)):
                    time.sleep(0.01)                             # the rest of the backward pass is still running ...
                    eng._ready(pre)                              # ... when this segment is final
                eng._ready("encoder.dense_layer.", "encoder.layer_norm.", "project_to_mel.")
                events.append(("backward_end", time.perf_counter()))
                return {"loss": torch.tensor([float(rank + 1)])[0]}

            eng.loss_and_grads = fake_loss_and_grads
            eng.refresh_weights = lambda force=False: None
            eng._flush_reductions = lambda: None
            return gflat

        eng = model.engine
        n = eng.grad_buffers()[0].numel()
        base = (torch.arange(n, dtype=torch.float32) % 997) * 1e-3
        gflat = stand_in(eng)
        real_all_reduce = dist.all_reduce

        def counting_all_reduce(t, *a, **k):
            events.append(("collective", time.perf_counter(), t.numel()))
            return real_all_reduce(t, *a, **k)

        dist.all_reduce = counting_all_reduce
        ddp = DDP(model, broadcast_buffers=False)
        red = T.install_engine_reduction(ddp, accumulation)
        assert T.install_engine_reduction(ddp, accumulation) is red                       # idempotent
        assert (red is not None) == (accumulation == 1)
        src, tgt = torch.zeros(2, 8000), torch.zeros(2, 4, dtype=torch.long)
        pad, labels = torch.zeros(2, 4, dtype=torch.bool), torch.zeros(2, 4, dtype=torch.long)

        def step(sync=True):
            ctx = ddp.no_sync() if not sync else __import__("contextlib").nullcontext()
            with ctx:
                loss = ddp(src=src, tgt=tgt, tgt_mask=None, tgt_padding_mask=pad, labels=labels)
                (loss / accumulation).backward()

        if accumulation == 1:
            step()
            mean = base * (sum(r + 1 for r in range(world)) / world)
            got = torch.cat([p.grad.reshape(-1) for p in eng.named.values()])
            assert torch.allclose(got, mean, rtol=1e-6, atol=1e-7), (got[:4], mean[:4])
            coll = [e for e in events if e[0] == "collective"]
            end = [e for e in events if e[0] == "backward_end"][0][1]
            segs = T.backward_segments(eng)
            assert len(coll) == len(segs) and sum(e[2] for e in coll) == n                # one all-reduce per segment, nothing else ...
            assert coll[0][1] < end and sum(1 for e in coll if e[1] < end) >= len(segs) - 3  # ... started while the backward pass was still running
            assert eng.hf_hook_stats["passed_through"] >= 1 and eng.hf_hook_stats["reduced_by_ddp"] == 0   # DDP's buckets sent nothing
            # a pass inside no_sync() sends nothing at all and leaves the local gradient (accumulated by autograd)
            events.clear()
            model.zero_grad(set_to_none=True)
            step(sync=False)
            assert not [e for e in events if e[0] == "collective"]
            got = torch.cat([p.grad.reshape(-1) for p in eng.named.values()])
            assert torch.allclose(got, base * (rank + 1) * 2, rtol=1e-6)
            # what ADTTrainer.compute_loss calls: the same averaged gradients and the same per-segment collectives, but DDP's reducer stays
            # out of it (no bucket copies, the comm hook is not even asked) -- the forward runs inside no_sync() with the engine told to reduce
            events.clear()
            model.zero_grad(set_to_none=True)
            stats0 = dict(eng.hf_hook_stats)
            loss = T.forward_engine_reduced(ddp, src=src, tgt=tgt, tgt_mask=None, tgt_padding_mask=pad, labels=labels)
            loss.backward()
            assert ddp.require_backward_grad_sync                                          # (the wrapper is left as it was)
            got = torch.cat([p.grad.reshape(-1) for p in eng.named.values()])
            assert torch.allclose(got, base * 3 * (sum(r + 1 for r in range(world)) / world), rtol=1e-6, atol=1e-7)
            coll = [e for e in events if e[0] == "collective"]
            assert len(coll) == len(segs) and sum(e[2] for e in coll) == n and dict(eng.hf_hook_stats) == stats0
            # ... and inside the caller's own no_sync() it is the plain call: nothing is sent
            events.clear()
            model.zero_grad(set_to_none=True)
            with ddp.no_sync():
                T.forward_engine_reduced(ddp, src=src, tgt=tgt, tgt_mask=None, tgt_padding_mask=pad, labels=labels).backward()
            assert not [e for e in events if e[0] == "collective"]
            # the model's engine object is REPLACED (set_precision / .to() / .float() do that): the comm hook must not go on testing the old
            # engine's frozen counters (it would pass every bucket through unreduced and the ranks would diverge).  Even if nobody calls
            # install_engine_reduction again, the first pass of the new engine is reduced by DDP itself and the later ones by the engine.
            model.set_precision("bf16")
            eng2 = model.engine
            assert eng2 is not eng and not getattr(eng2, "_hf_hook_installed", False)
            stand_in(eng2)
            before = dict(eng.hf_hook_stats)
            for k in (5, 6):                                     # calls["n"] is 4 here: passes 5 and 6
                model.zero_grad(set_to_none=True)
                events.clear()
                step()
                got = torch.cat([p.grad.reshape(-1) for p in eng2.named.values()])
                assert torch.allclose(got, base * k * (sum(r + 1 for r in range(world)) / world), rtol=1e-6, atol=1e-7), k
            assert eng2.hf_hook_stats["reduced_by_ddp"] > before["reduced_by_ddp"]        # pass 3: DDP reduced it ...
            assert eng2.hf_hook_stats["passed_through"] > before["passed_through"] and eng2.hf_reducer is not None   # ... pass 4: the engine did
        else:
            # with accumulation the engine leaves the reduction to DDP: local sums, reduced once on the last micro-step
            step(sync=False)
            step(sync=True)
            got = torch.cat([p.grad.reshape(-1) for p in eng.named.values()])
            local = lambda r: base * (r + 1) * (1 + 2) / accumulation
            mean = sum(local(r) for r in range(world)) / world
            assert torch.allclose(got, mean, rtol=1e-5, atol=1e-7)
            assert eng.hf_hook_stats["reduced_by_ddp"] >= 1 and eng.hf_hook_stats["passed_through"] == 0
        q.put((rank, "ok"))
    except Exception as e:                                      # pragma: no cover
        import traceback
        q.put((rank, "fail: " + repr(e) + traceback.format_exc()[-1500:]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("accumulation", [1, 2])
def test_hf_path_overlaps_its_all_reduce_with_the_backward_pass(accumulation):
    """``accelerate launch train.py <yaml>`` (the reference's README.md:53-57): under DDP the engine's backward pass drives one all-reduce per
    gradient segment while it is still running; DDP's own buckets then pass the averaged gradients through (world size 2, gloo)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_hf_overlap_worker, args=(r, 2, port, q, accumulation)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(r[1] == "ok" for r in res), res
