"""Host side of the native training loop (no GPU): HF-parity of the schedule and the decay groups, the batch assembly
against ``collate_fn``, the prefetcher, and ``run_native_training`` itself -- sharding, gradient accumulation, checkpoints,
resume and two gloo ranks -- driven with an injected CPU step in place of the HIP engine (the loop, not the kernels, is under
test here; the same loop runs the real kernels in tests/test_training_loop_gpu.py)."""
import os
import random
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn

from adt_str_amd import trainer as T
from adt_str_amd.data import GpuBatcher, HostBatch, NoteChunkDataset, Prefetcher, collate_fn
from adt_str_amd.tokenizer import MidiTokenizer, MidiTokenizerConfig


# ----------------------------------------------------------------------------- schedule / decay groups vs HF
@pytest.mark.parametrize("name,kw", [("cosine", {}), ("cosine_warmup_with_min_lr", {"min_lr": 1e-5}), ("linear", {}),
                                     ("constant_with_warmup", {})])
def test_lr_schedule_matches_hf_get_scheduler(name, kw):
    from transformers.optimization import get_scheduler
    lr, total, warm = 1e-4, 137, 14
    opt = torch.optim.SGD([nn.Parameter(torch.zeros(1))], lr=lr)
    sch = get_scheduler(name, opt, num_warmup_steps=warm, num_training_steps=total, scheduler_specific_kwargs=kw or None)
    ratio = kw.get("min_lr", 0.0) / lr
    for step in range(total):
        assert abs(sch.get_last_lr()[0] - lr * T.lr_multiplier(name, step, total, warm, ratio)) < 1e-12, (name, step)
        opt.step(); sch.step()


def test_warmup_steps_are_hf_ceil():
    import math
    assert math.ceil(101 * 0.1) == 11          # TrainingArguments.get_warmup_steps: ceil, not floor


def test_no_decay_groups_match_hf_trainer():
    from transformers import Trainer
    from adt_str_amd.network import ADTModel, ADTModelConfig
    model = ADTModel(ADTModelConfig(input_sec=0.5, time_res=0.01, win_length=2048, sample_rate=16000, enc_layers=2, dec_layers=2, nhead=2,
                                    d_query=128, dropout=0.0, tgt_vocab_size=1400, plain=True, n_mels=128))
    names = [n for n, _ in model.named_parameters()]
    # (HF's list also names nn.MultiheadAttention's registered-but-None q/k/v_proj_weight slots: not parameters)
    decay = set(Trainer.get_decay_parameter_names(None, model)) & set(names)
    skip = T.no_decay_names(model)
    assert set(names) - set(skip) == decay
    assert "encoder.layer_norm.weight" in skip and "decoder.generator.bias" in skip and "decoder.generator.weight" not in skip
    rng = T.no_decay_ranges(model.named_parameters(), skip)
    covered = int((rng[:, 1] - rng[:, 0]).sum())
    assert covered == sum(p.numel() for n, p in model.named_parameters() if n in set(skip))
    assert bool((rng[1:, 0] > rng[:-1, 1]).all()) and bool((rng % 4 == 0).all())           # sorted, merged, 4-aligned


# ----------------------------------------------------------------------------- batch assembly / prefetch
class _FakeSynth:
    def __init__(self):
        self.planned = []

    def plan(self, batch):
        self.planned.append([np.asarray(b, np.float32).reshape(-1, 4).copy() for b in batch])
        random.random()                                     # consumes the global stream like the real planner
        return ("plan", len(self.planned))

    def render_plan(self, plan, width=None):
        return torch.zeros((2, 4))


def _rows(n, seed=0):
    r = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        k = int(r.integers(1, 12))
        on = np.sort(r.uniform(0, 2.9, k)).astype(np.float32)
        out.append(np.stack([on, on + 0.1, r.choice([35, 36, 38, 42, 46, 49, 51], k), r.integers(1, 128, k)], 1).astype(np.float32).tobytes())
    return out


def test_host_batch_equals_collate_fn_and_items_follow_the_reference_rules():
    tk = MidiTokenizer(MidiTokenizerConfig(False, 2, 3, 1, 0, True))
    ds = NoteChunkDataset(_rows(40), GpuBatcher(tk, _FakeSynth(), empty_tokens_percentage=0.3, random_velocity_prob=0.5))
    random.seed(1); torch.manual_seed(1)
    items = [ds[i] for i in range(16)]
    assert any(len(it[0]) == 0 and it[1].tolist() == [2, 0, 3] for it in items)            # empty items: [BOS, SIL, EOS], no notes
    hb = ds.batcher.host_batch(items, snapshot_rng=True)
    ref = collate_fn([(torch.zeros(1), it[1]) for it in items])
    assert hb.tokens.dtype == np.int64 and np.array_equal(hb.tokens, ref["tokens"].numpy())
    assert np.array_equal(hb.token_lengths, ref["token_lengths"].numpy())
    assert hb.rng_state is not None and hb.rng_state[0] == random.getstate()
    out = ds.batcher.upload(hb)
    assert set(out) == {"wavs", "tokens", "token_lengths"} and out["tokens"].dtype == torch.int64


def test_prefetcher_is_order_preserving_deterministic_and_propagates_errors():
    def make(i):
        return (i, random.random())
    random.seed(5); inline = list(Prefetcher(make, 20, depth=0))
    random.seed(5); threaded = list(Prefetcher(make, 20, depth=3))
    assert inline == threaded and [i for i, _ in threaded] == list(range(20))
    assert [i for i, _ in Prefetcher(make, 20, start=17, depth=2)] == [17, 18, 19]

    def bad(i):
        if i == 3:
            raise KeyError("boom")
        return i
    pf = Prefetcher(bad, 10, depth=2)
    got = []
    with pytest.raises(KeyError):
        for x in pf:
            got.append(x)
    assert got == [0, 1, 2]
    pf2 = Prefetcher(lambda i: i, 1000, depth=2)            # closing early leaves no thread behind
    next(iter(pf2))
    pf2.close()
    assert not pf2._thread.is_alive()


# ----------------------------------------------------------------------------- the loop with an injected CPU step
class _ToyModel(nn.Module):
    def __init__(self):
        super().__init__()
        torch.manual_seed(0)
        self.lin = nn.Linear(4, 3)
        self.norm = nn.LayerNorm(3)


class _ToyTrainer:
    """A CPU stand-in with FlatTrainer's interface: SGD on a toy model, gradients averaged through the real GradReducer."""

    def __init__(self, model, lr, total_steps, warmup_ratio, min_lr_ratio, scheduler, grad_accum, seed, **kw):
        import math
        self.model, self.lr, self.total_steps, self.scheduler = model, lr, total_steps, scheduler
        self.warmup, self.min_lr_ratio = math.ceil(total_steps * warmup_ratio), min_lr_ratio
        self.grad_accum, self._micro, self.step_no = grad_accum, 0, 0
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.params = list(model.parameters())
        self.n = sum(p.numel() for p in self.params)
        self.gflat, self.acc = torch.zeros(self.n), torch.zeros(self.n)
        self.seen = []                                      # (micro-batch fingerprint)
        if self.world > 1:
            flat = torch.cat([p.data.reshape(-1) for p in self.params])
            dist.broadcast(flat, src=0)
            self._scatter(flat)

    def _scatter(self, flat):
        off = 0
        for p in self.params:
            p.data.copy_(flat[off:off + p.numel()].view_as(p)); off += p.numel()

    def current_lr(self):
        return self.lr * T.lr_multiplier(self.scheduler, self.step_no, self.total_steps, self.warmup, self.min_lr_ratio)

    def micro_step(self, wavs, tokens, token_lengths):
        x = tokens.float()[:, :4]
        loss = (self.model.norm(self.model.lin(x)) ** 2).mean() + 1e-3 * float(token_lengths.sum())
        self.model.zero_grad()
        loss.backward()
        self.gflat.copy_(torch.cat([p.grad.reshape(-1) for p in self.params]))
        if self.world > 1:
            red = T.GradReducer(self.gflat)
            red.segment_ready(0, self.n)
            red.finish()
        self.seen.append(int(tokens.sum()))
        self._micro += 1
        self.acc += self.gflat
        if self._micro < self.grad_accum:
            return loss.detach()
        lr = self.current_lr()
        self.step_no += 1
        self._scatter(torch.cat([p.data.reshape(-1) for p in self.params]) - lr * self.acc / self.grad_accum)
        self.acc.zero_(); self._micro = 0
        return loss.detach()

    def state_dict(self):
        return {"step_no": self.step_no}

    def load_state_dict(self, sd):
        self.step_no = sd["step_no"]


def _cfg(out_dir, **over):
    cfg = {"training": dict(batch_size=4, num_epochs=2, learning_rate=0.05, weight_decay=0.0, max_grad_norm=1.0, warmup_ratio=0.1,
                            gradient_accumulation_steps=1, min_learning_rate=None, lr_scheduler_type="cosine"),
           "logging": dict(output_dir=str(out_dir), logging_steps=1, save_every_n_steps=None),
           "checkpoint": dict(resume_from_checkpoint=None, auto_resume=False, max_checkpoints=None), "experiment": dict(seed=42)}
    for k, v in over.items():
        sec, key = k.split("__")
        cfg[sec][key] = v
    return cfg


def _dataset():
    tk = MidiTokenizer(MidiTokenizerConfig(False, 2, 3, 1, 0, True))
    return NoteChunkDataset(_rows(50, seed=3), GpuBatcher(tk, _FakeSynth(), empty_tokens_percentage=0.1, random_velocity_prob=0.5))


def _run(cfg, depth=2):
    random.seed(cfg["experiment"]["seed"]); torch.manual_seed(cfg["experiment"]["seed"])
    model = _ToyModel()
    tr = T.run_native_training(model, _dataset(), cfg, trainer_factory=_ToyTrainer, prefetch_depth=depth)
    return model, tr


def test_native_loop_prefetch_accumulation_and_final_save(tmp_path):
    m_inline, t_inline = _run(_cfg(tmp_path / "a"), depth=0)
    m_thread, t_thread = _run(_cfg(tmp_path / "b"), depth=2)
    assert t_inline.step_no == 2 * (50 // 4) and t_inline.seen == t_thread.seen                # the thread changes nothing
    assert all(torch.equal(a, b) for a, b in zip(m_inline.state_dict().values(), m_thread.state_dict().values()))
    assert len(t_thread.loss_history) == t_thread.step_no                                       # logging_steps = 1, drained at the end
    # every logged step after the first also reports step_ms and whole-job clips/s (batch 4 x world 1) since the previous logged step
    assert len(t_thread.rate_history) == t_thread.step_no - 1
    assert all(ms > 0 and abs(cps - 4 / (ms * 1e-3)) < 1e-6 * cps for _, ms, cps in t_thread.rate_history)
    from safetensors.torch import load_file
    saved = load_file(str(tmp_path / "b" / "default" / "model.safetensors"))       # output_dir / run_name                                # trainer.save_model()
    assert set(saved) == set(m_thread.state_dict()) and torch.equal(saved["lin.weight"], m_thread.lin.weight.data)
    _, t_acc = _run(_cfg(tmp_path / "c", training__gradient_accumulation_steps=3))
    assert t_acc.step_no == 2 * (50 // 12) and len(t_acc.seen) == 3 * t_acc.step_no and t_acc.seen == t_inline.seen[:12] + t_inline.seen[12:24]
    with pytest.raises(ValueError):
        _run(_cfg(tmp_path / "d", training__batch_size=64))


def test_checkpoints_rotate_and_resume_reproduces_the_uninterrupted_run(tmp_path):
    full_model, full = _run(_cfg(tmp_path / "full", logging__save_every_n_steps=5, checkpoint__max_checkpoints=2))
    run_dir = tmp_path / "full" / "default"                                                     # output_dir / run_name (reference train.py:171-176)
    assert T.output_path(_cfg(tmp_path / "full")) == str(run_dir)
    dirs = sorted(os.listdir(run_dir))
    assert [d for d in dirs if d.startswith("checkpoint-")] == ["checkpoint-15", "checkpoint-20"]                # save_total_limit
    assert os.path.exists(run_dir / "checkpoint-20" / "rng_state_0.pth")
    # a run that stops after 15 steps (kept checkpoint), then resumes: mid-epoch (12 steps per epoch), same data stream afterwards
    cfg = _cfg(tmp_path / "full", logging__save_every_n_steps=5, checkpoint__max_checkpoints=2,
               checkpoint__resume_from_checkpoint=str(run_dir / "checkpoint-15"))
    random.seed(999); torch.manual_seed(999)                                                  # resume must not depend on the ambient RNG
    model = _ToyModel()
    with torch.no_grad():
        for p in model.parameters():
            p.add_(1.0)                                                                       # ... nor on the initial weights
    tr = T.run_native_training(model, _dataset(), cfg, trainer_factory=_ToyTrainer)
    assert tr.step_no == full.step_no and tr.seen == full.seen[15:]
    assert all(torch.equal(a, b) for a, b in zip(model.state_dict().values(), full_model.state_dict().values()))
    assert T.latest_checkpoint(str(run_dir)).endswith("checkpoint-20")
    cfg2 = _cfg(tmp_path / "full", checkpoint__auto_resume=True)
    model2 = _ToyModel()
    tr2 = T.run_native_training(model2, _dataset(), cfg2, trainer_factory=_ToyTrainer)
    assert tr2.seen == full.seen[20:]


def test_pruning_never_removes_the_checkpoint_just_written(tmp_path):
    """Resuming from a non-latest checkpoint leaves stale higher-numbered directories behind: ``save_total_limit`` must drop
    those (oldest first), never the directory that was just written."""
    import time
    out = tmp_path / "run"
    for step in (30, 40):                                                     # stale checkpoints of an earlier, longer run
        d = out / f"checkpoint-{step}"
        os.makedirs(d)
        torch.save({}, d / "trainer_state.pt")
        time.sleep(0.02)
    new = out / "checkpoint-10"
    os.makedirs(new)
    torch.save({}, new / "trainer_state.pt")
    T._prune_checkpoints(str(out), 2, str(new))
    assert sorted(os.listdir(out)) == ["checkpoint-10", "checkpoint-40"]
    T._prune_checkpoints(str(out), 1, str(new))
    assert sorted(os.listdir(out)) == ["checkpoint-10"]


def test_scheduler_selection_follows_the_reference(tmp_path):
    """train.py:203-218 of the reference: the min-LR cosine replaces the schedule only for ``cosine`` with a positive floor."""
    def sched(**kw):
        _, tr = _run(_cfg(tmp_path / str(abs(hash(tuple(sorted(kw.items()))))), training__num_epochs=1, **{"training__" + k: v for k, v in kw.items()}))
        return tr.scheduler, tr.min_lr_ratio
    assert sched(lr_scheduler_type="cosine", min_learning_rate=0.005) == ("cosine_warmup_with_min_lr", pytest.approx(0.1))
    assert sched(lr_scheduler_type="linear", min_learning_rate=0.005) == ("linear", 0.0)
    assert sched(lr_scheduler_type="cosine", min_learning_rate=0.0) == ("cosine", 0.0)
    assert sched(lr_scheduler_type="constant", min_learning_rate=None) == ("constant", 0.0)


def test_missing_rank_rng_file_is_reported_on_resume(tmp_path):
    _, full = _run(_cfg(tmp_path / "r", logging__save_every_n_steps=5))
    ck = tmp_path / "r" / "default" / "checkpoint-10"
    assert (ck / "trainer_state.pt").exists() and not (ck / T.INCOMPLETE_SENTINEL).exists()   # a finished checkpoint carries no sentinel
    os.remove(ck / "rng_state_0.pth")
    # the checkpoint recorded an RNG file per rank: resuming without it would silently change the data stream -> refused ...
    with pytest.raises(FileNotFoundError, match="rng_state_0.pth is missing"):
        T.run_native_training(_ToyModel(), _dataset(), _cfg(tmp_path / "r", checkpoint__resume_from_checkpoint=str(ck)), trainer_factory=_ToyTrainer)
    # ... unless the user opts in, which still warns
    monkey = {"ADT_ALLOW_MISSING_RNG": "1"}
    os.environ.update(monkey)
    try:
        with pytest.warns(UserWarning, match="rng_state_0.pth is missing"):
            T.run_native_training(_ToyModel(), _dataset(), _cfg(tmp_path / "r", checkpoint__resume_from_checkpoint=str(ck)), trainer_factory=_ToyTrainer)
    finally:
        os.environ.pop("ADT_ALLOW_MISSING_RNG")


def test_checkpoint_marker_waits_for_every_ranks_rng_file_and_dead_directories_are_pruned(tmp_path, monkeypatch):
    """``trainer_state.pt`` (the completeness marker) is not written while a rank's RNG file is absent; directories a killed run left
    without the marker and older than a completed checkpoint are removed by the rotation."""
    out = tmp_path / "o"
    dead = out / "checkpoint-3"
    os.makedirs(dead)
    (dead / "model.safetensors").write_bytes(b"partial")
    (dead / T.INCOMPLETE_SENTINEL).write_bytes(b"")                      # what save_checkpoint drops at makedirs time
    # directories this loop did NOT create share output_dir (train.py without --native = HF Trainer; the reference's own checkpoints):
    # no trainer_state.pt, so the old rule would have deleted them at the first native save
    hf = out / "checkpoint-2"
    os.makedirs(hf)
    (hf / "trainer_state.json").write_text("{}")
    (hf / "optimizer.pt").write_bytes(b"hf")
    unknown = out / "checkpoint-1"                                        # no sentinel, nothing recognisable: not ours either
    os.makedirs(unknown)
    (unknown / "model.safetensors").write_bytes(b"someone else's")
    mixed = out / "checkpoint-4"                                          # sentinel AND a foreign file (an HF run re-used the directory)
    os.makedirs(mixed)
    (mixed / T.INCOMPLETE_SENTINEL).write_bytes(b"")
    (mixed / "trainer_state.json").write_text("{}")
    d = out / "checkpoint-5"
    os.makedirs(d)
    torch.save({}, d / "rng_state_0.pth")
    monkeypatch.setattr(T, "RNG_FILE_WAIT_S", 0.2)
    with pytest.raises(RuntimeError, match="rng_state files of ranks \\[1\\]"):
        T._finish_checkpoint(str(d), {"w": torch.zeros(2)}, None, {}, {"step": 5}, 2, str(out), 3, rng_files=2)
    assert not (d / "trainer_state.pt").exists() and T._checkpoint_dirs(str(out)) == []
    torch.save({}, d / "rng_state_1.pth")                               # the lagging rank arrives
    T._finish_checkpoint(str(d), {"w": torch.zeros(2)}, None, {}, {"step": 5}, 2, str(out), 3, rng_files=2)
    assert (d / "trainer_state.pt").exists() and not dead.exists()
    assert hf.exists() and (hf / "optimizer.pt").exists() and unknown.exists() and mixed.exists()
    # without save_total_limit nothing is ever removed, not even this loop's own dead directories
    dead2 = out / "checkpoint-0"
    os.makedirs(dead2)
    (dead2 / T.INCOMPLETE_SENTINEL).write_bytes(b"")
    T._prune_checkpoints(str(out), None, str(d))
    assert dead2.exists()
    st = torch.load(d / "trainer_state.pt", weights_only=False)
    assert st["rng_files"] == 2 and st["world"] == 2


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _rank_worker(rank, world, port, out_dir, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world))
    try:
        assert T.init_distributed("gloo") == (rank, rank, world) and dist.get_world_size() == world
        random.seed(42); torch.manual_seed(42 + rank)                                           # different initial weights per rank
        model = _ToyModel()
        with torch.no_grad():
            model.lin.weight.add_(float(rank))
        tr = T.run_native_training(model, _dataset(), _cfg(out_dir, logging__save_every_n_steps=4), trainer_factory=_ToyTrainer)
        flat = torch.cat([p.data.reshape(-1) for p in model.parameters()])
        both = [torch.zeros_like(flat) for _ in range(world)]
        dist.all_gather(both, flat)
        assert torch.equal(both[0], both[1]), "ranks must end with identical parameters"
        q.put((rank, tr.step_no, tr.seen, os.path.exists(os.path.join(out_dir, "default", "checkpoint-4", f"rng_state_{rank}.pth"))))
    except Exception as e:                                   # pragma: no cover
        import traceback
        q.put((rank, repr(e), traceback.format_exc(), False))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def test_two_gloo_ranks_through_run_native_training_end_identical(tmp_path):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank_worker, args=(r, 2, port, str(tmp_path), q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert all(isinstance(r[1], int) for r in res), res
    assert res[0][1] == res[1][1] == 2 * (50 // 8)                                              # steps per rank: len // (bs * world)
    assert res[0][2] != res[1][2] and res[0][3] and res[1][3]                                   # disjoint shards, per-rank RNG files


def test_bench_launch_line_is_the_drivers():
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    cmd = bench.launcher_command(8, 29511, ["--gpus", "8", "--steps", "5"])
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-4:] == ["--gpus", "8", "--steps", "5"]
    assert cmd[cmd.index("--master-port") + 2].endswith("bench.py")


def test_bench_refuses_a_launcher_with_another_world_size():
    """``bench.py --gpus N`` under a launcher that started a different number of ranks must stop (non-zero, before anything touches a GPU):
    a silent run would report N GPUs' worth of throughput for another rank count."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], cwd=root, env=env,
                         capture_output=True, text=True, timeout=300)
    assert out.returncode != 0
    assert "--gpus 2 but the launcher started 3 ranks" in out.stderr and not [l for l in out.stdout.splitlines() if l.startswith("{")]
