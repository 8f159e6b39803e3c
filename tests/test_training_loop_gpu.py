"""The native training loop on the real kernels (GPU): the fused clip + AdamW step against ``torch.optim.AdamW`` with HF
Trainer's parameter groups, ``run_native_training`` with and without the prefetch thread, checkpoint -> resume reproducing the
uninterrupted run bit for bit, and two ranks launched the way ``torchrun train.py <yaml> --native`` launches them."""
import os
import random
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

from tests.test_end_to_end_gpu import _make_workspace, _merged

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_flat_adamw_step_equals_torch_adamw_with_hf_groups():
    from tests.test_ddp_gpu import _batch, _make
    from adt_str_amd.trainer import FlatTrainer, no_decay_names
    model = _make(seed=3)
    ref_params = {n: p.detach().clone() for n, p in model.named_parameters()}
    tr = FlatTrainer(model, lr=1e-3, weight_decay=0.1, max_grad_norm=0.5, total_steps=20, warmup_ratio=0.1)
    assert tr.warmup == 2
    skip = set(no_decay_names(model))
    params = {n: torch.nn.Parameter(v.clone()) for n, v in ref_params.items()}
    opt = torch.optim.AdamW([{"params": [p for n, p in params.items() if n not in skip], "weight_decay": 0.1},
                             {"params": [p for n, p in params.items() if n in skip], "weight_decay": 0.0}],
                            lr=1e-3, betas=(0.9, 0.999), eps=1e-8)
    for step in range(3):
        wav, tok, tl = _batch(seed=7 + step)
        lr = tr.current_lr()
        tr.train_step(wav, tok, tl)
        for n, p in params.items():
            p.grad = tr.eng.G[n].detach().clone()                       # the step's own (unclipped) gradients
        torch.nn.utils.clip_grad_norm_(list(params.values()), 0.5)
        for g in opt.param_groups:
            g["lr"] = lr
        opt.step()
        for n, p in model.named_parameters():
            err = (p.data - params[n].data).abs().max().item()
            assert err <= 2e-7 + 1e-6 * params[n].data.abs().max().item(), (step, n, err)
            params[n].data.copy_(p.data)                                 # keep the two trajectories on the same weights
    # the decayed and the non-decayed groups really differ: a bias moved only by its gradient, a weight also shrank
    assert float((model.decoder.generator.weight.data - ref_params["decoder.generator.weight"]).abs().max()) > 0


def test_fused_adamw_for_the_hf_path_equals_torch_adamw():
    """adt_str_amd.optim.FusedAdamW (what ADTTrainer.create_optimizer builds) against torch.optim.AdamW on a twin model, driven the way
    the HF Trainer drives an optimizer: forward through the autograd bridge, backward, clip_grad_norm_, scheduler-written lr, step,
    zero_grad -- including a two-micro-batch accumulation step (p.grad accumulated in place) and a state_dict round trip."""
    import copy
    from tests.test_ddp_gpu import _batch, _make
    from adt_str_amd.masks import create_mask_plain
    from adt_str_amd.optim import FusedAdamW
    from adt_str_amd.trainer import no_decay_names
    model = _make(seed=5)
    twin = copy.deepcopy(model)
    skip = set(no_decay_names(model))

    def groups(m):
        named = list(m.named_parameters())
        return [{"params": [p for n, p in named if n not in skip], "weight_decay": 0.1}, {"params": [p for n, p in named if n in skip], "weight_decay": 0.0}]

    ptrs = {n: p.data_ptr() for n, p in model.named_parameters()}
    opt = FusedAdamW(groups(model), lr=1e-3, betas=(0.9, 0.999), eps=1e-8, engine=model.engine)
    ref = torch.optim.AdamW(groups(twin), lr=1e-3, betas=(0.9, 0.999), eps=1e-8)
    assert all(p.data_ptr() != ptrs[n] for n, p in model.named_parameters())          # re-pointed at the flat buffer, names / shapes kept
    assert [tuple(p.shape) for p in model.parameters()] == [tuple(p.shape) for p in twin.parameters()]

    def run(m, o, seeds, lr):
        m.train()
        for sd in seeds:                                                # more than one seed: gradient accumulation, as HF does it
            wav, tok, tl = _batch(seed=sd)
            _, pad = create_mask_plain(tok.shape[1] - 1, tl, wav.device)
            (m(src=wav, tgt=tok[:, :-1], tgt_mask=None, tgt_padding_mask=pad, labels=tok[:, 1:]) / len(seeds)).backward()
        torch.nn.utils.clip_grad_norm_(m.parameters(), 0.5)
        for g in o.param_groups:
            g["lr"] = lr
        o.step()
        o.zero_grad(set_to_none=True)

    def same(tag):
        for (n, p), q in zip(model.named_parameters(), twin.parameters()):
            err = (p.data - q.data).abs().max().item()
            assert err <= 2e-7 + 1e-6 * q.data.abs().max().item(), (tag, n, err)
            q.data.copy_(p.data)                                         # keep the two trajectories on the same weights

    for step, (seeds, lr) in enumerate([((11,), 5e-4), ((12, 13), 1e-3), ((14,), 7e-4)]):
        run(model, opt, seeds, lr)
        run(twin, ref, seeds, lr)
        same(step)
    # the bridge's gradient views were found as ONE flat buffer by address (single backward, accumulation and clipping all work in place on
    # them): the gather path, its second 276 MB buffer and its extra pass never ran
    assert opt._gather is None
    # state_dict round trip into a fresh optimizer over a fresh copy of the model: the next step is the same step
    model2 = copy.deepcopy(model)
    opt2 = FusedAdamW(groups(model2), lr=1e-3, betas=(0.9, 0.999), eps=1e-8, engine=model2.engine)
    opt2.load_state_dict(copy.deepcopy(opt.state_dict()))
    run(model, opt, (15,), 9e-4)
    run(model2, opt2, (15,), 9e-4)
    for p, q in zip(model.parameters(), model2.parameters()):
        assert torch.equal(p.data, q.data)
    run(twin, ref, (15,), 9e-4)
    same("after reload")
    # a gradient that is not the bridge's view (re-allocated by someone): gathered, same result
    wav, tok, tl = _batch(seed=16)
    _, pad = create_mask_plain(tok.shape[1] - 1, tl, wav.device)
    for m in (model, model2):
        m(src=wav, tgt=tok[:, :-1], tgt_mask=None, tgt_padding_mask=pad, labels=tok[:, 1:]).backward()
    for p in model2.parameters():
        p.grad = p.grad.clone()
    opt.step(); opt2.step()
    for p, q in zip(model.parameters(), model2.parameters()):
        assert torch.equal(p.data, q.data)


def _losses(tr):
    return [l for _, l in tr.loss_history]


def test_prefetch_thread_changes_nothing_and_resume_is_bit_exact(tmp_path):
    import train
    from adt_str_amd.trainer import run_native_training
    from data_modules.train_dataset import LakhDataset, LakhDatasetConfig

    os.makedirs(tmp_path / "w")
    base = _make_workspace(tmp_path / "w", n_items=48, batch=4)
    base["training"]["num_epochs"] = 2
    base["logging"]["save_every_n_steps"] = 5
    base["checkpoint"] = {"max_checkpoints": 2}

    def go(out_name, depth, resume=None, auto=False):
        cfg_d = {**base, "logging": dict(base["logging"], output_dir=str(tmp_path / out_name)),
                 "checkpoint": {"max_checkpoints": 2, "resume_from_checkpoint": resume, "auto_resume": auto}}
        cfg, _ = _merged(cfg_d, tmp_path, f"{out_name}-{depth}-{bool(resume)}.yaml")
        random.seed(cfg["experiment"]["seed"]); torch.manual_seed(cfg["experiment"]["seed"])
        model, tokenizer, synth = train.build_components(cfg)
        ds = LakhDataset(LakhDatasetConfig(**cfg["shared"], **cfg["TrainDatasetConfig"]), tokenizer, synth)
        tr = run_native_training(model.cuda(), ds, cfg, prefetch_depth=depth)
        torch.cuda.synchronize()
        return tr

    a = go("inline", 0)
    b = go("thread", 2)
    assert a.step_no == b.step_no == 2 * 12
    assert _losses(a) == _losses(b) and torch.equal(a.pflat, b.pflat)             # same draws, same batches, same bits
    run_dir = tmp_path / "thread" / "t"                                        # output_dir / run_name (reference train.py:171-176)
    assert sorted(d for d in os.listdir(run_dir) if d.startswith("checkpoint-")) == ["checkpoint-15", "checkpoint-20"]
    from safetensors.torch import load_file
    final = load_file(str(run_dir / "model.safetensors"))
    assert set(final) == set(b.model.state_dict()) and torch.equal(final["decoder.generator.bias"].cuda(), b.model.decoder.generator.bias.data)
    # resume from step 15 (mid-epoch 2, with dropout 0.1 and the FX draws in the stream): steps 16..24 repeat bit for bit
    c = go("thread", 2, resume=str(run_dir / "checkpoint-15"))
    assert c.step_no == 24 and _losses(c) == _losses(b)[15:] and torch.equal(c.pflat, b.pflat)
    assert torch.equal(c.m, b.m) and torch.equal(c.v, b.v)
    # build_model loads what the loop saved (the reference's checkpoint contract, build_model.py:49-66)
    from build_model import build_model
    cfg_d = {**base, "inference": {"checkpoint_path": str(run_dir), "batch_size": 2, "max_length": 8}}
    _, cfg_path = _merged(cfg_d, tmp_path, "infer.yaml")
    m2, _ = build_model(cfg_path)
    assert torch.equal(m2.state_dict()["decoder.generator.bias"].cpu(), final["decoder.generator.bias"])


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def test_torchrun_two_ranks_train_native_share_one_model(tmp_path):
    """``python -m torch.distributed.run --nproc-per-node 2 train.py cfg.yaml --native`` (the reference's ``accelerate launch
    train.py``, README.md:53-57): both ranks join one process group, shard the data, and save ONE model.  On this one-GPU box
    the two ranks share GPU 0 over gloo (ADT_SHARE_GPU=1: a one-GPU box cannot host two RCCL ranks); on a node they get
    cuda:LOCAL_RANK and RCCL."""
    os.makedirs(tmp_path / "w")
    ws = _make_workspace(tmp_path / "w", n_items=32, batch=4)
    ws["logging"]["save_every_n_steps"] = 2
    cfg, cfg_path = _merged(ws, tmp_path, "ddp.yaml")
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""), ADT_SHARE_GPU="1", ADT_DUMP_PARAMS=str(tmp_path))
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", str(_free_port()), os.path.join(ROOT, "train.py"), cfg_path, "--native"],
                       capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    p0, p1 = (torch.load(str(tmp_path / f"params_rank{i}.pt")) for i in range(2))
    assert p0["world"] == p1["world"] == 2 and p0["steps"] == p1["steps"] == 32 // (4 * 2)
    assert torch.equal(p0["pflat"], p1["pflat"]), "both ranks must hold the same model"
    from adt_str_amd.trainer import output_path
    out = output_path(cfg)
    assert os.path.exists(os.path.join(out, "model.safetensors")) and os.path.exists(os.path.join(out, "checkpoint-4", "rng_state_1.pth"))


def _torchrun_train(cfg_path, dump_dir, *extra, env_extra=None):
    os.makedirs(dump_dir, exist_ok=True)
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""), ADT_SHARE_GPU="1", ADT_DUMP_PARAMS=str(dump_dir),
               **(env_extra or {}))
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", str(_free_port()), os.path.join(ROOT, "train.py"), cfg_path, *extra],
                       capture_output=True, text=True, env=env, timeout=900)
    if r.returncode != 0:                                  # pytest abbreviates long assertion messages: show the ranks' own output
        print(r.stdout[-3000:])
        print(r.stderr[-12000:])
    assert r.returncode == 0, "torchrun train.py failed (output above)"
    return [torch.load(os.path.join(dump_dir, f"params_rank{i}.pt")) for i in range(2)]


def test_two_rank_resume_keeps_each_ranks_own_dropout_stream(tmp_path):
    """Data-parallel resume (dropout 0.1, FX draws in the stream): the run resumed from ``checkpoint-4`` ends with bitwise the
    parameters of the uninterrupted run on BOTH ranks.  That needs every rank to continue ITS OWN dropout counter (seeded per
    rank) and its own host RNG stream -- a checkpoint that handed rank 0's counter to rank 1 would fail here."""
    os.makedirs(tmp_path / "w")
    ws = _make_workspace(tmp_path / "w", n_items=64, batch=4)                 # 8 steps per rank pair
    ws["logging"]["save_every_n_steps"] = 2
    ws["checkpoint"] = {"max_checkpoints": 8}
    cfg, cfg_path = _merged(ws, tmp_path, "full.yaml")
    full = _torchrun_train(cfg_path, tmp_path / "dump_full", "--native")
    assert full[0]["steps"] == 8 and torch.equal(full[0]["pflat"], full[1]["pflat"])
    from adt_str_amd.trainer import output_path
    ck = os.path.join(output_path(cfg), "checkpoint-4")
    assert os.path.exists(os.path.join(ck, "trainer_state.pt")) and os.path.exists(os.path.join(ck, "rng_state_1.pth"))
    st = torch.load(os.path.join(ck, "trainer_state.pt"), weights_only=False)["trainer"]
    assert st["drop_steps"] == 4 and st["step_no"] == 4
    ws2 = dict(ws, checkpoint={"max_checkpoints": 8, "resume_from_checkpoint": ck})
    _, cfg_path2 = _merged(ws2, tmp_path, "resume.yaml")
    res = _torchrun_train(cfg_path2, tmp_path / "dump_resume", "--native")
    assert res[0]["steps"] == 8
    assert torch.equal(res[0]["pflat"], full[0]["pflat"]) and torch.equal(res[1]["pflat"], full[1]["pflat"])


def test_two_ranks_hf_trainer_ddp_without_native(tmp_path):
    """The reference's own multi-GPU launch (README.md:53-57: ``accelerate launch train.py <yaml>`` -> HF ``Trainer`` -> DDP), two
    ranks, NO ``--native``: ``train()`` creates the process group before any GPU work, accelerate adopts it, DDP averages the
    gradients that ``_ADTLossFn.backward`` hands to autograd, and both ranks end with the same parameters -- which are not the
    initial ones.  (One-GPU box: both ranks on GPU 0 over gloo, ADT_SHARE_GPU=1; on a node: cuda:LOCAL_RANK and RCCL.)"""
    os.makedirs(tmp_path / "w")
    ws = _make_workspace(tmp_path / "w", n_items=32, batch=4)
    cfg, cfg_path = _merged(ws, tmp_path, "hf.yaml")
    p0, p1 = _torchrun_train(cfg_path, tmp_path / "dump_hf")
    assert p0["world"] == p1["world"] == 2 and p0["steps"] == p1["steps"] == 32 // (4 * 2)
    assert torch.equal(p0["pflat"], p1["pflat"]), "DDP must leave both ranks with the same model"
    assert bool(torch.isfinite(p0["pflat"]).all())
    from adt_str_amd.trainer import output_path
    out = output_path(cfg)
    assert os.path.exists(os.path.join(out, "model.safetensors"))
    from safetensors.torch import load_file
    saved = load_file(os.path.join(out, "model.safetensors"))
    from build_model import model_config_from
    from adt_str_amd.network import ADTModel
    torch.manual_seed(cfg["experiment"]["seed"])
    fresh = ADTModel(model_config_from(cfg))
    moved = max(float((saved[k].float() - v.detach().float()).abs().max()) for k, v in fresh.named_parameters())
    assert moved > 1e-4, "training must have changed the weights"
