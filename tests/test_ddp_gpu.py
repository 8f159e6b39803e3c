"""Data-parallel training step on real kernels: two ranks (both on GPU 0, gloo transport -- a one-GPU box cannot host two RCCL
ranks) each run ``FlatTrainer.train_step`` on their own batch; afterwards both hold the same parameters, and those are the
parameters a single process gets from the averaged gradients of the two batches (DistributedDataParallel semantics:
broadcast of rank 0's initial weights, gradient mean, identical optimizer step)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _make(seed):
    from adt_str_amd.network import ADTModel, ADTModelConfig
    torch.manual_seed(seed)
    cfg = ADTModelConfig(input_sec=0.5, time_res=0.01, win_length=2048, sample_rate=16000, enc_layers=1, dec_layers=1, nhead=2, d_query=128,
                         dropout=0.0, tgt_vocab_size=1400, plain=True, n_mels=128)
    return ADTModel(cfg).cuda()


def _batch(seed, B=3, L=8000, T=10):
    rng = np.random.default_rng(seed)
    wav = torch.from_numpy(np.clip(rng.standard_normal((B, L)) * 0.1, -1, 1).astype(np.float32)).cuda()
    lens = rng.integers(4, T + 2, B)
    lens[0] = T + 1
    tokens = np.full((B, T + 1), 1, np.int64)
    for b in range(B):
        n = int(lens[b])
        tokens[b, :n] = np.concatenate([[2], rng.integers(4, 530, n - 2), [3]])
    tl = np.where(lens == lens.max(), lens - 1, lens).astype(np.int64)
    return wav, torch.from_numpy(tokens).cuda(), torch.from_numpy(tl).cuda()


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from adt_str_amd.trainer import FlatTrainer
        model = _make(seed=100 + rank)                     # different initial weights: the constructor must broadcast rank 0's
        tr = FlatTrainer(model, lr=1e-3, weight_decay=1e-2, max_grad_norm=1.0, total_steps=10, warmup_ratio=0.0)
        wav, tok, tl = _batch(seed=7 + rank)
        loss = tr.train_step(wav, tok, tl)
        torch.cuda.synchronize()
        q.put((rank, tr.pflat.detach().cpu().numpy(), float(loss)))
    except Exception as e:                                 # pragma: no cover
        q.put((rank, repr(e), None))
    finally:
        dist.destroy_process_group()


def test_two_rank_step_equals_single_process_mean_gradient_step():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=600) for _ in procs), key=lambda r: r[0])
    for p in procs:
        p.join(timeout=120)
    assert all(isinstance(r[1], np.ndarray) for r in res), res
    p0, p1 = res[0][1], res[1][1]
    assert np.array_equal(p0, p1), "both ranks must hold bitwise identical parameters after the step"

    # single process: rank 0's initial weights, gradients of the two batches averaged, the same clip + AdamW
    from adt_str_amd import kernels as K
    from adt_str_amd.trainer import FlatTrainer
    model = _make(seed=100)
    tr = FlatTrainer(model, lr=1e-3, weight_decay=1e-2, max_grad_norm=1.0, total_steps=10, warmup_ratio=0.0)
    eng = tr.eng
    grads = []
    for r in range(2):
        wav, tok, tl = _batch(seed=7 + r)
        tgt_in, labels = tok[:, :-1], tok[:, 1:]
        pad = torch.arange(tgt_in.shape[1], device="cuda").unsqueeze(0) >= tl.unsqueeze(1)
        model.train()
        eng.loss_and_grads(wav, tgt_in, pad, labels, want_grads=True)
        grads.append(tr.gflat.clone())
    tr.gflat.copy_((grads[0] + grads[1]) * 0.5)
    K.grad_norm(tr.gflat, tr.max_norm, out=tr.norm)
    K.adamw_step(tr.pflat, tr.gflat, tr.m, tr.v, 1, tr.current_lr(), tr.betas[0], tr.betas[1], tr.eps, tr.wd, tr.norm, nodecay=tr.nodecay)
    ref = tr.pflat.detach().cpu().numpy()
    assert np.abs(ref - p0).max() < 1e-6, np.abs(ref - p0).max()
