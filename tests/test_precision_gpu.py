"""The two fp32-operand parity arms (engine ``precision="fp32"`` / ``"bf16x3"``, kernels of csrc/precise.hip) on the GPU.

BASELINE's target is "logits within 1e-3 rel-tol of the CPU reference" (the reference's fp32 CPU path,
model.py:240-258).  Stated tolerances, all relative to the comparator's own magnitude:

  * logits:  max|got - ref| <= 1e-3 * max|ref|   AND   |got - ref| <= 1e-3 * |ref| + 1e-3 * std(ref) element-wise
             (asserted 10x tighter, 1e-4, which is what the fp32 path actually reaches: it differs from the CPU
             only in summation order)
  * loss:    <= 1e-4 relative (asserted 2e-5)
  * grads:   <= 1e-3 of each tensor's max
  * greedy token ids: identical

Comparators: oracle/adt.py in fp32 (pinned by tests/test_oracle_golden.py) and, directly, the tensors the reference's
own ADTModel produced (tests/golden/adt_tiny.npz: logits, memory, loss, nine parameter gradients, greedy ids).

``"bf16x3"`` (round 6) is the same path with every product taken as three bf16 MFMAs on hi / lo splits of the fp32 operands: ~1e-5
relative per product instead of 6e-8.  Logits are asserted at the exact arm's 1e-4 (measured 6e-6 ... 9e-6 of the largest logit, gradients 1.3e-5 of each tensor's maximum);
the loss at 1e-4 and single kernels at 1e-4 of the output's maximum (ten times the exact arm's bound); greedy ids must be identical."""
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import adt as o_adt
from oracle import dropout as o_drop

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
REL = 1e-4            # asserted logit tolerance of the exact arm (BASELINE states 1e-3)
ARMS = ("fp32", "bf16x3")
LOGIT_REL = {"fp32": 1e-4, "bf16x3": 1e-4}       # asserted (measured: 1e-6 / 9e-6 at config[3]); BASELINE's statement is 1e-3 for both
LOSS_REL = {"fp32": 2e-5, "bf16x3": 1e-4}
KERNEL_REL = {"fp32": 1e-5, "bf16x3": 1e-4}      # a single product against fp64, relative to the output's maximum
PRODUCTS = {"fp32": "f32", "bf16x3": "bf16x3"}


@pytest.fixture(autouse=True)
def _exact_products_by_default():
    """Kernel-level tests select the product form through ``kernels.set_f32_products``; every test starts and ends on the exact one."""
    from adt_str_amd import kernels as k
    k.set_f32_products("f32")
    yield
    k.set_f32_products("f32")


def rnd(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(DEV)


def assert_logits_close(got, ref, rel=REL):
    got, ref = got.double().cpu(), ref.double().cpu()
    diff = (got - ref).abs()
    assert diff.max() <= rel * ref.abs().max(), (diff.max().item(), ref.abs().max().item())
    assert bool((diff <= rel * ref.abs() + rel * ref.std()).all())


# ----------------------------------------------------------------------------- kernels
@pytest.mark.parametrize("arm", ARMS)
@pytest.mark.parametrize("M,N,Kd", [(36, 1400, 768), (153, 96, 32), (300, 260, 128), (1, 768, 3072), (129, 132, 20), (768, 3072, 9000)])
def test_gemm_f32_all_layouts(M, N, Kd, arm):
    """(the last shape is a weight-gradient-like product: few output tiles, long K -> the K-split path through fp32 slabs)"""
    from adt_str_amd import kernels as k
    k.set_f32_products(PRODUCTS[arm])
    a, w = rnd((M, Kd), 1), rnd((N, Kd), 2, 0.2)
    ref = a.double() @ w.double().t()
    tol = KERNEL_REL[arm] * ref.abs().max().item() + 1e-6
    assert (k.gemm(a, w).double() - ref).abs().max() <= tol                                   # y = x W^T
    assert (k.gemm(a, w.t().contiguous(), b_kn=True).double() - ref).abs().max() <= tol       # a [M,K] @ b [K,N]
    if M % 4 == 0:                                                                            # a stored [K, M]
        at, wt = a.t().contiguous(), w.t().contiguous()
        got = k.gemm(at, wt, trans=True)
        assert (got.double() - ref).abs().max() <= tol
        assert torch.equal(k.gemm(at, wt, trans=True), got)                                   # K splits are summed in a fixed order
    # strided views (packed projections): a = columns of a wider buffer, w = rows of a packed weight
    wide, packed = rnd((M, 3 * Kd), 3), rnd((3 * N, Kd), 4, 0.2)
    got = k.gemm(wide[:, Kd:2 * Kd], packed[N:2 * N])
    assert (got.double() - wide[:, Kd:2 * Kd].double() @ packed[N:2 * N].double().t()).abs().max() <= tol * 1.5


@pytest.mark.parametrize("arm", ARMS)
def test_gemm_f32_epilogue_order(arm):
    from adt_str_amd import kernels as k
    k.set_f32_products(PRODUCTS[arm])
    M, N, Kd, S = 96, 260, 64, 32
    tol = 1e-5 * (KERNEL_REL[arm] / 1e-5)           # absolute, on outputs of magnitude ~10
    a, w, bias, res, pe = rnd((M, Kd), 1), rnd((N, Kd), 2, 0.2), rnd((N,), 3), rnd((M, N), 4), rnd((S, N), 5)
    z = (a.double() @ w.double().t() + bias.double())
    u = torch.empty((M, N), device=DEV)
    h = k.gemm(a, w, bias=bias, act=1, pre_act_out=u)
    assert (u.double() - z).abs().max() < tol and (h.double() - F.gelu(z)).abs().max() < tol
    assert (k.gemm(a, w, bias=bias, act=2).double() - z.clamp(min=0)).abs().max() < tol
    assert (k.gemm(a, w, bias=bias, residual=res, alpha=0.5).double() - (0.5 * (z - bias.double()) + bias.double() + res.double())).abs().max() < tol
    assert (k.gemm(a, w, residual=pe, res_row_mod=S).double() - ((z - bias.double()).view(M // S, S, N) + pe.double()).view(M, N)).abs().max() < tol
    # dgrad through GELU: dy W * gelu'(u), plus the column sums of the result (the bias gradient)
    ur = u.double().clone().requires_grad_(True)
    dy = rnd((M, N), 6)
    F.gelu(ur).backward(dy.double())
    cs = torch.empty(Kd, device=DEV)
    got = k.gemm(dy, w, b_kn=True, gelu_grad_of=rnd((M, Kd), 7), colsum_out=cs)          # shapes only: [M,N] @ [N,Kd]
    assert got.shape == (M, Kd) and (cs.double() - got.double().sum(0)).abs().max() < 10 * tol
    one = k.gemm(torch.eye(N, device=DEV)[:M].contiguous(), torch.eye(N, device=DEV), gelu_grad_of=u, alpha=1.0)
    eye_grad = torch.zeros((M, N), dtype=torch.float64, device=DEV)
    eye_grad[:, :] = torch.eye(N, dtype=torch.float64, device=DEV)[:M]
    gp = torch.autograd.grad(F.gelu(ur).sum(), ur)[0]
    assert (one.double() - eye_grad * gp).abs().max() < tol                                 # gelu' exact to fp32
    # dropout positions: before / after the residual add, the same counter-based mask as the bf16 kernels
    site = k.drop_site(0.25, 11, 5)
    sc = o_drop.scale((M, N), *site).to(DEV).double()
    assert (k.gemm(a, w, bias=bias, residual=res, drop=site).double() - (z * sc + res.double())).abs().max() < tol
    assert (k.gemm(a, w, bias=bias, residual=res, drop=site, drop_after_residual=True).double() - (z + res.double()) * sc).abs().max() < tol


ATTN_CASES = [  # B, H, dh, Sq, Sk, causal, padded, p_drop
    (2, 2, 128, 128, 128, False, False, 0.0),
    (2, 3, 128, 77, 50, True, True, 0.0),
    (1, 2, 128, 1, 200, False, False, 0.0),
    (2, 2, 128, 257, 300, True, True, 0.0),
    (2, 2, 16, 36, 36, False, False, 0.0),       # the golden model's head size
    (3, 2, 16, 12, 36, True, True, 0.0),
    (1, 2, 64, 130, 97, False, True, 0.0),
    (1, 1, 32, 33, 65, True, False, 0.0),
    (2, 2, 128, 96, 150, False, False, 0.2),     # dropout on the probabilities, masks regenerated in both backward kernels
    (1, 2, 16, 70, 70, True, True, 0.2),
]


@pytest.mark.parametrize("arm", ARMS)
@pytest.mark.parametrize("B,H,dh,Sq,Sk,causal,padded,pdrop", ATTN_CASES)
def test_attention_f32_forward_backward(B, H, dh, Sq, Sk, causal, padded, pdrop, arm):
    from adt_str_amd import kernels as k
    k.set_f32_products(PRODUCTS[arm])
    ktol = KERNEL_REL[arm]
    d = H * dh
    qbuf, kvbuf = rnd((B * Sq, 3 * d), 1), rnd((B * Sk, 3 * d), 2)
    q, kk, v = qbuf[:, :d], kvbuf[:, d:2 * d], kvbuf[:, 2 * d:]
    key_len = torch.tensor([max(1, Sk - 7 * (i + 1)) for i in range(B)], dtype=torch.int32, device=DEV) if padded else None
    scale = 1.0 / math.sqrt(dh)
    site = k.drop_site(pdrop, 5, 4)
    o, lse = k.attn_fwd(q, kk, v, B, H, Sq, Sk, scale, causal, key_len, drop=site, head_dim=dh)
    qr, kr, vr = (t.double().clone().requires_grad_(True) for t in (q, kk, v))
    qh, kh, vh = (t.view(B, -1, H, dh).transpose(1, 2) for t in (qr, kr, vr))
    s = qh @ kh.transpose(-1, -2) * scale
    if causal:
        s = s + torch.triu(torch.ones(Sq, Sk, device=DEV, dtype=torch.float64), 1) * -1e4
    if padded:
        s = s + ((torch.arange(Sk, device=DEV)[None, :] >= key_len[:, None]).double() * -1e4)[:, None, None, :]
    p = torch.softmax(s, -1)
    if site is not None:
        p = p * o_drop.scale((B, H, Sq, Sk), *site).to(DEV).double()
    ref = (p @ vh).transpose(1, 2).reshape(B * Sq, d)
    assert (o.double() - ref).abs().max() <= ktol * ref.abs().max() + 1e-6
    assert (lse.double() - torch.logsumexp(s, -1)).abs().max() <= 2e-5 * (ktol / 1e-5)
    dout = rnd((B * Sq, d), 3)
    ref.backward(dout.double())
    dqb, dkvb = torch.zeros_like(qbuf), torch.zeros_like(kvbuf)
    bg = torch.full((3 * d,), float("nan"), device=DEV)
    k.attn_bwd(q, kk, v, o, dout, lse, dqb[:, :d], dkvb[:, d:2 * d], dkvb[:, 2 * d:], B, H, Sq, Sk, scale, causal, key_len, drop=site,
               bias_grad=bg, head_dim=dh)
    for name, got, rg in (("dq", dqb[:, :d], qr.grad), ("dk", dkvb[:, d:2 * d], kr.grad), ("dv", dkvb[:, 2 * d:], vr.grad)):
        assert (got.double() - rg).abs().max() <= ktol * rg.abs().max() + 1e-7, name
    assert (bg[:d].double() - qr.grad.sum(0)).abs().max() <= 1e-4 * qr.grad.abs().max() * math.sqrt(B * Sq)
    assert bool((bg[d:2 * d] == 0).all())
    assert (bg[2 * d:].double() - vr.grad.sum(0)).abs().max() <= 1e-4 * vr.grad.abs().max() * math.sqrt(B * Sk)
    assert bool((dqb[:, d:] == 0).all()) and bool((dkvb[:, :d] == 0).all())               # nothing written outside the views


def test_row_kernels_f32_outputs():
    """cross-entropy with fp32 gradients, LayerNorm backward with an fp32 branch gradient, the fp32 one-hot embedding gradient."""
    from adt_str_amd import kernels as k
    M, V, D = 70, 1400, 768
    logits, labels = rnd((M, V), 1, 2.0), torch.randint(0, V, (M,), generator=torch.Generator().manual_seed(0)).to(DEV)
    labels[::7] = 1
    lr = logits.double().clone().requires_grad_(True)
    ref = F.cross_entropy(lr, labels, ignore_index=1)
    ref.backward()
    loss, dl = k.cross_entropy(logits, labels, grad_dtype=torch.float32)
    assert dl.dtype == torch.float32 and abs(loss.item() - ref.item()) < 2e-6 * ref.item()
    assert (dl.double() - lr.grad).abs().max() < 1e-6 * lr.grad.abs().max() + 1e-9
    x, dy, g, b = rnd((M, D), 2, 2.0), rnd((M, D), 3), 1 + rnd((D,), 4, 0.1), rnd((D,), 5, 0.1)
    xr = x.double().clone().requires_grad_(True)
    F.layer_norm(xr, (D,), g.double(), b.double()).backward(dy.double())
    _, _, mean, rstd = k.layernorm_fwd(x, g, b)
    s_dx = k.drop_site(0.1, 1, 2)
    dx32, dxb = k.layernorm_bwd(dy, x, g, mean, rstd, dx16_drop=s_dx, branch_dtype=torch.float32)
    assert dxb.dtype == torch.float32 and (dx32.double() - xr.grad).abs().max() < 2e-5
    assert torch.equal(dxb, dx32 * o_drop.scale((M, D), *s_dx).to(DEV))
    dx32b, alias = k.layernorm_bwd(dy, x, g, mean, rstd, branch_dtype=torch.float32)
    assert alias is dx32b and torch.equal(dx32b, dx32)
    B, T, Dm = 5, 17, 256
    tokens = torch.randint(0, V, (B, T), generator=torch.Generator().manual_seed(0)).to(DEV)
    dyt, dtab = rnd((B * T, Dm), 6), torch.empty((V, Dm), device=DEV)
    k.embed_bwd(tokens, dyt, math.sqrt(Dm), dtab, f32=True)
    ref_d = torch.zeros((V, Dm), dtype=torch.float64, device=DEV).index_add_(0, tokens.reshape(-1), dyt.double() * math.sqrt(Dm))
    assert (dtab.double() - ref_d).abs().max() < 1e-6 * ref_d.abs().max()


# ----------------------------------------------------------------------------- network
def make_model(enc_layers, dec_layers, nhead, seed=0, precision="fp32", d_query=128, dropout=0.0):
    from adt_str_amd.network import ADTModel, ADTModelConfig
    cfg = ADTModelConfig(input_sec=0.5, time_res=0.01, win_length=2048, sample_rate=16000, enc_layers=enc_layers,
                         dec_layers=dec_layers, nhead=nhead, d_query=d_query, dropout=dropout, tgt_vocab_size=1400, plain=True, n_mels=128)
    model = ADTModel(cfg).set_precision(precision)
    state = o_adt.seeded_state(model.state_dict(), seed)
    model.load_state_dict(state)
    return model.to(DEV), {k: v.clone() for k, v in state.items()}, dict(nhead=nhead, sample_rate=16000, win_length=2048,
                                                                         time_res=0.01, n_mels=128)


def grad_state(state):
    return {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "pos_embedding" not in k and "compute_spec" not in k else v)
            for k, v in state.items()}


def run_engine(model, batch, want_grads=True):
    tok = torch.from_numpy(np.asarray(batch["tokens"])).to(DEV)
    T = tok.shape[1] - 1
    pad = (torch.arange(T)[None, :] >= torch.from_numpy(np.asarray(batch["token_lengths"]))[:, None]).to(DEV)
    return model.engine.loss_and_grads(torch.from_numpy(np.asarray(batch["wavs"])).to(DEV), tok[:, :-1], pad, tok[:, 1:],
                                       want_grads=want_grads, return_logits=True)


@pytest.mark.parametrize("arm", ARMS)
@pytest.mark.parametrize("enc_layers,dec_layers,nhead", [(1, 1, 2), (2, 2, 3)])
def test_fp32_logits_loss_grads_within_target_of_the_fp32_oracle(enc_layers, dec_layers, nhead, arm):
    from tests.test_network_gpu import make_batch
    model, state, cfg = make_model(enc_layers, dec_layers, nhead, precision=arm)
    assert model.engine.precision == arm
    batch = make_batch(3, 8000, 12, 1)
    st = grad_state(state)
    ref = o_adt.compute_loss(st, cfg, batch)
    ref["loss"].backward()
    model.train()
    out = run_engine(model, batch)
    assert out["logits"].dtype == torch.float32 and out["memory"].dtype == torch.float32
    assert_logits_close(out["logits"], ref["logits"].detach(), LOGIT_REL[arm])
    assert_logits_close(out["memory"].view(ref["memory"].shape), ref["memory"].detach(), LOGIT_REL[arm])
    assert abs(out["loss"].item() - ref["loss"].item()) <= LOSS_REL[arm] * ref["loss"].item()
    print(arm, "logits max|d| / max|ref|", ((out["logits"].cpu() - ref["logits"].detach()).abs().max() / ref["logits"].abs().max()).item())
    worst = 0.0
    for name, g in model.engine.G.items():
        rg = st[name].grad
        rel = (g.cpu() - rg).abs().max().item() / (rg.abs().max().item() + 1e-12)
        worst = max(worst, rel)
        assert rel <= 1e-3, f"{name}: grad rel err {rel}"
    print(arm, "path: worst grad rel err", worst)
    # run to run: every reduction has a fixed order
    g1, l1 = model.engine.gflat.clone(), out["loss"].clone()
    again = run_engine(model, batch)
    assert torch.equal(again["loss"], l1) and torch.equal(model.engine.gflat, g1)


def tiny_state(g):
    state = {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("w::")}
    d = state["encoder.dense_layer.weight"].shape[0]
    state["encoder.positional_encoding.pos_embedding"] = o_adt.positional_encoding(d)
    state["decoder.positional_encoding.pos_embedding"] = o_adt.positional_encoding(d)
    return state


@pytest.mark.parametrize("arm", ARMS)
def test_fp32_against_tensors_captured_from_the_reference(golden_dir, arm):
    """tests/golden/adt_tiny.npz holds what the reference's own ADTModel (model.py) computed on CPU in fp32 for explicit
    weights: encoder memory, logits, loss, nine gradients and the greedy ids of ADTModel.sample."""
    from adt_str_amd.network import ADTModel, ADTModelConfig
    g = np.load(os.path.join(golden_dir, "adt_tiny.npz"))
    state = tiny_state(g)
    d = state["encoder.dense_layer.weight"].shape[0]
    cfg = ADTModelConfig(input_sec=0.5, time_res=0.01, win_length=2048, sample_rate=16000, enc_layers=1, dec_layers=1, nhead=2,
                         d_query=d // 2, dropout=0.0, tgt_vocab_size=1400, plain=True, n_mels=128)
    model = ADTModel(cfg).set_precision(arm)
    missing = model.load_state_dict(state, strict=False)
    assert all(k.startswith("compute_spectrogram.") for k in missing.missing_keys) and not missing.unexpected_keys
    model = model.to(DEV).train()
    out = run_engine(model, {"wavs": g["wave"], "tokens": g["tokens"], "token_lengths": g["token_lengths"]})
    assert_logits_close(out["memory"].view(g["memory"].shape), torch.from_numpy(g["memory"]), LOGIT_REL[arm])
    assert_logits_close(out["logits"], torch.from_numpy(g["logits"]), LOGIT_REL[arm])
    assert abs(out["loss"].item() - float(g["loss"])) <= LOSS_REL[arm] * float(g["loss"])
    for key in g.files:
        if key.startswith("g::"):
            ref = torch.from_numpy(g[key])
            assert (model.engine.G[key[3:]].cpu() - ref).abs().max() <= 1e-3 * ref.abs().max(), key
    wave = torch.from_numpy(g["wave"]).to(DEV)
    for use_cache in (True, False):
        ids = model.sample(wave, None, None, max_length=10, use_cache=use_cache).cpu().numpy()
        assert np.array_equal(ids, g["sample_ids"]), (use_cache, ids, g["sample_ids"])


@pytest.mark.parametrize("arm", ARMS)
def test_fp32_greedy_ids_identical_to_the_oracle(arm):
    from tests.test_network_gpu import make_batch
    model, state, cfg = make_model(2, 2, 2, seed=3, precision=arm)
    src = torch.from_numpy(make_batch(4, 8000, 6, 3)["wavs"])
    ref = o_adt.greedy_sample(state, cfg, src, max_length=12)
    for use_cache in (True, False):
        got = model.sample(src.to(DEV), None, None, max_length=12, use_cache=use_cache).cpu()
        assert torch.equal(got, ref), (use_cache, got, ref)


@pytest.mark.parametrize("arm", ARMS)
def test_fp32_with_dropout_matches_the_oracle_with_the_same_masks(arm):
    from adt_str_amd import kernels as k
    from tests.test_network_gpu import make_batch
    model, state, cfg = make_model(2, 2, 2, dropout=0.1, precision=arm)
    batch = make_batch(3, 8000, 12, 1)
    model.train()
    out = run_engine(model, batch)
    eng = model.engine
    seed, sites = eng.drop_seed, dict(eng._sites)

    def drop(site):
        key = k.drop_site(0.1, seed, sites[site])
        return lambda shape: o_drop.scale(tuple(shape), *key)

    st = grad_state(state)
    ref = o_adt.compute_loss(st, cfg, batch, drop=drop)
    ref["loss"].backward()
    assert_logits_close(out["logits"], ref["logits"].detach(), LOGIT_REL[arm])
    assert abs(out["loss"].item() - ref["loss"].item()) <= LOSS_REL[arm] * ref["loss"].item()
    for name, g in eng.G.items():
        rg = st[name].grad
        assert (g.cpu() - rg).abs().max().item() <= 1e-3 * rg.abs().max().item() + 1e-12, name


# ----------------------------------------------------------------------------- BASELINE config[3] at full size
def _setting1(precision, input_sec=10.0, sample_rate=16000):
    from adt_str_amd.network import ADTModel, ADTModelConfig
    cfg = ADTModelConfig(input_sec=input_sec, time_res=0.01, win_length=2048, sample_rate=sample_rate, enc_layers=4, dec_layers=4, nhead=6,
                         d_query=128, dropout=0.0, tgt_vocab_size=1400, plain=True, n_mels=128)
    model = ADTModel(cfg).set_precision(precision)
    state = o_adt.seeded_state(model.state_dict(), 0)
    model.load_state_dict(state)
    return model.to(DEV).train(), state


def _config3_batch(B=64, L=160000, T=128, seed=11, sr=16000):
    rng = np.random.default_rng(seed)
    t = np.arange(L, dtype=np.float32) / float(sr)
    wave = np.zeros((B, L), np.float32)
    for b in range(B):                                     # decaying bursts on a noise floor: drum-like, non-trivial spectra
        w = rng.standard_normal(L).astype(np.float32) * 0.01
        for onset in rng.uniform(0, 0.95 * L / sr, 24):
            i0 = int(onset * sr)
            n = min(L - i0, sr // 2)
            w[i0:i0 + n] += (rng.uniform(0.2, 0.9) * np.exp(-t[:n] * rng.uniform(8, 40)) * np.sin(2 * np.pi * rng.uniform(50, 4000) * t[:n])).astype(np.float32)
        wave[b] = np.clip(w, -1, 1)
    lens = rng.integers(32, T + 2, B)
    lens[0] = T + 1
    tokens = np.full((B, T + 1), 1, np.int64)
    for b in range(B):
        n = int(lens[b])
        tokens[b, :n] = np.concatenate([[2], rng.integers(4, 530, n - 2), [3]])
    return {"wavs": wave, "tokens": tokens, "token_lengths": np.where(lens == lens.max(), lens - 1, lens).astype(np.int64)}


def test_config3_full_size_step_both_precisions():
    """BASELINE config[3]: setting-1 network (69.0 M parameters), B = 64 clips of 10 s @ 16 kHz (F = 986 frames), T = 128 target
    positions, dropout 0.  Both precisions: finite, bit-repeatable, and clips 0-1 of the full batch against the oracle run on
    those two clips (the reference's arithmetic is independent across the batch: every op is per clip and the loss is not
    used here)."""
    batch = _config3_batch()
    two = {k: v[:2] for k, v in batch.items()}
    cfg = dict(nhead=6, sample_rate=16000, win_length=2048, time_res=0.01, n_mels=128)
    for precision in ("fp32", "bf16x3", "bf16"):
        model, state = _setting1(precision)
        out = run_engine(model, batch)
        eng = model.engine
        assert out["logits"].shape == (64, 128, 1400) and out["memory"].shape[0] == 64 * 986
        assert bool(torch.isfinite(out["logits"]).all()) and math.isfinite(out["loss"].item()) and bool(torch.isfinite(eng.gflat).all())
        assert float(eng.gflat.abs().max()) > 0
        l1, g1, lg1 = out["loss"].clone(), eng.gflat.clone(), out["logits"].clone()
        again = run_engine(model, batch)
        assert torch.equal(again["loss"], l1) and torch.equal(eng.gflat, g1) and torch.equal(again["logits"], lg1)
        ref = o_adt.compute_loss(state, cfg, two, bf16=(precision == "bf16"))
        if precision != "bf16":
            print(f"config[3] {precision} path vs fp32 oracle, clips 0-1: max |dlogit| / max |logit|",
                  ((lg1[:2].cpu() - ref["logits"]).abs().max() / ref["logits"].abs().max()).item())
            assert_logits_close(lg1[:2], ref["logits"], LOGIT_REL[precision])
        else:
            err = (lg1[:2].cpu() - ref["logits"]).abs().max().item()
            print("config[3] bf16 path vs bf16-operand oracle, clips 0-1: max |dlogit|", err)
            assert err < 6e-2
        del model, eng, out, again
        torch.cuda.empty_cache()


def test_reference_native_operating_point_both_precisions():
    """The reference's OWN operating point (configs/train/setting-1.yaml:9-11: 2.56 s clips @ 24 kHz -> hop 240, F = 246 frames), setting-1
    network, B = 64, T = 128, dropout 0: the same checks as at config[3] -- finite, bit-repeatable, and clips 0-1 of the full batch against
    the oracle run on those two clips, in both precisions.  (A different regime for the kernels: M = 15 744 rows instead of 63 104,
    attention 246 x 246 -- one key block, the two-kernel backward -- and a 24 kHz filterbank.)"""
    sr, L = 24000, 61440
    batch = _config3_batch(L=L, seed=12, sr=sr)
    two = {k: v[:2] for k, v in batch.items()}
    cfg = dict(nhead=6, sample_rate=sr, win_length=2048, time_res=0.01, n_mels=128)
    for precision in ("fp32", "bf16x3", "bf16"):
        model, state = _setting1(precision, input_sec=2.56, sample_rate=sr)
        out = run_engine(model, batch)
        eng = model.engine
        assert out["logits"].shape == (64, 128, 1400) and out["memory"].shape[0] == 64 * 246
        assert bool(torch.isfinite(out["logits"]).all()) and math.isfinite(out["loss"].item()) and bool(torch.isfinite(eng.gflat).all())
        l1, g1, lg1 = out["loss"].clone(), eng.gflat.clone(), out["logits"].clone()
        again = run_engine(model, batch)
        assert torch.equal(again["loss"], l1) and torch.equal(eng.gflat, g1) and torch.equal(again["logits"], lg1)
        ref = o_adt.compute_loss(state, cfg, two, bf16=(precision == "bf16"))
        if precision != "bf16":
            print(f"native operating point, {precision} path vs fp32 oracle, clips 0-1: max |dlogit| / max |logit|",
                  ((lg1[:2].cpu() - ref["logits"]).abs().max() / ref["logits"].abs().max()).item())
            assert_logits_close(lg1[:2], ref["logits"], LOGIT_REL[precision])
        else:
            err = (lg1[:2].cpu() - ref["logits"]).abs().max().item()
            print("native operating point, bf16 path vs bf16-operand oracle, clips 0-1: max |dlogit|", err)
            assert err < 6e-2
        del model, eng, out, again
        torch.cuda.empty_cache()


# ---- the split-bf16 arm's large products on the persistent bf16 kernels (adt_split_bf16x2 / adt_gemm_bf16x3, round 6) -----------------
def test_split_planes_are_the_two_bf16_terms_of_x():
    from adt_str_amd import kernels as K
    x = rnd((200, 136), 3) * torch.logspace(-3, 3, 136, device=DEV)
    pl = K.split_planes(x)
    hi, lo = pl[:, :136].float(), pl[:, 136:].float()
    assert torch.equal(pl[:, :136], x.bfloat16()) and torch.equal(pl[:, 136:], (x - x.bfloat16().float()).bfloat16())
    assert float(((hi + lo) - x).abs().max() / x.abs().max()) < 2.0 ** -16
    pt = K.split_planes(x, transpose=True)                              # the planes of x^T: [cols, 2 rows]
    assert pt.shape == (136, 400) and torch.equal(pt[:, :200], pl[:, :136].T) and torch.equal(pt[:, 200:], pl[:, 136:].T)
    xs = rnd((72, 256), 4)[:, 64:128]                                   # a column slice: row stride != cols
    assert torch.equal(K.split_planes(xs)[:, :64], xs.bfloat16())


@pytest.mark.parametrize("M,N,Kd", [(8192, 2304, 768), (63104, 768, 768), (32768, 3072, 128)])
def test_gemm_bf16x3_on_the_persistent_kernels(M, N, Kd):
    """NT, NN (the data gradient against W) and TN (the weight gradient) through K.gemm on the bf16x3 arm: shapes the persistent kernels
    take must run there (adt_gemm_bf16x3_supported), agree with an fp64 product to the arm's tolerance and with the tiled split kernel."""
    from adt_str_amd import _ffi, kernels as K
    K.set_f32_products("bf16x3")
    lib = _ffi.load()
    a, b = rnd((M, Kd), 1), rnd((N, Kd), 2, 0.05)
    ref = a.double() @ b.double().T
    assert lib.adt_gemm_bf16x3_supported(0, M, N, Kd) == 1 and lib.adt_gemm_bf16x3_supported(1, Kd, N, M) in (0, 1)
    for kw, A, B in (({}, a, b), (dict(b_kn=True), a, b.T.contiguous())):
        got = K.gemm(A, B, **kw)
        os.environ["ADT_X3_TILED"] = "1"
        try:
            tiled = K.gemm(A, B, **kw)
        finally:
            os.environ.pop("ADT_X3_TILED")
        assert float((got.double() - ref).abs().max() / ref.abs().max()) < KERNEL_REL["bf16x3"]
        assert float((got - tiled).abs().max() / ref.abs().max()) < 1e-5
    b2 = rnd((M, N), 5, 0.05)                                            # weight gradient [Kd, N] = a^T b2 over M rows
    gw = K.gemm(a, b2, trans=True)
    refw = a.double().T @ b2.double()
    assert float((gw.double() - refw).abs().max() / refw.abs().max()) < KERNEL_REL["bf16x3"]


def test_gemm_bf16x3_epilogue_with_fp32_side_arrays_and_weight_registry():
    from adt_str_amd import _ffi, kernels as K
    K.set_f32_products("bf16x3")
    M, N, Kd = 16384, 3072, 768
    assert _ffi.load().adt_gemm_bf16x3_supported(0, M, N, Kd) == 1
    a, w, bias = rnd((M, Kd), 1), rnd((N, Kd), 2, 0.05), rnd((N,), 3)
    site = K.drop_site(0.1, 3, 7)

    def ffn1():
        u = torch.empty((M, N), device=DEV)
        return K.gemm(a, w, bias=bias, act=1, act_grad_out=u, drop=site), u
    h, u = ffn1()
    os.environ["ADT_X3_TILED"] = "1"
    try:
        ht, ut = ffn1()
        dy = rnd((M, Kd), 6)
        dut = K.gemm(dy, w, act_grad=ut)
    finally:
        os.environ.pop("ADT_X3_TILED")
    assert float((h - ht).abs().max()) < 1e-5 * float(ht.abs().max()) and float((u - ut).abs().max()) < 1e-5
    assert bool(((h == 0) == (ht == 0))[ht.abs() > 1e-6].all())                      # the same dropout mask
    du = K.gemm(dy, w, act_grad=u)                                                   # the backward multiplies by the fp32 factor as stored
    assert float((du - dut).abs().max()) < 1e-5 * float(dut.abs().max())
    # a registered weight is found by address, row and column slices included, and gives the same bits as splitting the operand on the fly
    plain = K.gemm(a, w[768:1536])
    plain_kn = K.gemm(rnd((M, 768), 8), w[768:1536], b_kn=True)
    K.x3_register_weights([w])
    try:
        assert K._x3_weight_planes(w[768:1536], False) is not None and K._x3_weight_planes(w[768:1536], True) is not None
        assert torch.equal(K.gemm(a, w[768:1536]), plain) and torch.equal(K.gemm(rnd((M, 768), 8), w[768:1536], b_kn=True), plain_kn)
        w.mul_(1.0)                                                                   # an in-place change invalidates the registered planes
        assert K._x3_weight_planes(w, False) is None
    finally:
        K._x3_weights.clear()


def test_gemm_bf16x3_refuses_what_it_cannot_run():
    from adt_str_amd import _ffi, kernels as K
    lib = _ffi.load()
    assert lib.adt_gemm_bf16x3_supported(0, 36, 1400, 768) == 0 and lib.adt_gemm_bf16x3_supported(0, 63104, 768, 100) == 0
    a2, b2 = K.split_planes(rnd((64, 768), 1)), K.split_planes(rnd((1400, 768), 2))
    out = torch.empty((64, 1400), device=DEV)
    ep = _ffi.GemmEpilogue()
    ep.alpha, ep.out_fp32 = 1.0, 1
    with pytest.raises(_ffi.AdtError) as e:
        _ffi.call("adt_gemm_bf16x3", 0, 64, 1400, 768, _ffi.dptr(a2), a2.stride(0), 768, _ffi.dptr(b2), b2.stride(0), 768, _ffi.dptr(out), out.stride(0),
                  __import__("ctypes").byref(ep), None, 0, _ffi.current_stream())
    assert "not supported" in str(e.value)
    ep.side_fp32 = 1                                                     # fp32 side arrays belong to adt_gemm_bf16x3
    x, wt = rnd((64, 768), 1).bfloat16(), rnd((1400, 768), 2).bfloat16()
    with pytest.raises(_ffi.AdtError):
        _ffi.call("adt_gemm_bf16", 0, 64, 1400, 768, _ffi.dptr(x), 768, _ffi.dptr(wt), 768, _ffi.dptr(out), out.stride(0), __import__("ctypes").byref(ep), None, 0,
                  _ffi.current_stream())
