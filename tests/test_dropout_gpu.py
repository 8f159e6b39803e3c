"""Dropout: every kernel that applies a mask (GEMM epilogue, LayerNorm, embedding, attention) and the
whole network forward/backward against references that apply the SAME masks (oracle/dropout.py mirrors
the kernels' counter-based hash).  Shows that masks are regenerated consistently in the backward."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import adt as o_adt
from oracle import dropout as o_drop

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def rnd(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(DEV)


def test_mask_statistics_and_host_mirror():
    from adt_str_amd import kernels as k
    M, D = 512, 768
    x = torch.ones((M, D), device=DEV)
    site = k.drop_site(0.1, 7, 3)
    y32, _, _, _ = k.layernorm_fwd(x + rnd((M, D), 1), torch.ones(D, device=DEV), torch.zeros(D, device=DEV), drop=site)
    ref_noscale, _, _, _ = k.layernorm_fwd(x + rnd((M, D), 1), torch.ones(D, device=DEV), torch.zeros(D, device=DEV))
    sc = o_drop.scale((M, D), *site).to(DEV)
    assert torch.equal(y32, ref_noscale * sc)
    keep = (sc != 0).float().mean().item()
    assert abs(keep - 0.9) < 5e-3
    assert k.drop_site(0.0, 7, 3) is None
    other = o_drop.scale((M, D), *k.drop_site(0.1, 8, 3))
    assert (other != sc.cpu()).float().mean() > 0.1                      # a new step seed gives a new mask


def test_gemm_dropout_positions():
    from adt_str_amd import kernels as k
    M, N, Kd = 256, 384, 128
    a, b = rnd((M, Kd), 1).bfloat16(), rnd((N, Kd), 2, 0.1).bfloat16()
    bias, res = rnd((N,), 3), rnd((M, N), 4)
    z = a.float() @ b.float().t() + bias
    site = k.drop_site(0.25, 11, 5)
    sc = o_drop.scale((M, N), *site).to(DEV)
    before = k.gemm(a, b, bias=bias, residual=res, out_dtype=torch.float32, drop=site)
    after = k.gemm(a, b, bias=bias, residual=res, out_dtype=torch.float32, drop=site, drop_after_residual=True)
    assert (before - (z * sc + res)).abs().max() < 1e-4 * z.abs().max()
    assert (after - (z + res) * sc).abs().max() < 1e-4 * z.abs().max()
    u = torch.empty((M, N), dtype=torch.bfloat16, device=DEV)
    h = k.gemm(a, b, bias=bias, act=1, pre_act_out=u, drop=site)
    ref_h = F.gelu(u.float()) * sc
    assert (h.float() - ref_h).abs().max() <= 8e-3 * ref_h.abs().max()          # bf16 output rounding (2^-8 relative)
    # slow-path kernel (K not a multiple of 64) applies the same mask
    a2, b2 = rnd((M, 72), 5).bfloat16(), rnd((N, 72), 6).bfloat16()
    got = k.gemm(a2, b2, out_dtype=torch.float32, drop=site)
    assert (got - (a2.float() @ b2.float().t()) * sc).abs().max() < 1e-3


def test_layernorm_backward_masks():
    from adt_str_amd import kernels as k
    M, D = 300, 768
    x, dy = rnd((M, D), 1, 2.0), rnd((M, D), 2)
    g, b = 1 + rnd((D,), 3, 0.1), rnd((D,), 4, 0.1)
    s_out, s_dx = k.drop_site(0.1, 1, 1), k.drop_site(0.1, 1, 2)
    m_out, m_dx = o_drop.scale((M, D), *s_out).to(DEV), o_drop.scale((M, D), *s_dx).to(DEV)
    xr = x.clone().requires_grad_(True)
    (F.layer_norm(xr, (D,), g, b) * m_out).backward(dy)
    _, _, mean, rstd = k.layernorm_fwd(x, g, b, drop=s_out)
    dxs = torch.empty(D, device=DEV)
    dx32, dx16 = k.layernorm_bwd(dy, x, g, mean, rstd, None, None, dxs, dy_drop=s_out, dx16_drop=s_dx)
    assert (dx32 - xr.grad).abs().max() < 1e-4
    assert (dx16.float() - xr.grad * m_dx).abs().max() < 3e-2
    assert (dxs - (xr.grad * m_dx).sum(0)).abs().max() < 5e-3


def test_embedding_dropout():
    from adt_str_amd import kernels as k
    B, T, V, D = 5, 17, 1400, 256
    tokens = torch.randint(0, V, (B, T), generator=torch.Generator().manual_seed(0)).to(DEV)
    table, pe = rnd((V, D), 1, 0.05), o_adt.positional_encoding(D)[0].to(DEV)
    site = k.drop_site(0.1, 3, 9)
    sc = o_drop.scale((B * T, D), *site).to(DEV)
    y32, _ = k.embed_pe_fwd(tokens, table, pe, math.sqrt(D), drop=site)
    ref = (table[tokens] * math.sqrt(D) + pe[:T]).reshape(B * T, D) * sc
    assert (y32 - ref).abs().max() < 1e-5
    dy = rnd((B * T, D), 2)
    dtab = torch.zeros_like(table)
    k.embed_bwd(tokens, dy, math.sqrt(D), dtab, drop=site)
    ref_d = torch.zeros_like(table, dtype=torch.float64).index_add_(0, tokens.reshape(-1), (dy * sc * math.sqrt(D)).bfloat16().double())
    assert (dtab.double() - ref_d).abs().max() < 1e-5 * ref_d.abs().max()      # bf16 gradient rows, summed on the TN GEMM


@pytest.mark.parametrize("B,H,Sq,Sk,causal", [(2, 2, 128, 128, False), (2, 3, 77, 150, False), (2, 2, 96, 96, True), (1, 2, 300, 200, False), (1, 1, 449, 70, True)])
def test_attention_dropout_forward_backward(B, H, Sq, Sk, causal):
    from adt_str_amd import kernels as k
    d = H * 128
    q, kk, v = (rnd((B * n, d), s).bfloat16() for n, s in ((Sq, 1), (Sk, 2), (Sk, 3)))
    site = k.drop_site(0.2, 5, 4)
    sc = o_drop.scale((B, H, Sq, Sk), *site).to(DEV)
    scale = 1 / math.sqrt(128)
    qr, kr, vr = (t.float().clone().requires_grad_(True) for t in (q, kk, v))
    qh, kh, vh = (t.view(B, -1, H, 128).transpose(1, 2) for t in (qr, kr, vr))
    s = qh @ kh.transpose(-1, -2) * scale
    if causal:
        s = s + torch.triu(torch.ones(Sq, Sk, device=DEV), 1) * -1e4
    ref = ((torch.softmax(s, -1) * sc) @ vh).transpose(1, 2).reshape(B * Sq, d)
    o, lse = k.attn_fwd(q, kk, v, B, H, Sq, Sk, scale, causal, drop=site)
    assert (o.float() - ref).abs().max() <= 2e-2 * ref.abs().max()
    dout = rnd((B * Sq, d), 4).bfloat16()
    ref.backward(dout.float())
    dq, dk, dv = torch.empty_like(q), torch.empty_like(kk), torch.empty_like(v)
    k.attn_bwd(q, kk, v, o, dout, lse, dq, dk, dv, B, H, Sq, Sk, scale, causal, drop=site)
    for name, got, rg in (("dq", dq, qr.grad), ("dk", dk, kr.grad), ("dv", dv, vr.grad)):
        assert (got.float() - rg).abs().max() <= 4e-2 * rg.abs().max(), name
    # the training step's path: the forward leaves its keep decisions as bits and the one-kernel backward reads them back.  The bits ARE
    # the oracle's mask (every element, both forward forms write them alike), and the gradients are the same-mask reference's.
    o2, saved = k.attn_fwd(q, kk, v, B, H, Sq, Sk, scale, causal, drop=site, save_bits="force")
    assert torch.equal(o2, o) and torch.equal(saved.lse, lse)
    assert torch.equal(k.keep_bits_to_mask(saved.bits, B, H, Sq, Sk), sc > 0)
    dq2, dk2, dv2 = torch.empty_like(q), torch.empty_like(kk), torch.empty_like(v)
    k.attn_bwd(q, kk, v, o2, dout, saved, dq2, dk2, dv2, B, H, Sq, Sk, scale, causal, drop=site)
    for name, got, rg in (("dq", dq2, qr.grad), ("dk", dk2, kr.grad), ("dv", dv2, vr.grad)):
        assert (got.float() - rg).abs().max() <= 4e-2 * rg.abs().max(), name + " (keep bits)"


def test_network_with_dropout_matches_oracle_with_same_masks():
    """dropout 0.1 (the reference's shipped value): logits, loss and every parameter gradient against the
    oracle evaluated with the masks of this very step."""
    from adt_str_amd import kernels as k
    from tests.test_network_gpu import make_batch
    from adt_str_amd.network import ADTModel, ADTModelConfig
    cfg = ADTModelConfig(input_sec=0.5, time_res=0.01, win_length=2048, sample_rate=16000, enc_layers=2, dec_layers=2, nhead=2,
                         d_query=128, dropout=0.1, tgt_vocab_size=1400, plain=True, n_mels=128)
    model = ADTModel(cfg)
    state = o_adt.seeded_state(model.state_dict(), 0)
    model.load_state_dict(state)
    model = model.to(DEV).train()
    batch = make_batch(3, 8000, 12, 1)
    tok = torch.from_numpy(batch["tokens"]).to(DEV)
    T = tok.shape[1] - 1
    pad = (torch.arange(T)[None, :] >= torch.from_numpy(batch["token_lengths"])[:, None]).to(DEV)
    eng = model.engine
    out = eng.loss_and_grads(torch.from_numpy(batch["wavs"]).to(DEV), tok[:, :-1], pad, tok[:, 1:], want_grads=True, return_logits=True)
    seed, sites = eng.drop_seed, dict(eng._sites)
    assert len(sites) == 1 + 2 * 4 + 1 + 1 + 2 * 6                       # every dropout module of the reference has a site

    def drop(site):
        key = k.drop_site(0.1, seed, sites[site])
        return lambda shape: o_drop.scale(tuple(shape), *key)

    st = {kk: (v.clone().requires_grad_(True) if v.is_floating_point() and "pos_embedding" not in kk and "compute_spec" not in kk else v)
          for kk, v in state.items()}
    ocfg = dict(nhead=2, sample_rate=16000, win_length=2048, time_res=0.01, n_mels=128)
    ref = o_adt.compute_loss(st, ocfg, batch, drop=drop)
    ref["loss"].backward()
    nodrop = o_adt.compute_loss(state, ocfg, batch)
    assert (ref["logits"].detach() - nodrop["logits"]).abs().max() > 0.1   # the masks really change the result
    assert (out["logits"].cpu() - ref["logits"].detach()).abs().max() < 8e-2
    assert abs(out["loss"].item() - ref["loss"].item()) < 1e-2 * ref["loss"].item()
    for name, g in eng.G.items():
        rg = st[name].grad
        rel = (g.cpu() - rg).abs().max().item() / (rg.abs().max().item() + 1e-12)
        assert rel < 6e-2, f"{name}: grad rel err {rel}"
    # the same step again (same masks): every reduction on the path has a fixed order, so loss and gradients repeat bit for bit
    g1, loss1 = eng.gflat.clone(), out["loss"].clone()
    eng.drop_seed = seed - 1
    again = eng.loss_and_grads(torch.from_numpy(batch["wavs"]).to(DEV), tok[:, :-1], pad, tok[:, 1:], want_grads=True)
    assert eng.drop_seed == seed and torch.equal(again["loss"], loss1) and torch.equal(eng.gflat, g1)
    # a second step draws new masks; eval mode has none
    out2 = eng.loss_and_grads(torch.from_numpy(batch["wavs"]).to(DEV), tok[:, :-1], pad, tok[:, 1:], want_grads=False)
    assert eng.drop_seed == seed + 1 and abs(out2["loss"].item() - out["loss"].item()) > 1e-6
    model.eval()
    out3 = eng.loss_and_grads(torch.from_numpy(batch["wavs"]).to(DEV), tok[:, :-1], pad, tok[:, 1:], want_grads=False)
    assert abs(out3["loss"].item() - nodrop["loss"].item()) < 1e-2 * nodrop["loss"].item()
