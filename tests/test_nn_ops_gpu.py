"""K6/K7/K8 + optimizer kernels vs plain PyTorch fp32 references of the same op
(floating-point kernels: tolerances stated per test)."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def rnd(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(DEV)


@pytest.mark.parametrize("M,D", [(1, 768), (37, 768), (4096, 768), (63, 32), (130, 1024), (1001, 96), (77, 128), (50, 64), (33, 192)])
def test_layernorm_fwd_bwd(M, D):
    from adt_str_amd import kernels as k
    x = rnd((M, D), 1, 2.0) + 0.5
    gamma, beta = 1 + rnd((D,), 2, 0.1), rnd((D,), 3, 0.1)
    dy = rnd((M, D), 4)
    xr = x.clone().requires_grad_(True); gr = gamma.clone().requires_grad_(True); br = beta.clone().requires_grad_(True)
    ref = F.layer_norm(xr, (D,), gr, br, 1e-5)
    ref.backward(dy)
    y32, y16, mean, rstd = k.layernorm_fwd(x, gamma, beta)
    assert (y32 - ref).abs().max() < 2e-5
    assert (y16.float() - ref).abs().max() < 2e-2
    dg, db, dxs = (torch.empty(D, device=DEV) for _ in range(3))
    dx32, dx16 = k.layernorm_bwd(dy, x, gamma, mean, rstd, dg, db, dxs)
    assert (dx32 - xr.grad).abs().max() < 5e-5 * max(1.0, xr.grad.abs().max().item())
    assert (dx16.float() - xr.grad).abs().max() < 2e-2 * max(1.0, xr.grad.abs().max().item())
    tol = 2e-4 * math.sqrt(M) + 1e-4
    assert (dg - gr.grad).abs().max() < tol and (db - br.grad).abs().max() < tol
    assert (dxs - xr.grad.sum(0)).abs().max() < tol


@pytest.mark.parametrize("M,N", [(5, 8), (1000, 768), (8192, 2304), (300, 1400)])
def test_colsum(M, N):
    from adt_str_amd import kernels as k
    x = rnd((M, N), 5).bfloat16()
    got = k.colsum(x)
    ref = x.float().sum(0)
    assert (got - ref).abs().max() < 1e-4 * math.sqrt(M) + 1e-5


def test_embed_pe():
    from adt_str_amd import kernels as k
    from oracle.adt import positional_encoding
    B, T, V, D = 7, 33, 1400, 768
    g = torch.Generator().manual_seed(0)
    tokens = torch.randint(0, V, (B, T), generator=g).to(DEV)
    table = rnd((V, D), 6, 0.05)
    pe = positional_encoding(D)[0].to(DEV)
    y32, y16 = k.embed_pe_fwd(tokens, table, pe, math.sqrt(D))
    ref = (table[tokens] * math.sqrt(D) + pe[:T]).reshape(B * T, D)
    assert (y32 - ref).abs().max() < 1e-6 * ref.abs().max()
    assert (y16.float() - ref).abs().max() < 1e-2 * ref.abs().max()
    dy = rnd((B * T, D), 7)
    dtab = torch.full_like(table, float("nan"))                       # overwritten, not accumulated into
    k.embed_bwd(tokens, dy, math.sqrt(D), dtab)
    # the sum runs as onehot^T . bf16(scale * dy) on the TN GEMM: exact against the bf16-rounded rows, 2^-8 against fp32 ones
    rows16 = (dy * math.sqrt(D)).bfloat16().double()
    ref16 = torch.zeros_like(table, dtype=torch.float64).index_add_(0, tokens.reshape(-1), rows16)
    assert (dtab.double() - ref16).abs().max() < 1e-5 * ref16.abs().max()
    ref_d = torch.zeros_like(table).index_add_(0, tokens.reshape(-1), dy * math.sqrt(D))
    assert (dtab - ref_d).abs().max() < 6e-3 * ref_d.abs().max()
    # a few tokens take most rows (as drum vocabularies do), out-of-range ids clamp; two runs agree bit for bit
    n = 8192
    g = torch.Generator().manual_seed(3)
    skew = torch.where(torch.rand(n, generator=g) < 0.8, torch.randint(0, 3, (n,), generator=g), torch.randint(-2, V + 2, (n,), generator=g)).to(DEV)
    dy2 = rnd((n, D), 9)
    d1, d2 = torch.empty_like(table), torch.empty_like(table)
    k.embed_bwd(skew, dy2, 2.0, d1)
    k.embed_bwd(skew, dy2, 2.0, d2)
    assert torch.equal(d1, d2)
    ref2 = torch.zeros_like(table, dtype=torch.float64).index_add_(0, skew.clamp(0, V - 1), (dy2 * 2.0).bfloat16().double())
    assert (d1.double() - ref2).abs().max() < 1e-5 * ref2.abs().max()
    # vocabularies that are not a multiple of 8 take the fp32-atomics kernel
    t3 = torch.randint(0, 1399, (B * T,), generator=g).to(DEV)
    d3 = torch.empty((1399, D), device=DEV)
    k.embed_bwd(t3, dy, 1.5, d3)
    ref3 = torch.zeros((1399, D), device=DEV).index_add_(0, t3, dy * 1.5)
    assert (d3 - ref3).abs().max() < 1e-4 * ref3.abs().max()


@pytest.mark.parametrize("M,V", [(8192, 1400), (5, 1400), (64, 17)])
def test_cross_entropy(M, V):
    from adt_str_amd import kernels as k
    logits = rnd((M, V), 8, 3.0)
    g = torch.Generator().manual_seed(1)
    labels = torch.randint(0, V, (M,), generator=g).to(DEV)
    labels[::3] = 1                                                   # ignored rows
    lr = logits.clone().requires_grad_(True)
    ref = F.cross_entropy(torch.nan_to_num(lr, nan=0.0, posinf=1e4, neginf=-1e4), labels, ignore_index=1)
    ref.backward()
    loss, dl = k.cross_entropy(logits, labels)
    assert abs(loss.item() - ref.item()) < 2e-5 * abs(ref.item())
    assert (dl.float() - lr.grad).abs().max() < 1e-2 * lr.grad.abs().max()
    assert torch.all(dl[::3] == 0)


def test_cross_entropy_nonfinite_and_all_ignored():
    from adt_str_amd import kernels as k
    logits = rnd((6, 50), 9)
    logits[0, 3] = float("nan"); logits[1, 4] = float("inf"); logits[2, 5] = float("-inf")
    labels = torch.tensor([2, 4, 7, 1, 9, 11], device=DEV)
    ref = F.cross_entropy(torch.nan_to_num(logits, nan=0.0, posinf=1e4, neginf=-1e4), labels, ignore_index=1)
    loss, _ = k.cross_entropy(logits, labels)
    assert abs(loss.item() - ref.item()) < 1e-3 * abs(ref.item())
    loss2, dl2 = k.cross_entropy(logits, torch.ones(6, dtype=torch.int64, device=DEV))
    assert math.isnan(loss2.item()) and torch.all(dl2 == 0)           # torch: mean over zero kept rows = nan


def test_cast_and_transpose():
    from adt_str_amd import kernels as k
    x = rnd((1400, 768), 10)
    y, yt = k.cast_bf16(x, True, True)
    assert torch.equal(y, x.bfloat16()) and torch.equal(yt, x.bfloat16().t().contiguous())
    x2 = rnd((77, 130), 11)
    y2, yt2 = k.cast_bf16(x2, True, True)
    assert torch.equal(y2, x2.bfloat16()) and torch.equal(yt2, x2.bfloat16().t().contiguous())


def test_batched_cast_table():
    """One launch for a whole list of weights: every row-major and transposed bf16 copy equals the single-matrix path."""
    from adt_str_amd import kernels as k
    ws = [rnd(shape, 20 + i) for i, shape in enumerate([(768, 128), (2304, 768), (1400, 768), (64, 64), (77, 136), (3072, 768)])]
    tab = k.CastTable(ws)
    tab.run()
    for w, y, yt in zip(ws, tab.y, tab.y_t):
        assert torch.equal(y, w.bfloat16()) and torch.equal(yt, w.bfloat16().t().contiguous())
    ws[1].mul_(2.0)                                       # in-place update (optimizer step): same pointers, same table
    assert tab.matches(ws)
    tab.run()
    assert torch.equal(tab.y[1], ws[1].bfloat16())
    assert not tab.matches([w.clone() for w in ws])


def test_grad_norm_and_adamw_match_torch():
    from adt_str_amd import kernels as k
    n = 1_000_003
    p0, g0 = rnd((n,), 12, 0.1), rnd((n,), 13, 0.01)
    ref_p = torch.nn.Parameter(p0.clone())
    opt = torch.optim.AdamW([ref_p], lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2)
    p, m, v = p0.clone(), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    p16 = torch.empty(n, dtype=torch.bfloat16, device=DEV)
    for step in range(1, 4):
        g = g0 * step
        ref_p.grad = g.clone()
        nrm_ref = torch.nn.utils.clip_grad_norm_([ref_p], 1.0)
        opt.step()
        nc = k.grad_norm(g, 1.0)
        assert abs(nc[0].item() - nrm_ref.item()) < 1e-4 * nrm_ref.item()
        k.adamw_step(p, g, m, v, step, 1e-3, 0.9, 0.999, 1e-8, 1e-2, nc, p16)
        assert (p - ref_p.data).abs().max() < 2e-6
    assert torch.equal(p16, p.bfloat16())


def test_greedy_step_is_the_torch_tail():
    """adt_greedy_step against the tensor ops it replaces in the decode step (argmax with torch's tie / NaN rules, finished rows
    keep the end token, done_at latches the column count once every row has finished)."""
    from adt_str_amd import kernels as k
    B, V, Tmax, end = 9, 1400, 12, 3
    g = torch.Generator().manual_seed(0)
    gen = torch.full((B, Tmax), end, dtype=torch.long, device=DEV)
    st = dict(t=torch.zeros(1, dtype=torch.long, device=DEV), tok=torch.zeros((B, 1), dtype=torch.long, device=DEV),
              klen=torch.ones(B, dtype=torch.int32, device=DEV), finished=torch.zeros(B, dtype=torch.bool, device=DEV),
              done_at=torch.full((1,), Tmax, dtype=torch.long, device=DEV))
    ref = {n: v.clone() for n, v in st.items()}
    ref_gen = gen.clone()
    eos = torch.full((B,), end, dtype=torch.long, device=DEV)
    for step in range(Tmax - 1):
        logits = torch.randn((B, V + 8), generator=g).to(DEV)[:, :V]                 # a strided view
        logits[0, 7] = logits[0, 900] = 50.0                                         # tie: the first index wins
        if step == 1:
            logits[1, 1234] = float("nan")                                           # NaN counts as the maximum
        if step >= 2:
            logits[2:, end] = 100.0                                                  # rows 2.. finish
        if step >= 5:
            logits[:2, end] = 100.0                                                  # ... and then the rest
        k.greedy_step(logits, st["finished"], gen, st["t"], st["tok"], st["klen"], st["done_at"], end)
        nxt = torch.where(ref["finished"], eos, torch.argmax(logits, dim=-1))
        ref["t"].add_(1)
        ref_gen.index_copy_(1, ref["t"], nxt.unsqueeze(1))
        ref["finished"].logical_or_(nxt == end)
        all_done = ref["finished"].all() & (ref["done_at"] == Tmax)
        ref["done_at"].copy_(torch.where(all_done, ref["t"] + 1, ref["done_at"]))
        ref["tok"].copy_(nxt.unsqueeze(1))
        ref["klen"].add_(1)
        assert torch.equal(gen, ref_gen), step
        for n in st:
            assert torch.equal(st[n], ref[n]), (n, step)
    assert int(st["done_at"]) == 7 and int(gen[0, 1]) == 7 and int(gen[1, 2]) == 1234
