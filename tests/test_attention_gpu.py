"""K4 attention parity on the GPU vs a plain PyTorch fp32 reference of the same op
(softmax(q k^T * scale + additive mask) v on the same bf16-rounded q, k, v).

Tolerances: the kernel rounds P (and dS) to bf16 before the second product and
stores bf16, so |o - ref| <= 2e-2 * max|ref| and gradients within 4e-2 of their
max; log-sum-exp is fp32 (1e-3 absolute on scores that are bf16 products)."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(autouse=True)
def _check_fan_in(monkeypatch):
    """Every one-kernel backward of this module reads the kernel's give-up count back: a wave that stopped waiting for a dQ tile makes the
    call fail (ADT_EHIP) instead of returning an incomplete gradient."""
    monkeypatch.setenv("ADT_ATTN_BWD_CHECK", "1")


def rnd(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(DEV)


def reference(q, k, v, B, H, Sq, Sk, scale, causal, key_len, mask_value=-1e4):
    qh = q.float().view(B, Sq, H, 128).transpose(1, 2)
    kh = k.float().view(B, Sk, H, 128).transpose(1, 2)
    vh = v.float().view(B, Sk, H, 128).transpose(1, 2)
    s = qh @ kh.transpose(-1, -2) * scale
    if causal:
        s = s + torch.triu(torch.ones(Sq, Sk, device=DEV, dtype=torch.bool), 1).float() * mask_value
    if key_len is not None:
        pad = (torch.arange(Sk, device=DEV)[None, :] >= key_len[:, None]).float() * mask_value
        s = s + pad[:, None, None, :]
    p = torch.softmax(s, dim=-1)
    o = (p @ vh).transpose(1, 2).reshape(B * Sq, H * 128)
    return o, torch.logsumexp(s, dim=-1)


CASES = [  # B, H, Sq, Sk, causal, padded
    (2, 2, 128, 128, False, False),
    (1, 1, 32, 64, False, False),
    (2, 6, 986, 986, False, False),      # encoder self-attention shape (10 s clip)
    (3, 6, 128, 986, False, False),      # cross-attention
    (4, 6, 128, 128, True, True),        # decoder self-attention with both additive masks
    (2, 3, 77, 50, True, True),          # ragged sizes
    (1, 2, 1, 200, False, False),        # single query row (greedy decode step)
    (1, 1, 200, 96, False, False),       # 4 query tiles: the first refill of a dK/dV stage, last tile partial
    (2, 2, 257, 300, True, True),        # 5 query tiles (one row in the last), 3 key blocks, both masks
    (1, 1, 192, 33, False, True),        # exactly 3 query tiles (no refill), second key block almost empty
    (1, 2, 449, 64, False, False),       # 8 query tiles (7 full + 1 row): every stage refilled twice
    (1, 3, 100, 2100, False, True),      # 9 key blocks of 256 (the one-kernel backward's fan-in over nine workgroups), 4 query slices, padded keys
    (5, 1, 333, 700, True, True),        # 3 key blocks, 11 query slices, both masks, 5 (batch, head)s: not a multiple of the 8 XCD groups
]


@pytest.mark.parametrize("B,H,Sq,Sk,causal,padded", CASES)
def test_forward_and_backward(B, H, Sq, Sk, causal, padded):
    from adt_str_amd import kernels as k
    d = H * 128
    # packed projections: q from a [B*Sq, 3d] buffer, k/v from a [B*Sk, 3d] buffer (row-strided views)
    qkv_q = rnd((B * Sq, 3 * d), 1).bfloat16()
    qkv_k = qkv_q if Sq == Sk else rnd((B * Sk, 3 * d), 2).bfloat16()
    q, kk, v = qkv_q[:, :d], qkv_k[:, d:2 * d], qkv_k[:, 2 * d:]
    key_len = None
    if padded:
        key_len = torch.tensor([max(1, Sk - 7 * (i + 1)) for i in range(B)], dtype=torch.int32, device=DEV)
    scale = 1.0 / math.sqrt(128)
    o, lse = k.attn_fwd(q, kk, v, B, H, Sq, Sk, scale, causal, key_len)
    qr, kr, vr = (t.float().clone().requires_grad_(True) for t in (q, kk, v))
    ref_o, ref_lse = reference(qr, kr, vr, B, H, Sq, Sk, scale, causal, key_len.long() if padded else None)
    assert (o.float() - ref_o).abs().max() <= 2e-2 * ref_o.abs().max()
    assert (lse - ref_lse).abs().max() <= 2e-3
    dout = rnd((B * Sq, d), 3).bfloat16()
    ref_o.backward(dout.float())
    dqkv_q = torch.zeros_like(qkv_q)
    dqkv_k = dqkv_q if Sq == Sk else torch.zeros_like(qkv_k)
    dq, dk, dv = dqkv_q[:, :d], dqkv_k[:, d:2 * d], dqkv_k[:, 2 * d:]
    k.attn_bwd(q, kk, v, o, dout, lse, dq, dk, dv, B, H, Sq, Sk, scale, causal, key_len)
    for name, got, ref in (("dq", dq, qr.grad), ("dk", dk, kr.grad), ("dv", dv, vr.grad)):
        err = (got.float() - ref).abs().max().item()
        assert err <= 4e-2 * ref.abs().max().item() + 1e-6, f"{name}: {err} vs max {ref.abs().max().item()}"
    # the in-projection's bias gradient = column sums of dq | dk | dv as stored, taken in the kernels' epilogues
    bg = torch.full((3 * d,), float("nan"), device=DEV)
    dq2, dk2, dv2 = torch.zeros_like(dq), torch.zeros_like(dk), torch.zeros_like(dv)
    if Sq == Sk:
        buf = torch.zeros_like(qkv_q)
        dq2, dk2, dv2 = buf[:, :d], buf[:, d:2 * d], buf[:, 2 * d:]
    else:
        bq, bk = torch.zeros_like(qkv_q), torch.zeros_like(qkv_k)
        dq2, dk2, dv2 = bq[:, :d], bk[:, d:2 * d], bk[:, 2 * d:]
    k.attn_bwd(q, kk, v, o, dout, lse, dq2, dk2, dv2, B, H, Sq, Sk, scale, causal, key_len, bias_grad=bg)
    assert torch.equal(dq2, dq) and torch.equal(dk2, dk) and torch.equal(dv2, dv)
    for name, got_sum, mat in (("dq", bg[:d], dq), ("dv", bg[2 * d:], dv)):
        assert (got_sum.double() - mat.double().sum(0)).abs().max().item() <= 2e-6 * mat.float().abs().sum(0).max().item() + 1e-7, name
    # the key bias gradient vanishes identically (rows of dS sum to zero): exact zeros, where summing the stored dk gives noise
    assert not bg[d:2 * d].any()
    assert kr.grad.sum(0).abs().max().item() <= 1e-4 * kr.grad.abs().sum(0).max().item() + 1e-6
    # reproducible: no atomics anywhere
    o2, lse2 = k.attn_fwd(q, kk, v, B, H, Sq, Sk, scale, causal, key_len)
    assert torch.equal(o, o2) and torch.equal(lse, lse2)


# The forward has two workgroup forms (attention_fwd.hip): 4 waves x 2 workgroups per CU (at most 128 queries) and ONE persistent 8-wave
# workgroup per CU that walks its (batch, head, query block) items with the next item's operands prefetched.  The persistent form's item
# switch only happens with more items than CUs, and differs for even / odd numbers of 64-key tiles: enough items here for >= 2 per
# workgroup on a 256-CU chip, both parities, masks, dropout -- against the 4-wave form (same bits) and the fp32 reference.
@pytest.mark.parametrize("B,H,Sq,Sk,causal,padded", [
    (50, 6, 257, 300, False, True),      # 600 items, 5 tiles (odd): the next item's first tiles are fetched after the last tile
    (48, 6, 300, 200, True, True),       # 576 items, 4 tiles (even): K(0), V(0), K(1) of the next item land during the last tile
    (130, 3, 129, 986, False, False),    # 390 items, 16 tiles, the encoder's key count (last tile: 26 keys, no second block)
    (300, 1, 200, 40, False, True),      # 300 items, ONE tile per item
])
def test_persistent_forward_switches_items_correctly(monkeypatch, B, H, Sq, Sk, causal, padded):
    from adt_str_amd import kernels as k
    d = H * 128
    q = rnd((B * Sq, d), 21).bfloat16()
    kv = rnd((B * Sk, 2 * d), 22).bfloat16()
    kk, v = kv[:, :d], kv[:, d:]
    key_len = torch.tensor([max(1, Sk - (3 * i) % max(1, Sk - 1)) for i in range(B)], dtype=torch.int32, device=DEV) if padded else None
    scale = 1.0 / math.sqrt(128)
    ref_o, ref_lse = reference(q, kk, v, B, H, Sq, Sk, scale, causal, key_len.long() if padded else None)
    for drop in (None, (0.1, 4242)):
        out = {}
        for waves in ("4", "8"):
            monkeypatch.setenv("ADT_ATTN_FWD_WAVES", waves)
            out[waves] = k.attn_fwd(q, kk, v, B, H, Sq, Sk, scale, causal, key_len, drop=drop)
            torch.cuda.synchronize()
        monkeypatch.delenv("ADT_ATTN_FWD_WAVES")
        o8, l8 = out["8"]
        assert torch.equal(o8, out["4"][0]) and torch.equal(l8, out["4"][1]), f"4-wave and 8-wave forms differ (dropout {drop})"
        assert (l8 - ref_lse).abs().max() <= 2e-3
        if drop is None:
            assert (o8.float() - ref_o).abs().max() <= 2e-2 * ref_o.abs().max()
        else:          # E[dropout(P)] = P: the mean over all outputs stays put, and 10 % of the weights being zero shows in the spread
            assert abs((o8.float() - ref_o).mean().item()) <= 5e-3 * ref_o.abs().mean().item() + 1e-4
        o8b, l8b = k.attn_fwd(q, kk, v, B, H, Sq, Sk, scale, causal, key_len, drop=drop)
        assert torch.equal(o8b if Sq > 128 else o8, o8) and torch.equal(l8b if Sq > 128 else l8, l8)


@pytest.mark.parametrize("Sq", [100, 200])          # the 4-wave and the 8-wave form of the forward
def test_mild_additive_mask(monkeypatch, Sq):
    """The reference's masks are additive (-1e4, and they add up where causal and padding overlap: model.py:173-181); the kernels take
    the value as a parameter.  With a mild one masked keys keep weight: forward and both backward paths against the fp32 reference."""
    from adt_str_amd import kernels as k
    B, H, Sk, mv = 3, 2, 150, -3.0
    d = H * 128
    q = rnd((B * Sq, d), 31).bfloat16()
    kv = rnd((B * Sk, 2 * d), 32).bfloat16()
    kk, v = kv[:, :d], kv[:, d:]
    key_len = torch.tensor([150, 97, 1], dtype=torch.int32, device=DEV)
    scale = 1.0 / math.sqrt(128)
    o, lse = k.attn_fwd(q, kk, v, B, H, Sq, Sk, scale, True, key_len, mask_value=mv)
    qr, kr, vr = (t.float().clone().requires_grad_(True) for t in (q, kk, v))
    ref_o, ref_lse = reference(qr, kr, vr, B, H, Sq, Sk, scale, True, key_len.long(), mask_value=mv)
    assert (o.float() - ref_o).abs().max() <= 2e-2 * ref_o.abs().max()
    assert (lse - ref_lse).abs().max() <= 2e-3
    dout = rnd((B * Sq, d), 33).bfloat16()
    ref_o.backward(dout.float())
    for mode in ("split", "fused"):
        monkeypatch.setenv("ADT_ATTN_BWD", mode)
        dq, dkv = torch.zeros_like(q), torch.zeros_like(kv)
        k.attn_bwd(q, kk, v, o, dout, lse, dq, dkv[:, :d], dkv[:, d:], B, H, Sq, Sk, scale, True, key_len, mask_value=mv)
        for name, got, ref in (("dq", dq, qr.grad), ("dk", dkv[:, :d], kr.grad), ("dv", dkv[:, d:], vr.grad)):
            err = (got.float() - ref).abs().max().item()
            assert err <= 4e-2 * ref.abs().max().item() + 1e-6, f"{mode} {name}: {err} vs max {ref.abs().max().item()}"
    monkeypatch.delenv("ADT_ATTN_BWD")


def _bwd(k, mode, q, kk, v, o, dout, lse, B, H, Sq, Sk, scale, causal, key_len, drop, monkeypatch, dkv=None, waves=None):
    monkeypatch.setenv("ADT_ATTN_BWD", mode)
    if waves:
        monkeypatch.setenv("ADT_ATTN_BWD_WAVES", waves)
    if dkv:
        monkeypatch.setenv("ADT_ATTN_DKV", dkv)
    d = q.shape[1]
    dq, dkv_ = torch.zeros_like(q), torch.zeros((kk.shape[0], 2 * d), dtype=q.dtype, device=q.device)
    k.attn_bwd(q, kk, v, o, dout, lse, dq, dkv_[:, :d], dkv_[:, d:], B, H, Sq, Sk, scale, causal, key_len, drop=drop)
    torch.cuda.synchronize()
    monkeypatch.delenv("ADT_ATTN_BWD")
    if waves:
        monkeypatch.delenv("ADT_ATTN_BWD_WAVES")
    if dkv:
        monkeypatch.delenv("ADT_ATTN_DKV")
    return dq, dkv_


@pytest.mark.parametrize("B,H,Sq,Sk,causal,padded", CASES)
def test_the_two_backward_paths_agree(monkeypatch, B, H, Sq, Sk, causal, padded):
    """The one-kernel backward (attention_bwd_fused.hip: 5 products, dQ summed over the key-block workgroups by a scheduled fan-in in a
    fixed order) and the two-kernel backward (dQ kernel + dK/dV kernel), forced by ADT_ATTN_BWD, on every shape of CASES (1 to 4 key blocks
    of 256, 1 to 31 query slices, ragged ends, both additive masks), without and with dropout: both regenerate the same dropout masks, so
    they agree to rounding (the summation orders differ); each is bitwise repeatable; without dropout both are checked against the fp32
    reference.  Also the two-kernel path's staggered dK/dV arm (ADT_ATTN_DKV=3), and the one-kernel backward's two forms
    (attention_bwd_fused.hip: 4 waves; attention_bwd_fused8.hip: 8 waves, the default without dropout and with keep bits): the same
    products summed in the same orders, so bit for bit the same gradients."""
    from adt_str_amd import kernels as k
    d = H * 128
    q = rnd((B * Sq, d), 11).bfloat16()
    kv = rnd((B * Sk, 2 * d), 12).bfloat16()
    kk, v = kv[:, :d], kv[:, d:]
    key_len = torch.tensor([max(1, Sk - 7 * (i + 1)) for i in range(B)], dtype=torch.int32, device=DEV) if padded else None
    scale = 1.0 / math.sqrt(128)
    dout = rnd((B * Sq, d), 13).bfloat16()
    for drop in (None, (0.1, 777)):
        o, lse = k.attn_fwd(q, kk, v, B, H, Sq, Sk, scale, causal, key_len, drop=drop)
        args = (q, kk, v, o, dout, lse, B, H, Sq, Sk, scale, causal, key_len, drop, monkeypatch)
        fused, fused2 = _bwd(k, "fused", *args), _bwd(k, "fused", *args)
        split, stag = _bwd(k, "split", *args), _bwd(k, "split", *args, dkv="3")
        assert torch.equal(fused[0], fused2[0]) and torch.equal(fused[1], fused2[1])          # fixed summation order: no atomics anywhere
        assert torch.equal(split[0], stag[0])                                                  # dQ does not depend on the dK/dV arm
        for name, got, ref in (("dq", fused[0], split[0]), ("dkv", fused[1], split[1]), ("dkv staggered", stag[1], split[1])):
            err = (got.float() - ref.float()).abs().max().item()
            assert math.isfinite(err) and err <= 1.5e-2 * ref.float().abs().max().item() + 1e-6, (name, drop, err)
        if drop is not None and Sq > 1:
            # the training step's default: the forward leaves its keep decisions as bits (adt_attn_desc.keep_bits) and the backward --
            # one kernel, no environment switch -- reads them back instead of hashing: the SAME masks, so the hashing one-kernel path's bits
            ob, saved = k.attn_fwd(q, kk, v, B, H, Sq, Sk, scale, causal, key_len, drop=drop, save_bits="force")
            assert isinstance(saved, k.AttnSaved) and torch.equal(ob, o) and torch.equal(saved.lse, lse)
            by_waves = {}
            for waves in ("4", "8", None):
                if waves:
                    monkeypatch.setenv("ADT_ATTN_BWD_WAVES", waves)
                dqb, dkvb = torch.zeros_like(q), torch.zeros_like(kv)
                k.attn_bwd(q, kk, v, ob, dout, saved, dqb, dkvb[:, :d], dkvb[:, d:], B, H, Sq, Sk, scale, causal, key_len, drop=drop)
                if waves:
                    monkeypatch.delenv("ADT_ATTN_BWD_WAVES")
                by_waves[waves] = (dqb, dkvb)
            assert torch.equal(by_waves["4"][0], fused[0]) and torch.equal(by_waves["4"][1], fused[1]), "keep-bits backward differs from the hashing one"
            assert torch.equal(by_waves["8"][0], fused[0]) and torch.equal(by_waves["8"][1], fused[1]), "8-wave keep-bits backward"
            assert torch.equal(by_waves[None][0], by_waves["8"][0]) and torch.equal(by_waves[None][1], by_waves["8"][1])      # the default form, and repeatable
        else:
            four = _bwd(k, "fused", *args, waves="4")
            assert torch.equal(four[0], fused[0]) and torch.equal(four[1], fused[1]), "the one-kernel backward's two forms"
        if drop is None:
            qr, kr, vr = (t.float().clone().requires_grad_(True) for t in (q, kk, v))
            ref_o, _ = reference(qr, kr, vr, B, H, Sq, Sk, scale, causal, key_len.long() if padded else None)
            ref_o.backward(dout.float())
            for path, (gq, gkv) in (("fused", fused), ("split", split)):
                for name, g_, r_ in (("dq", gq, qr.grad), ("dk", gkv[:, :d], kr.grad), ("dv", gkv[:, d:], vr.grad)):
                    assert (g_.float() - r_).abs().max().item() <= 4e-2 * r_.abs().max().item() + 1e-6, (path, name)


def test_keep_bits_from_both_forward_forms_and_long_key_ranges(monkeypatch):
    """(a) The persistent 8-wave forward and the 4-wave form leave the same keep bits (many items per workgroup, an odd and an even number of
    key tiles): the default backward fed either gives the same gradients, which are the hashing path's.  (b) Beyond 16 key blocks per
    head the one-kernel backward's fan-in is not guaranteed to make progress (attention.hip fused_can_run): asked for, the two-kernel path
    runs instead and the result is the two-kernel path's, bits or no bits."""
    from adt_str_amd import kernels as k
    scale = 1.0 / math.sqrt(128)
    for (B, H, Sq, Sk, causal) in [(50, 6, 257, 300, False), (48, 6, 300, 200, True)]:
        d = H * 128
        q, kv, dout = rnd((B * Sq, d), 71).bfloat16(), rnd((B * Sk, 2 * d), 72).bfloat16(), rnd((B * Sq, d), 73).bfloat16()
        kk, v = kv[:, :d], kv[:, d:]
        key_len = torch.tensor([max(1, Sk - (3 * i) % (Sk - 1)) for i in range(B)], dtype=torch.int32, device=DEV)
        drop = (0.1, 99)
        grads = {}
        for waves in ("4", "8"):
            monkeypatch.setenv("ADT_ATTN_FWD_WAVES", waves)
            o, saved = k.attn_fwd(q, kk, v, B, H, Sq, Sk, scale, causal, key_len, drop=drop, save_bits="force")
            dq, dkv = torch.zeros_like(q), torch.zeros_like(kv)
            k.attn_bwd(q, kk, v, o, dout, saved, dq, dkv[:, :d], dkv[:, d:], B, H, Sq, Sk, scale, causal, key_len, drop=drop)
            grads[waves] = (dq, dkv)
        monkeypatch.delenv("ADT_ATTN_FWD_WAVES")
        hashed = _bwd(k, "fused", q, kk, v, o, dout, saved.lse, B, H, Sq, Sk, scale, causal, key_len, drop, monkeypatch)
        for w in ("4", "8"):        # (the default backward with bits is the 8-wave form, the hashing one the 4-wave form: same bits)
            assert torch.equal(grads[w][0], hashed[0]) and torch.equal(grads[w][1], hashed[1]), f"{w}-wave forward's bits"
    B, H, Sq, Sk = 1, 1, 64, 4400                                       # 18 key blocks of 256
    q, kv, dout = rnd((B * Sq, 128), 81).bfloat16(), rnd((B * Sk, 256), 82).bfloat16(), rnd((B * Sq, 128), 83).bfloat16()
    kk, v = kv[:, :128], kv[:, 128:]
    o, saved = k.attn_fwd(q, kk, v, B, H, Sq, Sk, scale, drop=(0.1, 5), save_bits="force")
    args = (q, kk, v, o, dout, saved.lse, B, H, Sq, Sk, scale, False, None, (0.1, 5), monkeypatch)
    split, asked = _bwd(k, "split", *args), _bwd(k, "fused", *args)
    assert torch.equal(split[0], asked[0]) and torch.equal(split[1], asked[1])
    dq, dkv = torch.zeros_like(q), torch.zeros_like(kv)
    k.attn_bwd(q, kk, v, o, dout, saved, dq, dkv[:, :128], dkv[:, 128:], B, H, Sq, Sk, scale, drop=(0.1, 5))
    assert torch.equal(dq, split[0]) and torch.equal(dkv, split[1])


def test_fused_backward_reuses_a_dirty_workspace(monkeypatch):
    """The fan-in's tiles and flags live in the caller's workspace, which the next call reuses as it is: flags are zeroed by the launcher,
    stale tiles of another shape / other data must never be read (they are published write-through and read past L1).  Alternate two
    problems with different data on one workspace and compare each result with its first run."""
    from adt_str_amd import kernels as k
    scale = 1.0 / math.sqrt(128)
    probs = []
    for seed, (B, H, Sq, Sk) in enumerate([(2, 6, 986, 986), (3, 2, 300, 700)]):
        d = H * 128
        q = rnd((B * Sq, d), 40 + seed).bfloat16()
        kv = rnd((B * Sk, 2 * d), 50 + seed).bfloat16()
        dout = rnd((B * Sq, d), 60 + seed).bfloat16()
        o, lse = k.attn_fwd(q, kv[:, :d], kv[:, d:], B, H, Sq, Sk, scale)
        probs.append((q, kv[:, :d], kv[:, d:], o, dout, lse, B, H, Sq, Sk, scale, False, None, None, monkeypatch))
    first = [_bwd(k, "fused", *p) for p in probs]
    for _ in range(3):
        for p, f in zip(probs, first):
            g = _bwd(k, "fused", *p)
            assert torch.equal(g[0], f[0]) and torch.equal(g[1], f[1])


def test_exact_selector():
    """One-hot keys and integer values: P is exactly one-hot after softmax of a huge score gap,
    so the output must equal the selected V row bit for bit (catches any key/lane permutation)."""
    from adt_str_amd import kernels as k
    B, H, S = 1, 1, 128
    q = torch.zeros((S, 128), device=DEV)
    kk = torch.zeros((S, 128), device=DEV)
    perm = torch.randperm(S, generator=torch.Generator().manual_seed(0)).to(DEV)
    q[torch.arange(S), torch.arange(S) % 128] = 64.0                    # query i looks along axis i
    kk[perm, torch.arange(S) % 128] = 64.0                              # key perm[i] answers it
    v = torch.randint(-8, 9, (S, 128), generator=torch.Generator().manual_seed(1)).float().to(DEV)
    o, _ = k.attn_fwd(q.bfloat16(), kk.bfloat16(), v.bfloat16(), B, H, S, S, 1.0)
    assert torch.equal(o.float(), v[perm])


def test_rejects_other_head_dims():
    from adt_str_amd import _ffi
    import ctypes as C
    d = _ffi.AttnDesc()
    d.batch, d.heads, d.q_len, d.k_len, d.head_dim = 1, 1, 8, 8, 64
    d.ldq = d.ldk = d.ldv = d.ldo = 64
    x = torch.zeros(64, device=DEV)
    with pytest.raises(_ffi.AdtError) as e:
        _ffi.call("adt_attn_fwd", C.byref(d), x.data_ptr(), x.data_ptr(), x.data_ptr(), x.data_ptr(), x.data_ptr(), 0)
    assert e.value.code == -2


@pytest.mark.parametrize("B,H,Sk,lens", [(3, 2, 1, None), (2, 6, 63, None), (2, 3, 64, [64, 1]), (4, 6, 65, [65, 7, 64, 1]), (8, 6, 986, None),
                                         (2, 2, 1500, [1500, 1025]), (8, 6, 1000, [1, 2, 17, 64, 65, 500, 999, 1000])])
def test_single_query_decode_kernel(monkeypatch, B, H, Sk, lens):
    """One query per (batch, head) -- a step of the KV-cached greedy decode -- takes the decode kernel (lane <-> key scores, sixteen
    waves over the keys, keys behind the padding length skipped: their weight underflows to exactly 0 under the reference's -1e4 mask);
    against the fp32 reference and against the tiled kernel (ADT_ATTN_NO_DECODE=1), with the cache's packed [B, Tmax, 3d] layout."""
    from adt_str_amd import kernels as k
    d = H * 128
    cache = rnd((B * Sk, 3 * d), 7).bfloat16()
    q = rnd((B, 3 * d), 8).bfloat16()[:, :d]
    kk, v = cache[:, d:2 * d], cache[:, 2 * d:]
    key_len = None if lens is None else torch.tensor(lens, dtype=torch.int32, device=DEV)
    scale = 1 / math.sqrt(128)
    o, lse = k.attn_fwd(q, kk, v, B, H, 1, Sk, scale, key_len=key_len)
    ref_o, ref_lse = reference(q, kk, v, B, H, 1, Sk, scale, False, key_len)
    assert (o.float() - ref_o).abs().max() < 1e-2 and (lse.view(B, H, 1) - ref_lse).abs().max() < 2e-3
    monkeypatch.setenv("ADT_ATTN_NO_DECODE", "1")
    o2, lse2 = k.attn_fwd(q, kk, v, B, H, 1, Sk, scale, key_len=key_len)
    assert (o.float() - o2.float()).abs().max() < 1e-2 and (lse - lse2).abs().max() < 2e-3
    # a mild mask value keeps the masked keys in the softmax: every key is visited
    if lens is not None:
        monkeypatch.delenv("ADT_ATTN_NO_DECODE")
        o3, _ = k.attn_fwd(q, kk, v, B, H, 1, Sk, scale, key_len=key_len, mask_value=-2.0)
        ref3, _ = reference(q, kk, v, B, H, 1, Sk, scale, False, key_len, mask_value=-2.0)
        assert (o3.float() - ref3).abs().max() < 1e-2


def test_single_query_attention_is_invariant_to_the_order_of_the_keys():
    """A size-independent property at the decode step's full size (B = 8, 6 heads, 986 memory keys): permuting the keys together
    with their values leaves the output unchanged up to fp32 summation order, and the log-sum-exp as well."""
    from adt_str_amd import kernels as k
    B, H, Sk = 8, 6, 986
    d = H * 128
    kv = rnd((B * Sk, 2 * d), 21).bfloat16()
    q = rnd((B, d), 22).bfloat16()
    scale = 1 / math.sqrt(128)
    o1, lse1 = k.attn_fwd(q, kv[:, :d], kv[:, d:], B, H, 1, Sk, scale)
    perm = torch.stack([torch.randperm(Sk, generator=torch.Generator().manual_seed(b)) + b * Sk for b in range(B)]).reshape(-1).to(DEV)
    kv2 = kv[perm].contiguous()
    o2, lse2 = k.attn_fwd(q, kv2[:, :d], kv2[:, d:], B, H, 1, Sk, scale)
    assert (o1.float() - o2.float()).abs().max() <= 2e-2 * o1.float().abs().max() + 1e-3      # bf16 outputs, bf16-rounded probabilities
    assert (lse1 - lse2).abs().max() < 1e-4
