"""K3/K5 GEMM parity on the GPU.  The check is against an fp32 torch matmul of the
same bf16-rounded operands (the kernel accumulates in fp32), so the tolerance only
has to cover accumulation order: |err| <= 2e-3 * sqrt(K) * scale for bf16 outputs
(bf16 rounding of the result, 2^-8 relative) and 1e-4 relative for fp32 outputs."""
import math

import numpy as np

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def rnd(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(DEV)


def close(got, ref, rel, what=""):
    err = (got.float() - ref).abs().max().item()
    tol = rel * ref.abs().max().item() + 1e-6
    assert err <= tol, f"{what}: max err {err} > {tol}"


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (256, 384, 128), (300, 200, 192), (63104 // 8, 768, 768),
                                   (1000, 1400, 768), (64, 2304, 768), (8192, 768, 3072), (77, 24, 64),
                                   (4096, 288, 96), (1000, 96, 96), (513, 384, 32), (300, 96, 160), (200, 40, 72)])
def test_nt_plain(M, N, K):
    from adt_str_amd import kernels as k
    a, b = rnd((M, K), 1).bfloat16(), rnd((N, K), 2).bfloat16()
    ref = a.float() @ b.float().t()
    close(k.gemm(a, b, out_dtype=torch.float32), ref, 1e-4, "fp32 out")
    close(k.gemm(a, b), ref, 6e-3, "bf16 out")


def test_nt_exact_integers_asymmetric():
    """A = I-like selector and small integers: any row/col swap or k mix-up shows up exactly."""
    from adt_str_amd import kernels as k
    M, N, K = 256, 256, 128
    g = torch.Generator().manual_seed(0)
    a = torch.randint(-3, 4, (M, K), generator=g).float().to(DEV).bfloat16()
    b = torch.randint(-2, 3, (N, K), generator=g).float().to(DEV).bfloat16()
    out = k.gemm(a, b, out_dtype=torch.float32)
    assert torch.equal(out, a.float() @ b.float().t())


@pytest.mark.parametrize("K,M,N", [(64, 128, 128), (256, 128, 256), (1000, 200, 136), (63104 // 4, 768, 768),
                                   (8192, 1400, 768), (2048, 768, 3072), (70, 40, 24), (48 * 986, 768, 3072), (16 * 986 + 8, 2304, 768)])
def test_tn_wgrad(K, M, N):
    from adt_str_amd import kernels as k
    a, b = rnd((K, M), 3).bfloat16(), rnd((K, N), 4).bfloat16()
    ref = a.float().t() @ b.float()
    got = k.gemm(a, b, trans=True, out_dtype=torch.float32)
    close(got, ref, 2e-4, "tn fp32")
    again = k.gemm(a, b, trans=True, out_dtype=torch.float32)
    assert torch.equal(got, again)                      # split-K slabs are summed in a fixed order


def test_tn_exact_integers():
    from adt_str_amd import kernels as k
    K, M, N = 192, 144, 136
    g = torch.Generator().manual_seed(1)
    a = torch.randint(-3, 4, (K, M), generator=g).float().to(DEV).bfloat16()
    b = torch.randint(-2, 3, (K, N), generator=g).float().to(DEV).bfloat16()
    assert torch.equal(k.gemm(a, b, trans=True, out_dtype=torch.float32), a.float().t() @ b.float())


def test_epilogues():
    from adt_str_amd import kernels as k
    M, N, K = 384, 512, 256
    a, b = rnd((M, K), 5).bfloat16(), rnd((N, K), 6, 0.1).bfloat16()
    bias = rnd((N,), 7)
    res = rnd((M, N), 8)
    z = a.float() @ b.float().t()
    close(k.gemm(a, b, bias=bias, out_dtype=torch.float32), z + bias, 1e-4, "bias")
    close(k.gemm(a, b, bias=bias, residual=res, out_dtype=torch.float32), z + bias + res, 1e-4, "bias+residual")
    pe = rnd((96, N), 9)                                             # row-periodic table (positional encoding)
    close(k.gemm(a, b, residual=pe, res_row_mod=96, out_dtype=torch.float32),
          z + pe.repeat(M // 96, 1), 1e-4, "periodic residual")
    close(k.gemm(a, b, alpha=0.25, out_dtype=torch.float32), 0.25 * z, 1e-4, "alpha")
    aux = torch.empty((M, N), dtype=torch.bfloat16, device=DEV)
    full = k.gemm(a, b, bias=bias, residual=res, out_dtype=torch.float32, aux_bf16_out=aux)
    assert torch.equal(aux, full.bfloat16())
    # GELU forward with saved pre-activation
    u = torch.empty((M, N), dtype=torch.bfloat16, device=DEV)
    h = k.gemm(a, b, bias=bias, act=1, pre_act_out=u)
    uref = (z + bias)
    close(u, uref, 6e-3, "pre-activation")
    close(h, torch.nn.functional.gelu(u.float()), 6e-3, "gelu(u)")
    # dgrad through GELU: out = z * gelu'(u)
    uu = u.float().requires_grad_(True)
    torch.nn.functional.gelu(uu).sum().backward()
    close(k.gemm(a, b, gelu_grad_of=u, out_dtype=torch.float32), z * uu.grad, 2e-4, "gelu grad")


def test_views_with_leading_dimension():
    from adt_str_amd import kernels as k
    big_a, big_b = rnd((200, 512), 10).bfloat16(), rnd((300, 512), 11).bfloat16()
    a, b = big_a[:, 128:384], big_b[:, 128:384]
    out = torch.zeros((200, 400), dtype=torch.float32, device=DEV)
    k.gemm(a, b, out=out[:, 50:350])
    assert torch.all(out[:, :50] == 0) and torch.all(out[:, 350:] == 0)
    close(out[:, 50:350], a.float() @ b.float().t(), 1e-4, "strided")


def test_rejects_misaligned():
    from adt_str_amd import _ffi, kernels as k
    a, b = rnd((64, 68), 1).bfloat16(), rnd((64, 68), 2).bfloat16()
    with pytest.raises(_ffi.AdtError):
        k.gemm(a, b)


def test_big_tile_persistent_kernel_epilogues_and_dropout():
    """Shapes that select the persistent 256 x 256 kernel on their own (>= 512 tiles, K >= 256): plain, ragged M / N edges,
    bias + residual, GELU with the saved pre-activation, and the dropout instantiation checked against the oracle's mask."""
    from adt_str_amd import kernels as k
    from oracle import dropout as o_drop
    M, N, K = 8192 - 40, 4096 + 64, 256                      # 32 x 17 tiles, last tile row / column partly outside
    a, b = rnd((M, K), 21).bfloat16(), (rnd((N, K), 22) * 0.1).bfloat16()
    bias, res = rnd((N,), 23), rnd((M, N), 24)
    z = a.float() @ b.float().t()
    close(k.gemm(a, b, out_dtype=torch.float32), z, 1e-4, "plain fp32")
    close(k.gemm(a, b), z, 6e-3, "plain bf16")
    plain = k.gemm(a, b, bias=bias, residual=res, out_dtype=torch.float32)
    close(plain, z + bias + res, 1e-4, "bias + residual")
    u = torch.empty((M, N), dtype=torch.bfloat16, device=DEV)
    h = k.gemm(a, b, bias=bias, act=1, pre_act_out=u)
    close(u, z + bias, 6e-3, "pre-activation")
    close(h, torch.nn.functional.gelu(u.float()), 6e-3, "gelu(u)")
    for rep in range(3):                                     # several launches: the work counters carry over between them
        nb = k.gemm(a, b, bias=bias, out_dtype=torch.float32)
        dropped = k.gemm(a, b, bias=bias, out_dtype=torch.float32, drop=(0.1, 1000 + rep))
        mask = o_drop.scale((M, N), 0.1, 1000 + rep).to(DEV)
        assert torch.equal(dropped, nb * mask)
    y = k.gemm(a, b, bias=bias, residual=res, out_dtype=torch.float32, drop=(0.25, 9), drop_after_residual=False)
    mask = o_drop.scale((M, N), 0.25, 9).to(DEV)
    close(y, (z + bias) * mask + res, 1e-4, "dropout then residual")


def test_fused_column_sums():
    """colsum_out: the column sums of the stored bf16 output (bias gradient of the producing layer), fused in the persistent kernel's
    epilogue (ragged edges, GELU-gradient + dropout form of the FFN dgrad) and computed by the stand-alone kernel for smaller tilings."""
    from adt_str_amd import _ffi, kernels as k
    for (M, N, K, tag) in [(8192 - 40, 4096 + 64, 256, "256^2"), (4096 + 8, 3072, 768, "256^2 FFN"), (500, 264, 96, "128^2"), (77, 24, 64, "v1")]:
        a, b = rnd((M, K), M).bfloat16(), (rnd((N, K), N) * 0.1).bfloat16()
        u = rnd((M, N), 5).bfloat16()
        for kw in ({}, {"gelu_grad_of": u, "drop": (0.1, 77)}):
            cs = torch.full((N,), float("nan"), device=DEV)
            y = k.gemm(a, b, colsum_out=cs, **kw)
            assert torch.equal(y, k.gemm(a, b, **kw)), tag                   # the output itself does not change
            ref = y.double().sum(0)
            scale = y.float().abs().sum(0).max().item()
            assert (cs.double() - ref).abs().max().item() <= 2e-6 * scale + 1e-6, tag
            cs2 = torch.empty_like(cs)
            k.gemm(a, b, colsum_out=cs2, **kw)
            assert torch.equal(cs, cs2), "fixed summation order"
    with pytest.raises(_ffi.AdtError, match="colsum_out"):
        k.gemm(a, b, colsum_out=cs, out_dtype=torch.float32)


def test_randomised_shapes_cover_every_kernel_choice():
    """Random shapes around the dispatch thresholds (persistent 256^2 NT / TN, 128^2 LDS-DMA, register-staged) incl. ragged edges,
    leading dimensions larger than the row, fp32 and bf16 outputs."""
    from adt_str_amd import kernels as k
    rng = np.random.default_rng(2024)
    nt_shapes = [(int(rng.integers(4000, 9000)) // 8 * 8 + 8 * int(rng.integers(0, 2)), int(rng.integers(160, 4600)) // 8 * 8,
                  int(rng.choice([256, 320, 384, 512, 768, 1024]))) for _ in range(6)]
    nt_shapes += [(int(rng.integers(100, 3000)), int(rng.integers(8, 1200)) // 8 * 8, int(rng.integers(1, 40)) * 8) for _ in range(8)]
    nt_shapes += [(66000, 1024, 256), (70000, 160, 512), (8192 * 4, 4096, 64 * 5)]
    for (M, N, K) in nt_shapes:
        big_a = rnd((M, K + 16), M + N).bfloat16()
        a, b = big_a[:, 8:8 + K], (rnd((N, K), K) * 0.2).bfloat16()
        ref = a.float() @ b.float().t()
        out = torch.zeros((M, N + 8), dtype=torch.float32, device=DEV)
        k.gemm(a, b, out=out[:, :N])
        close(out[:, :N], ref, 2e-4, f"NT fp32 {M}x{N}x{K}")
        assert not out[:, N:].any()
        close(k.gemm(a, b), ref, 8e-3, f"NT bf16 {M}x{N}x{K}")
    tn_shapes = [(int(rng.integers(64, 200)) * 64, int(rng.integers(300, 3200)) // 8 * 8, int(rng.integers(300, 1600)) // 8 * 8) for _ in range(5)]
    tn_shapes += [(int(rng.integers(1, 60)) * 8, int(rng.integers(8, 900)) // 8 * 8, int(rng.integers(8, 900)) // 8 * 8) for _ in range(6)]
    tn_shapes += [(64 * 70, 2048, 520), (64 * 64, 512, 1024), (64 * 200, 776, 776)]
    for (K, M, N) in tn_shapes:
        a, b = rnd((K, M), K + M).bfloat16(), (rnd((K, N), N) * 0.2).bfloat16()
        ref = a.float().t() @ b.float()
        got = k.gemm(a, b, trans=True, out_dtype=torch.float32)
        close(got, ref, 3e-4, f"TN {K}x{M}x{N}")
        assert torch.equal(got, k.gemm(a, b, trans=True, out_dtype=torch.float32)), "split-K slabs are summed in a fixed order"


def test_persistent_work_counters_reset_themselves_and_capture_takes_the_tiled_kernels():
    """The persistent 256^2 kernels draw tiles from device counters that the LAST ticket of each launch puts back to zero
    (no host mirror): back-to-back launches of different shapes, NT and TN, on one stream stay exact; a GEMM captured into a HIP
    graph uses the tiled kernels (no shared counters, nothing allocated under capture) and replays with fresh inputs; eager
    persistent launches keep working afterwards."""
    from adt_str_amd import kernels as k
    a, b = rnd((8192, 768), 1).bfloat16(), rnd((3072, 768), 2).bfloat16()           # 32 x 12 tiles: the persistent NT kernel
    a2, b2 = rnd((4096, 768), 3).bfloat16(), rnd((768, 768), 4).bfloat16()          # 16 x 3 tiles: fewer tiles than CUs
    ref, ref2 = a.float() @ b.float().t(), a2.float() @ b2.float().t()
    gw_ref = a.float().t() @ rnd((8192, 1024), 5).bfloat16().float()
    dy = rnd((8192, 1024), 5).bfloat16()
    for _ in range(3):
        close(k.gemm(a, b, out_dtype=torch.float32), ref, 1e-4, "persistent NT")
        close(k.gemm(a2, b2, out_dtype=torch.float32), ref2, 1e-4, "persistent NT, small grid")
        close(k.gemm(a, dy, trans=True, out_dtype=torch.float32), gw_ref, 1e-4, "persistent TN")
    out = torch.empty((8192, 3072), dtype=torch.float32, device=DEV)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        k.gemm(a, b, out=out)                                                      # warm-up on the capture stream
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        k.gemm(a, b, out=out)
    for seed in (7, 8, 9):
        a.copy_(rnd((8192, 768), seed).bfloat16())
        out.zero_()
        graph.replay()
        close(out, a.float() @ b.float().t(), 1e-4, "graph replay")
    close(k.gemm(a, b, out_dtype=torch.float32), a.float() @ b.float().t(), 1e-4, "persistent NT after the replays")


@pytest.mark.parametrize("M,N,Kd", [(8192, 3072, 768), (300, 264, 96)])      # persistent 256^2 kernel / 128^2 kernel
def test_saved_activation_gradient_factor(M, N, Kd):
    """adt_gemm_epilogue.act_grad_mode: the FFN forward saves gelu'(z) * dropout-keep (bf16) instead of z, the backward
    multiplies by it as stored -- against autograd through dropout(gelu(z)) with the kernels' own mask."""
    import torch.nn.functional as F
    from adt_str_amd import kernels as k
    from oracle import dropout as o_drop
    a, w, bias = rnd((M, Kd), 1).bfloat16(), rnd((N, Kd), 2, 0.2).bfloat16(), rnd((N,), 3)
    for site in (None, k.drop_site(0.1, 9, 2)):
        keep = o_drop.scale((M, N), *site).to(DEV) if site else torch.ones((M, N), device=DEV)
        z = (a.float() @ w.float().t() + bias).requires_grad_(True)
        ref_h = F.gelu(z) * keep
        dy = rnd((M, N), 4)
        ref_h.backward(dy)
        fac = torch.empty((M, N), dtype=torch.bfloat16, device=DEV)
        h = k.gemm(a, w, bias=bias, act=1, act_grad_out=fac, drop=site)
        close(h, ref_h.detach(), 6e-3, "h")
        zd = z.detach().double()
        ref_fac = ((0.5 * (1 + torch.erf(zd / math.sqrt(2))) + zd * torch.exp(-0.5 * zd * zd) / math.sqrt(2 * math.pi)) * keep).float()
        assert (z.grad - dy * ref_fac).abs().max() <= 1e-5 * dy.abs().max()       # = gelu'(z) * keep, as autograd has it
        assert (fac.float() - ref_fac).abs().max() <= 8e-3 * ref_fac.abs().max() + 1e-6
        # backward: (dy16 @ W2^T) * factor, with the column sums of the result
        dy16, w2t = rnd((M, 64), 5).bfloat16(), rnd((N, 64), 6, 0.3).bfloat16()
        cs = torch.empty(N, device=DEV)
        du = k.gemm(dy16, w2t, act_grad=fac, colsum_out=cs)
        ref_du = (dy16.float() @ w2t.float().t()) * fac.float()
        close(du, ref_du, 8e-3, "du")
        assert (cs - du.float().sum(0)).abs().max() <= 2e-3 * du.float().abs().sum(0).max()
    # fp32-operand path: the same contract in fp32
    a32, w32 = a.float(), w.float()
    fac32 = torch.empty((M, N), device=DEV)
    h32 = k.gemm(a32, w32, bias=bias, act=1, act_grad_out=fac32)
    z = (a32 @ w32.t() + bias).double().requires_grad_(True)
    F.gelu(z).sum().backward()
    ref_h32 = F.gelu(z.detach())
    assert (h32.double() - ref_h32).abs().max() <= 5e-6 * ref_h32.abs().max() and (fac32.double() - z.grad).abs().max() < 2e-5
    dy32 = rnd((M, N), 7)
    got = k.gemm(dy32, torch.eye(N, device=DEV), act_grad=fac32)
    assert (got - dy32 * fac32).abs().max() < 1e-5


def test_grouped_weight_gradients_match_the_single_launches():
    """adt_gemm_bf16_tn_grouped: many small trans = 1 products in one launch (the decoder's weight gradients) -- against fp32
    torch, and against the single-launch path (which sums split-K slabs: same values up to fp32 summation order)."""
    from adt_str_amd import kernels as K
    dev = DEV
    g = torch.Generator(device=dev).manual_seed(11)
    Kd = 8192
    shapes = [(768, 3072), (3072, 768), (768, 768), (2304, 768), (1400, 768), (8, 136), (264, 40)]
    items, refs = [], []
    flat = torch.zeros(sum(m * n for m, n in shapes), device=dev)
    off = 0
    for m, n in shapes:
        a = (torch.randn(Kd, m, device=dev, generator=g) * 0.5).bfloat16()
        b = (torch.randn(Kd, n, device=dev, generator=g) * 0.5).bfloat16()
        out = flat[off:off + m * n].view(m, n); off += m * n
        items.append((a, b, out))
        refs.append(a.float().t() @ b.float())
    K.gemm_tn_grouped(items)
    for (a, b, out), ref in zip(items, refs):
        scale = ref.abs().max().item()
        assert (out - ref).abs().max().item() <= 2e-5 * scale + 1e-3, (a.shape, b.shape)
        single = K.gemm(a, b, trans=True, out_dtype=torch.float32)
        assert (out - single).abs().max().item() <= 2e-5 * scale + 1e-3
    # strided operands and a strided destination (a slice of a packed in-projection gradient)
    big_a = (torch.randn(Kd, 1536, device=dev, generator=g) * 0.5).bfloat16()
    big_b = (torch.randn(Kd, 1024, device=dev, generator=g) * 0.5).bfloat16()
    dst = torch.zeros(2304, 768, device=dev)
    K.gemm_tn_grouped([(big_a[:, 768:], big_b[:, :768], dst[768:1536])])
    ref = big_a[:, 768:].float().t() @ big_b[:, :768].float()
    assert (dst[768:1536] - ref).abs().max().item() <= 2e-5 * ref.abs().max().item() + 1e-3
    assert torch.all(dst[:768] == 0) and torch.all(dst[1536:] == 0)
    with pytest.raises(Exception):
        K.gemm_tn_grouped([(big_a[:100], big_b[:100], torch.zeros(1536, 1024, device=dev))])      # K not a multiple of 64


def test_persistent_gemm_runs_beside_a_kernel_that_holds_cus():
    """Data parallelism puts RCCL kernels on the chip while the backward's GEMMs run.  adt_debug_occupy stands in for one: 16
    workgroups with 64 KB of LDS each for 40 ms on a side stream.  The persistent GEMM (one workgroup per CU, the whole LDS)
    must give the same bits and must not wait for the occupier."""
    import time
    from adt_str_amd import kernels as K, _ffi
    g = torch.Generator(device=DEV).manual_seed(3)
    a = (torch.randn(63104 // 2, 768, device=DEV, generator=g) * 0.5).bfloat16()
    w = (torch.randn(3072, 768, device=DEV, generator=g) * 0.03).bfloat16()
    ref = K.gemm(a, w)
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    _ffi.call("adt_debug_occupy", 16, 65536, 40000, side.cuda_stream)
    time.sleep(0.003)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    outs = [K.gemm(a, w) for _ in range(10)]
    e1.record()
    e1.synchronize()
    busy = not side.query()                     # the occupier is still on the chip when the ten GEMMs are done
    torch.cuda.synchronize()
    assert all(torch.equal(o, ref) for o in outs)
    assert busy and e0.elapsed_time(e1) < 20.0, (busy, e0.elapsed_time(e1))


@pytest.mark.parametrize("M", [1, 3, 8, 16, 17, 32, 47, 64])
@pytest.mark.parametrize("N,K", [(768, 768), (2304, 768), (768, 3072), (1400, 768), (3072, 768), (24, 128)])
def test_skinny_rows_kernel(M, N, K, monkeypatch):
    """M <= 64 (the decode step's projections; 1, 2 or 4 row tiles of 16): a 16-column workgroup per output slice, K split over its four waves, the generic
    epilogue.  Against the fp32 product, and against the tile kernel the same call takes with ADT_GEMM_NO_SKINNY... (set before
    the library reads it: the comparison with the tiled kernels is through the fp32 reference both must match)."""
    from adt_str_amd import kernels as k
    a_full = rnd((M, K + 64), 3 + M).bfloat16()
    a = a_full[:, :K]                                                    # row stride != K
    w, bias, res = rnd((N, K), 4, 0.05).bfloat16(), rnd((N,), 5), rnd((M, N), 6)
    z = a.float() @ w.float().t()
    close(k.gemm(a, w, out_dtype=torch.float32), z, 1e-4, "plain fp32")
    close(k.gemm(a, w), z, 6e-3, "plain bf16")
    close(k.gemm(a, w, bias=bias, residual=res, out_dtype=torch.float32), z + bias + res, 1e-4, "bias + residual")
    close(k.gemm(a, w, bias=bias, act=1), torch.nn.functional.gelu(z + bias), 8e-3, "bias + gelu")
    site = k.drop_site(0.25, 3, 1)
    from oracle import dropout as o_drop
    sc = o_drop.scale((M, N), *site).to(DEV)
    close(k.gemm(a, w, bias=bias, out_dtype=torch.float32, drop=site), (z + bias) * sc, 1e-4, "dropout: the same mask as the tiled kernels")
    aux = torch.empty((M, N), dtype=torch.bfloat16, device=DEV)
    out = k.gemm(a, w, out_dtype=torch.float32, aux_bf16_out=aux)
    assert torch.equal(aux, out.bfloat16())


@pytest.mark.parametrize("M", [1, 8, 17, 64])
@pytest.mark.parametrize("N,K", [(768, 768), (2304, 768), (1400, 768), (3072, 768), (768, 1024), (64, 128)])
def test_layernorm_fused_into_the_skinny_gemm(M, N, K):
    """adt_ln_gemm_bf16 (the decode step's LayerNorm -> projection pairs in one launch) against adt_layernorm_fwd + adt_gemm_bf16:
    the LayerNorm output it also emits, and the product with bias / GELU / residual / fp32 output."""
    from adt_str_amd import kernels as k
    y = rnd((M, K + 4), 11 + M, 2.0)[:, :K] + 0.5                        # row stride != K, non-zero mean
    gamma, beta = rnd((K,), 12) * 0.2 + 1.0, rnd((K,), 13) * 0.1
    w, bias, res = rnd((N, K), 14, 0.05).bfloat16(), rnd((N,), 15), rnd((M, N), 16)
    x32_ref, x16_ref, _, _ = k.layernorm_fwd(y.contiguous(), gamma, beta)
    out, x32 = k.ln_gemm(y, gamma, beta, w, bias=bias)
    assert (x32 - x32_ref).abs().max() < 2e-5
    close(out, k.gemm(x16_ref, w, bias=bias).float(), 8e-3, "bias")
    out, _ = k.ln_gemm(y, gamma, beta, w, bias=bias, residual=res, out_dtype=torch.float32, want_x32=False)
    close(out, k.gemm(x16_ref, w, bias=bias, residual=res, out_dtype=torch.float32), 2e-3, "bias + residual, fp32 out")
    out, _ = k.ln_gemm(y, gamma, beta, w, bias=bias, act=1)
    close(out, k.gemm(x16_ref, w, bias=bias, act=1).float(), 1e-2, "bias + gelu")
    with pytest.raises(RuntimeError):
        k.ln_gemm(rnd((65, K), 1), gamma, beta, w)


@pytest.mark.parametrize("group", ["1", "2", "5", "7", "99"])
def test_column_group_tile_orders_cover_the_output(group):
    """The persistent NT kernel walks the tile grid in column groups (default 3 tile columns; ADT_GEMM_GROUP_N is read once per
    process, hence the subprocess): every group width -- including ones that leave a narrower last group (9 and 12 tile columns
    with 5 or 7 per group) and the row-major order (99) -- must produce the same bits as the default order."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import torch, hashlib\n"
        "from adt_str_amd import kernels as k\n"
        "g = torch.Generator().manual_seed(3)\n"
        "out = []\n"
        "for M, N, K in ((63104 // 4, 2304, 768), (8192, 3072, 768), (5000, 1400, 256)):\n"
        "    a = torch.randn((M, K), generator=g).cuda().bfloat16(); w = (torch.randn((N, K), generator=g) * 0.05).cuda().bfloat16()\n"
        "    bias = torch.randn(N, generator=g).cuda()\n"
        "    y = k.gemm(a, w, bias=bias, act=1)\n"
        "    out.append(hashlib.sha1(y.view(torch.int16).cpu().numpy().tobytes()).hexdigest())\n"
        "print('HASH', *out)\n")
    def run(env_group):
        env = dict(os.environ, PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""))
        env.pop("ADT_GEMM_GROUP_N", None)
        if env_group is not None:
            env["ADT_GEMM_GROUP_N"] = env_group
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        return [l for l in r.stdout.splitlines() if l.startswith("HASH")][0]
    assert run(group) == run(None)


@pytest.mark.parametrize("M,N,K", [(63104 // 8, 768, 768), (8192, 768, 3072), (300, 96, 64), (8, 768, 768), (100, 40, 72)])
def test_residual_rebuilt_from_the_pre_layernorm_tensor(M, N, K):
    """adt_gemm_epilogue.res_ln_*: the residual the epilogue adds is LayerNorm(y), rebuilt from y and the row statistics that
    adt_layernorm_fwd saved -- on every kernel the shapes select (persistent 256^2, 128^2, skinny, scalar) and with dropout."""
    from adt_str_amd import kernels as k
    from oracle import dropout as o_drop
    a, w, bias = rnd((M, K), 1).bfloat16(), rnd((N, K), 2, 0.05).bfloat16(), rnd((N,), 3)
    y = rnd((M, N), 4, 2.0) + 0.3
    gamma, beta = rnd((N,), 5) * 0.2 + 1.0, rnd((N,), 6) * 0.1
    if N % 4 == 0 and N <= 1024:
        x32, _, mean, rstd = k.layernorm_fwd(y, gamma, beta)
    else:
        pytest.skip("LayerNorm kernel shape")
    site = k.drop_site(0.1, 5, 2)
    plain = k.gemm(a, w, bias=bias, residual=x32, out_dtype=torch.float32, drop=site)
    lean = k.gemm(a, w, bias=bias, residual=y, residual_ln=(mean, rstd, gamma, beta), out_dtype=torch.float32, drop=site)
    assert (plain - lean).abs().max() <= 2e-6 * max(1.0, plain.abs().max().item())
    z = (a.float() @ w.float().t() + bias) * o_drop.scale((M, N), *site).to(DEV) + x32
    close(lean, z, 1e-4, "against the fp32 product")
    with pytest.raises(AssertionError):
        k.gemm(a, w, residual=y, residual_ln=(mean, rstd, gamma, beta), res_row_mod=4, out_dtype=torch.float32)


def test_weight_gradient_with_a_ragged_row_count_stays_on_the_fast_path():
    """K = batch x frames is not a multiple of the 64-deep K-tile for most batch sizes (48 x 986 = 47 328): the product is taken as
    floor(K / 64) * 64 rows on the LDS-DMA kernels plus a < 64-row remainder added by the fallback kernel, so it costs about what
    the neighbouring multiple of 64 costs instead of several times more."""
    from adt_str_amd import kernels as k
    M, N = 768, 3072
    def timed(K):
        a, b = rnd((K, M), 5).bfloat16(), rnd((K, N), 6).bfloat16()
        out = torch.empty((M, N), device=DEV)
        for _ in range(3):
            k.gemm(a, b, trans=True, out=out)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            k.gemm(a, b, trans=True, out=out)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / 10
    even, ragged = timed(47296), timed(47328)
    assert ragged < 1.5 * even, (even, ragged)
