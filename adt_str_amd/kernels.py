"""Thin Python wrappers over the C ABI (one function per entry point).

Tensors are PyTorch GPU tensors used as plain device buffers; every call runs
on PyTorch's current HIP stream.  No autograd here (see ``adt_str_amd.network``).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

from . import _ffi

_ws_cache: dict = {}


def mix32(x):
    """The integer hash of adt_str_amd/csrc/dropout.h, on Python ints (host-side key derivation)."""
    x &= 0xFFFFFFFF
    x ^= x >> 16; x = (x * 0x7FEB352D) & 0xFFFFFFFF; x ^= x >> 15; x = (x * 0x846CA68B) & 0xFFFFFFFF; x ^= x >> 16
    return x


def drop_site(p: float, seed: int, site: int):
    """(p, key) of one dropout site for one training step; None when dropout is off."""
    return (float(p), mix32(mix32(seed) ^ (site * 0x9E3779B9 & 0xFFFFFFFF))) if p and p > 0 else None


def _drop_ptr(d):
    if d is None:
        return None, None
    obj = _ffi.Dropout(d[0], d[1])
    return obj, C.byref(obj)


def _workspace(nbytes: int, device) -> torch.Tensor:
    """Grow-only scratch buffer per (device, stream) (never freed: stream-ordered reuse; kernels on different streams
    may be in flight at the same time, so they must not share one)."""
    key = (str(device), _ffi.current_stream() if torch.device(device).type == "cuda" else 0)
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)
        _ws_cache[key] = buf
    return buf


# Measurement tap (bench.py): {"match": callable(trans, M, N, K, epilogue) -> bool, "events": []}.  While set, every matching
# adt_gemm_bf16 launch is bracketed by a pair of HIP events on the launch stream, so a kernel can be timed where the workload
# launches it -- among the step's other kernels -- instead of in a loop of its own.
gemm_tap = None


def gemm(a: torch.Tensor, b: torch.Tensor, *, trans: bool = False, out: Optional[torch.Tensor] = None,
         out_dtype=torch.bfloat16, bias: Optional[torch.Tensor] = None, residual: Optional[torch.Tensor] = None,
         res_row_mod: int = 0, act: int = 0, pre_act_out: Optional[torch.Tensor] = None,
         gelu_grad_of: Optional[torch.Tensor] = None, alpha: float = 1.0,
         aux_bf16_out: Optional[torch.Tensor] = None, drop=None, drop_after_residual: bool = False,
         colsum_out: Optional[torch.Tensor] = None, b_kn: bool = False, act_grad_out: Optional[torch.Tensor] = None,
         act_grad: Optional[torch.Tensor] = None, residual_ln=None) -> torch.Tensor:
    """``trans=False``: ``a[M,K] @ b[N,K].T``; ``trans=True``: ``a[K,M].T @ b[K,N]`` (bf16 in, fp32 accumulate).
    ``colsum_out`` (fp32 [N]) also receives the column sums of the bf16 output.
    ``act_grad_out`` (forward, with ``act=1``): receives ``gelu'(z) * dropout-keep`` instead of the pre-activation;
    ``act_grad`` (backward): that saved factor, multiplied in as stored (``adt_gemm_epilogue.act_grad_mode``).
    ``residual_ln = (mean, rstd, gamma, beta)``: ``residual`` is the PRE-LayerNorm tensor and the epilogue adds LayerNorm(residual)
    rebuilt from those statistics (the LayerNorm then never writes its fp32 output; bf16 path only).
    fp32 operands take the fp32-operand parity path (``adt_gemm_f32``: same epilogue, fp32 everywhere); there
    ``b_kn=True`` reads ``b`` as ``[K, N]`` (``a[M,K] @ b[K,N]``, the data gradient against the master weight)."""
    mode = 0
    if act_grad_out is not None or act_grad is not None:
        assert pre_act_out is None and gelu_grad_of is None and (act_grad_out is None or act == 1) and (act_grad is None or drop is None)
        pre_act_out, gelu_grad_of, mode = act_grad_out, act_grad, 1
    if a.dtype == torch.float32:
        assert residual_ln is None, "residual_ln is a bf16-path epilogue"
        return _gemm_f32(a, b, trans=trans, b_kn=b_kn, out=out, bias=bias, residual=residual, res_row_mod=res_row_mod, act=act,
                         pre_act_out=pre_act_out, gelu_grad_of=gelu_grad_of, alpha=alpha, drop=drop,
                         drop_after_residual=drop_after_residual, colsum_out=colsum_out, aux_out=aux_bf16_out, mode=mode)
    assert a.dtype == torch.bfloat16 and b.dtype == torch.bfloat16 and a.dim() == 2 and b.dim() == 2 and not b_kn
    assert a.stride(1) == 1 and b.stride(1) == 1
    if trans:
        K, M = a.shape
        K2, N = b.shape
    else:
        M, K = a.shape
        N, K2 = b.shape
    assert K == K2, (a.shape, b.shape)
    if out is None:
        out = torch.empty((M, N), dtype=out_dtype, device=a.device)
    assert out.shape == (M, N) and out.stride(1) == 1 and out.dtype in (torch.bfloat16, torch.float32)
    ep = _ffi.GemmEpilogue()
    ep.alpha = alpha
    ep.act = act
    ep.act_grad_mode = mode
    ep.out_fp32 = 1 if out.dtype == torch.float32 else 0
    if drop is not None:
        ep.drop.p, ep.drop.key = drop
        ep.drop_after_residual = 1 if drop_after_residual else 0
    if bias is not None:
        assert bias.dtype == torch.float32 and bias.numel() == N
        ep.bias = _ffi.dptr(bias)
    if residual is not None:
        assert residual.dtype == torch.float32 and residual.stride(-1) == 1
        ep.residual, ep.ld_res, ep.res_row_mod = _ffi.dptr(residual), residual.stride(-2), res_row_mod
    if residual_ln is not None:
        mean, rstd, ln_g, ln_b = residual_ln
        assert residual is not None and res_row_mod == 0 and residual.shape == (M, N)
        for t_, n_ in ((mean, M), (rstd, M), (ln_g, N), (ln_b, N)):
            assert t_.dtype == torch.float32 and t_.is_contiguous() and t_.numel() == n_
        ep.res_ln_mean, ep.res_ln_rstd, ep.res_ln_gamma, ep.res_ln_beta = _ffi.dptr(mean), _ffi.dptr(rstd), _ffi.dptr(ln_g), _ffi.dptr(ln_b)
    if pre_act_out is not None:
        assert pre_act_out.dtype == torch.bfloat16 and pre_act_out.shape == (M, N)
        ep.pre_act_out, ep.ld_pre_act = _ffi.dptr(pre_act_out), pre_act_out.stride(0)
    if gelu_grad_of is not None:
        assert gelu_grad_of.dtype == torch.bfloat16 and gelu_grad_of.shape == (M, N)
        ep.gelu_grad_of, ep.ld_gelu_grad = _ffi.dptr(gelu_grad_of), gelu_grad_of.stride(0)
    if aux_bf16_out is not None:
        assert aux_bf16_out.dtype == torch.bfloat16 and aux_bf16_out.shape == (M, N)
        ep.aux_bf16_out, ep.ld_aux = _ffi.dptr(aux_bf16_out), aux_bf16_out.stride(0)
    ws_bytes = _ffi.load().adt_gemm_workspace_bytes(int(trans), M, N, K) if trans else 0
    if colsum_out is not None:
        assert not trans and colsum_out.dtype == torch.float32 and colsum_out.numel() == N
        assert colsum_out.is_contiguous()
        ep.colsum_out = _ffi.dptr(colsum_out)
        ws_bytes = _ffi.load().adt_gemm_colsum_workspace_bytes(M, N)
    ws = _workspace(ws_bytes, a.device) if ws_bytes else None
    tap = gemm_tap
    timed = tap is not None and tap["match"](trans, M, N, K, ep)
    if timed:
        e0 = torch.cuda.Event(enable_timing=True)
        e0.record()
    _ffi.call("adt_gemm_bf16", int(trans), M, N, K, _ffi.dptr(a), a.stride(0), _ffi.dptr(b), b.stride(0),
              _ffi.dptr(out), out.stride(0), C.byref(ep), _ffi.dptr(ws) if ws is not None else None, ws_bytes,
              _ffi.current_stream())
    if timed:
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
        tap["events"].append((e0, e1))
    return out


def gemm_tn_grouped(items) -> None:
    """Weight gradients ``out_i = a_i.T @ b_i`` (``a_i`` [K, M] and ``b_i`` [K, N] bf16, ``out_i`` [M, N] fp32) for a list of
    ``(a, b, out)`` in one launch per 32 items (``adt_gemm_bf16_tn_grouped``: whole-K tiles, no split-K slabs)."""
    items = list(items)
    for lo in range(0, len(items), 32):
        chunk = items[lo:lo + 32]
        arr = (_ffi.GemmTnItem * len(chunk))()
        for it, (a, b, out) in zip(arr, chunk):
            assert a.dtype == torch.bfloat16 and b.dtype == torch.bfloat16 and out.dtype == torch.float32
            assert a.dim() == 2 and b.dim() == 2 and out.dim() == 2 and a.stride(1) == 1 and b.stride(1) == 1 and out.stride(1) == 1
            assert a.shape[0] == b.shape[0] and out.shape == (a.shape[1], b.shape[1])
            it.A, it.lda, it.B, it.ldb, it.C, it.ldc = _ffi.dptr(a), a.stride(0), _ffi.dptr(b), b.stride(0), _ffi.dptr(out), out.stride(0)
            it.M, it.N, it.K = a.shape[1], b.shape[1], a.shape[0]
        _ffi.call("adt_gemm_bf16_tn_grouped", C.byref(arr), len(chunk), _ffi.current_stream())


# How the fp32-operand entry points (adt_gemm_f32, adt_attn_fwd_f32, adt_attn_bwd_f32) form their products: "f32" = exact f32-input
# MFMA (the parity arm of rounds 4-5), "bf16x3" = split-bf16: every fp32 operand as hi + lo bf16 planes, three bf16 MFMAs per product
# into one fp32 accumulator (~1e-5 relative per product, several times faster).  The engine sets it on entry to each of its passes
# (network._Engine: precision "fp32" / "bf16x3"); module-level because the calls below are synchronous host code.
f32_products = "f32"


def set_f32_products(mode: str) -> None:
    global f32_products
    if mode not in ("f32", "bf16x3"):
        raise ValueError(f"f32_products must be 'f32' or 'bf16x3', not {mode!r}")
    f32_products = mode


# ---- split-bf16 products on the persistent bf16 kernels (adt_gemm_bf16x3): operands as [hi | lo] bf16 plane pairs ------------------
def split_planes(x: torch.Tensor, transpose: bool = False) -> torch.Tensor:
    """fp32 ``x [rows, cols]`` -> bf16 ``[rows, 2 cols]`` = ``[bf16(x) | bf16(x - bf16(x))]`` (``transpose``: the planes of ``x.T``,
    ``[cols, 2 rows]``) -- ``adt_split_bf16x2``."""
    assert x.dtype == torch.float32 and x.dim() == 2 and x.stride(1) == 1
    rows, cols = x.shape
    w = rows if transpose else cols
    out = torch.empty((cols if transpose else rows, 2 * w), dtype=torch.bfloat16, device=x.device)
    _ffi.call("adt_split_bf16x2", _ffi.dptr(x), x.stride(0), rows, cols, _ffi.dptr(out), out.stride(0), w, int(transpose), _ffi.current_stream())
    return out


# Weights are split once per refresh by the engine (both orientations) and found again by address: ptr -> (tensor, version, rows, cols,
# planes [rows, 2 cols], planes of W^T [cols, 2 rows]).  A GEMM operand that is a row / column slice of a registered weight takes a view.
_x3_weights: dict = {}
_x3_owner = None          # id of the engine whose weights are registered (network._Engine.refresh_weights)


def x3_register_weights(tensors) -> None:
    """(The registry keeps the tensors themselves: while an address is registered its storage cannot be handed to another tensor.)"""
    _x3_weights.clear()
    for w in tensors:
        assert w.dtype == torch.float32 and w.dim() == 2 and w.is_contiguous()
        _x3_weights[w.data_ptr()] = (w, w._version, w.shape[0], w.shape[1], split_planes(w), split_planes(w, transpose=True))


def _x3_weight_planes(b: torch.Tensor, kn: bool):
    """Planes for GEMM operand ``b`` if it lies inside a registered weight: (view, lo_off) of W's planes for ``b [N, K]`` (``kn=False``)
    or of W^T's planes for ``b [K, N]`` read as ``[K, N]`` (``kn=True``); None otherwise."""
    if not _x3_weights or b.dim() != 2 or b.stride(1) != 1:
        return None
    ptr = b.data_ptr()
    for base, (w, ver, rows, cols, pw, pt) in _x3_weights.items():
        if base <= ptr < base + rows * cols * 4:
            if w._version != ver or b.stride(0) != cols:
                return None
            off = (ptr - base) // 4
            r0, c0 = off // cols, off % cols
            if r0 + b.shape[0] > rows or c0 + b.shape[1] > cols or (c0 & 7) or (r0 & 7):
                return None
            return (pt[c0:c0 + b.shape[1], r0:], rows) if kn else (pw[r0:r0 + b.shape[0], c0:], cols)
    return None


def _x3_activation_planes(x: torch.Tensor):
    """``[hi | lo]`` planes of an fp32 activation, split once per tensor object and content version (an activation is typically the
    operand of a forward or data-gradient product AND of a weight-gradient product)."""
    c = getattr(x, "_adt_x3", None)
    if c is not None and c[0] == x._version and c[1] == x.data_ptr():
        return c[2], x.shape[1]
    pl = split_planes(x)
    try:
        x._adt_x3 = (x._version, x.data_ptr(), pl)
    except Exception:          # (a tensor subclass without a __dict__)
        pass
    return pl, x.shape[1]


def _gemm_x3_fast(a, b, layout, M, N, K, out, ep) -> bool:
    """Try the persistent-kernel form of the split-bf16 product; False when the shape does not take it (the caller then runs adt_gemm_f32)."""
    lib = _ffi.load()
    trans = layout == 3
    if os.environ.get("ADT_X3_TILED") or not lib.adt_gemm_bf16x3_supported(int(trans), M, N, K):
        return False
    if (a.data_ptr() & 15) or (b.data_ptr() & 15) or (a.stride(0) & 3) or (b.stride(0) & 3) or (out.data_ptr() & 15) or (out.stride(0) & 3):
        return False
    a2, a_lo = _x3_activation_planes(a)
    if layout == 2:
        wp = _x3_weight_planes(b, True)
        b2, b_lo = wp if wp is not None else (split_planes(b, transpose=True), b.shape[0])
    else:
        wp = _x3_weight_planes(b, False) if layout == 0 else None
        b2, b_lo = wp if wp is not None else _x3_activation_planes(b)
    ep.side_fp32 = 1
    ws_bytes = lib.adt_gemm_bf16x3_workspace_bytes(int(trans), M, N, K) if trans else 0
    ws = _workspace(ws_bytes, a.device) if ws_bytes else None
    _ffi.call("adt_gemm_bf16x3", int(trans), M, N, K, _ffi.dptr(a2), a2.stride(0), a_lo, _ffi.dptr(b2), b2.stride(0), b_lo, _ffi.dptr(out), out.stride(0),
              C.byref(ep), _ffi.dptr(ws) if ws is not None else None, ws_bytes, _ffi.current_stream())
    return True


def _gemm_f32(a, b, *, trans, b_kn, out, bias, residual, res_row_mod, act, pre_act_out, gelu_grad_of, alpha, drop,
              drop_after_residual, colsum_out, aux_out, mode=0):
    """fp32-operand GEMM (parity path).  Layout bits: 1 = a is [K, M], 2 = b is [K, N], 4 = split-bf16 products (``f32_products``)."""
    assert b.dtype == torch.float32 and a.dim() == 2 and b.dim() == 2 and a.stride(1) == 1 and b.stride(1) == 1
    assert aux_out is None, "the fp32 path has a single (fp32) output"
    if trans:
        (K, M), (K2, N), layout = a.shape, b.shape, 3
    elif b_kn:
        (M, K), (K2, N), layout = a.shape, b.shape, 2
    else:
        (M, K), (N, K2), layout = a.shape, b.shape, 0
    assert K == K2, (a.shape, b.shape)
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=a.device)
    assert out.shape == (M, N) and out.stride(1) == 1 and out.dtype == torch.float32
    ep = _ffi.GemmEpilogue()
    ep.alpha, ep.act, ep.out_fp32, ep.act_grad_mode = alpha, act, 1, mode
    if drop is not None:
        ep.drop.p, ep.drop.key = drop
        ep.drop_after_residual = 1 if drop_after_residual else 0
    if bias is not None:
        assert bias.dtype == torch.float32 and bias.numel() == N
        ep.bias = _ffi.dptr(bias)
    if residual is not None:
        assert residual.dtype == torch.float32 and residual.stride(-1) == 1
        ep.residual, ep.ld_res, ep.res_row_mod = _ffi.dptr(residual), residual.stride(-2), res_row_mod
    if pre_act_out is not None:
        assert pre_act_out.dtype == torch.float32 and pre_act_out.shape == (M, N)
        ep.pre_act_out, ep.ld_pre_act = _ffi.dptr(pre_act_out), pre_act_out.stride(0)
    if gelu_grad_of is not None:
        assert gelu_grad_of.dtype == torch.float32 and gelu_grad_of.shape == (M, N)
        ep.gelu_grad_of, ep.ld_gelu_grad = _ffi.dptr(gelu_grad_of), gelu_grad_of.stride(0)
    if f32_products == "bf16x3":
        if _gemm_x3_fast(a, b, layout, M, N, K, out, ep):          # large shapes: the persistent bf16 kernels over [hi | lo] planes
            if colsum_out is not None:
                colsum(out, out=colsum_out)
            return out
        layout |= 4
    ws_bytes = _ffi.load().adt_gemm_f32_workspace_bytes(layout, M, N, K) if trans else 0      # (K splits pay for the weight gradients only)
    ws = _workspace(ws_bytes, a.device) if ws_bytes else None
    _ffi.call("adt_gemm_f32", layout, M, N, K, _ffi.dptr(a), a.stride(0), _ffi.dptr(b), b.stride(0), _ffi.dptr(out), out.stride(0),
              C.byref(ep), _ffi.dptr(ws) if ws is not None else None, ws_bytes, _ffi.current_stream())
    if colsum_out is not None:
        colsum(out, out=colsum_out)
    return out


def _p(t: Optional[torch.Tensor]):
    return _ffi.dptr(t) if t is not None else None


def layernorm_fwd(x, gamma, beta, eps=1e-5, want32=True, want16=True, drop=None):
    """x fp32 [M, D] -> (y32 | None, y16 | None, mean[M], rstd[M])."""
    assert x.dtype == torch.float32 and x.dim() == 2 and x.stride(1) == 1
    M, D = x.shape
    y32 = torch.empty((M, D), dtype=torch.float32, device=x.device) if want32 else None
    y16 = torch.empty((M, D), dtype=torch.bfloat16, device=x.device) if want16 else None
    mean = torch.empty(M, dtype=torch.float32, device=x.device)
    rstd = torch.empty(M, dtype=torch.float32, device=x.device)
    _keep, dp = _drop_ptr(drop)
    _ffi.call("adt_layernorm_fwd", _ffi.dptr(x), x.stride(0), _ffi.dptr(gamma), _ffi.dptr(beta), eps, _p(y32), _p(y16), D,
              _ffi.dptr(mean), _ffi.dptr(rstd), M, D, dp, _ffi.current_stream())
    return y32, y16, mean, rstd


def layernorm_bwd(dy, x, gamma, mean, rstd, dgamma=None, dbeta=None, dxsum=None, want32=True, want16=True, dy_drop=None,
                  dx16_drop=None, branch_dtype=torch.bfloat16):
    """-> (dx32 | None, dx16 | None); dgamma/dbeta/dxsum [D] fp32 are overwritten when given.  ``branch_dtype=float32``
    (fp32-operand path): the branch gradient ``dx16`` is fp32; without a branch dropout it IS ``dx32``."""
    assert dy.dtype == torch.float32 and x.dtype == torch.float32 and dy.shape == x.shape
    M, D = x.shape
    if branch_dtype == torch.float32:
        alias = want16 and want32 and dx16_drop is None
        dx32 = torch.empty((M, D), dtype=torch.float32, device=x.device) if want32 else None
        dxb = None if (alias or not want16) else torch.empty((M, D), dtype=torch.float32, device=x.device)
        nb = _ffi.load().adt_layernorm_bwd_workspace_bytes(M, D)
        ws = _workspace(nb, x.device)
        _k1, p1 = _drop_ptr(dy_drop)
        _k2, p2 = _drop_ptr(dx16_drop)
        _ffi.call("adt_layernorm_bwd_f32", _ffi.dptr(dy), dy.stride(0), _ffi.dptr(x), x.stride(0), _ffi.dptr(gamma), _ffi.dptr(mean),
                  _ffi.dptr(rstd), _p(dx32), _p(dxb), D, _p(dgamma), _p(dbeta), _p(dxsum), M, D, p1, p2, _ffi.dptr(ws), nb,
                  _ffi.current_stream())
        return dx32, (dx32 if alias else dxb)
    dx32 = torch.empty((M, D), dtype=torch.float32, device=x.device) if want32 else None
    dx16 = torch.empty((M, D), dtype=torch.bfloat16, device=x.device) if want16 else None
    nb = _ffi.load().adt_layernorm_bwd_workspace_bytes(M, D)
    ws = _workspace(nb, x.device)
    _k1, p1 = _drop_ptr(dy_drop)
    _k2, p2 = _drop_ptr(dx16_drop)
    _ffi.call("adt_layernorm_bwd", _ffi.dptr(dy), dy.stride(0), _ffi.dptr(x), x.stride(0), _ffi.dptr(gamma), _ffi.dptr(mean),
              _ffi.dptr(rstd), _p(dx32), _p(dx16), D, _p(dgamma), _p(dbeta), _p(dxsum), M, D, p1, p2, _ffi.dptr(ws), nb,
              _ffi.current_stream())
    return dx32, dx16


def colsum(x, out=None):
    assert x.dtype in (torch.bfloat16, torch.float32) and x.dim() == 2 and x.stride(1) == 1
    M, N = x.shape
    if out is None:
        out = torch.empty(N, dtype=torch.float32, device=x.device)
    if x.dtype == torch.float32:
        nb = _ffi.load().adt_colsum_f32_workspace_bytes(M, N)
        ws = _workspace(nb, x.device)
        _ffi.call("adt_colsum_f32", _ffi.dptr(x), x.stride(0), M, N, _ffi.dptr(out), _ffi.dptr(ws), nb, _ffi.current_stream())
        return out
    nb = _ffi.load().adt_colsum_workspace_bytes(M, N)
    ws = _workspace(nb, x.device)
    _ffi.call("adt_colsum_bf16", _ffi.dptr(x), x.stride(0), M, N, _ffi.dptr(out), _ffi.dptr(ws), nb, _ffi.current_stream())
    return out


def greedy_step(logits, finished, gen, t, tok, klen, done_at, end_token: int) -> None:
    """The tail of a greedy-decode step on device state (``adt_greedy_step``; reference model.py:300-322): ``gen[:, t + 1]`` <- the
    arg-max token (the end token for rows already finished), ``finished`` / ``tok`` / ``klen`` / ``done_at`` / ``t`` updated in place."""
    assert logits.dtype == torch.float32 and logits.dim() == 2 and logits.stride(1) == 1
    assert finished.dtype == torch.bool and gen.dtype == torch.int64 and gen.stride(1) == 1 and t.dtype == torch.int64
    assert tok.dtype == torch.int64 and tok.is_contiguous() and klen.dtype == torch.int32 and done_at.dtype == torch.int64
    B, V = logits.shape
    _ffi.call("adt_greedy_step", _ffi.dptr(logits), logits.stride(0), B, V, _ffi.dptr(finished), _ffi.dptr(gen), gen.stride(0), _ffi.dptr(t),
              _ffi.dptr(tok), _ffi.dptr(klen), _ffi.dptr(done_at), int(end_token), gen.shape[1], _ffi.current_stream())


def ln_gemm(y32: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, w: torch.Tensor, *, eps: float = 1e-5, want_x32: bool = True,
            out_dtype=torch.bfloat16, bias: Optional[torch.Tensor] = None, residual: Optional[torch.Tensor] = None, act: int = 0):
    """``LayerNorm(y32) @ w.T`` (+ bias / activation / residual) in one launch for at most 64 rows (``adt_ln_gemm_bf16``: the decode
    step's LayerNorm -> projection pairs) -> ``(out [M, N], x32 = LayerNorm(y32) fp32 or None)``."""
    assert y32.dtype == torch.float32 and y32.dim() == 2 and y32.stride(1) == 1 and w.dtype == torch.bfloat16 and w.dim() == 2 and w.stride(1) == 1
    M, K = y32.shape
    N = w.shape[0]
    assert w.shape[1] == K and gamma.dtype == torch.float32 and beta.dtype == torch.float32 and gamma.numel() == K and beta.numel() == K
    out = torch.empty((M, N), dtype=out_dtype, device=y32.device)
    x32 = torch.empty((M, K), dtype=torch.float32, device=y32.device) if want_x32 else None
    ep = _ffi.GemmEpilogue()
    ep.alpha = 1.0
    ep.act = act
    ep.out_fp32 = 1 if out_dtype == torch.float32 else 0
    if bias is not None:
        assert bias.dtype == torch.float32 and bias.numel() == N
        ep.bias = _ffi.dptr(bias)
    if residual is not None:
        assert residual.dtype == torch.float32 and residual.stride(-1) == 1 and residual.shape == (M, N)
        ep.residual, ep.ld_res = _ffi.dptr(residual), residual.stride(-2)
    _ffi.call("adt_ln_gemm_bf16", M, N, K, _ffi.dptr(y32), y32.stride(0), _ffi.dptr(gamma), _ffi.dptr(beta), eps, _ffi.dptr(w), w.stride(0),
              _ffi.dptr(out), out.stride(0), C.byref(ep), _p(x32), K, _ffi.current_stream())
    return out, x32


class reduce_queue:
    """Context manager around adt_reduce_queue_begin / _flush / _end (include/adt_hip.h): inside it the second-stage reductions
    of `layernorm_bwd`, `colsum`, `gemm(colsum_out=...)` and `attn_bwd(bias_grad=...)` on the current stream are queued and
    launched together by `flush()` (and on a clean exit); their outputs are undefined until then.  An exception drops what is
    queued and closes the queue, so the library never keeps a pointer into a freed arena."""

    def __init__(self, arena: torch.Tensor):
        assert arena.dtype == torch.uint8 and arena.is_contiguous()
        self.arena = arena

    def __enter__(self):
        _ffi.call("adt_reduce_queue_begin", _ffi.dptr(self.arena), self.arena.numel(), _ffi.current_stream())
        return self

    def flush(self):
        _ffi.call("adt_reduce_queue_flush")

    def __exit__(self, exc_type, exc, tb):
        _ffi.call("adt_reduce_queue_end", 0 if exc_type is None else 1)
        return False


def embed_pe_fwd(tokens, table, pe, scale, want32=True, want16=True, drop=None):
    """tokens int64 [B, T] -> (y32, y16) [B*T, D]."""
    assert tokens.dtype == torch.int64 and tokens.is_contiguous()
    B, T = tokens.shape
    V, D = table.shape
    y32 = torch.empty((B * T, D), dtype=torch.float32, device=table.device) if want32 else None
    y16 = torch.empty((B * T, D), dtype=torch.bfloat16, device=table.device) if want16 else None
    _keep, dp = _drop_ptr(drop)
    _ffi.call("adt_embed_pe_fwd", _ffi.dptr(tokens), _ffi.dptr(table), _ffi.dptr(pe), scale, _p(y32), _p(y16), B * T, T, D, V, dp,
              _ffi.current_stream())
    return y32, y16


def embed_bwd(tokens, dy, scale, dtable, drop=None, f32=False):
    """dtable[v] = scale * sum of the (dropped) rows of dy whose token is v; dtable is overwritten.  The sum runs as a TN GEMM
    of a one-hot matrix with the bf16 gradient rows (fixed order, like every other weight gradient); vocabularies or widths
    that are not multiples of 8 take the fp32-atomics kernel."""
    n, D = dy.shape
    V = dtable.shape[0]
    assert dy.dtype == torch.float32 and dy.is_contiguous() and dtable.dtype == torch.float32 and dtable.is_contiguous()
    _keep, dp = _drop_ptr(drop)
    if f32 and n > 0 and V % 4 == 0 and D % 4 == 0:              # fp32-operand path: the same one-hot TN GEMM on fp32 operands
        onehot = torch.empty((n, V), dtype=torch.float32, device=dy.device)
        dys = torch.empty((n, D), dtype=torch.float32, device=dy.device)
        _ffi.call("adt_embed_bwd_operands_f32", _ffi.dptr(tokens), _ffi.dptr(dy), scale, _ffi.dptr(onehot), V, _ffi.dptr(dys), n, D, V, dp,
                  _ffi.current_stream())
        gemm(onehot, dys, trans=True, out=dtable)
        return
    if n > 0 and V % 8 == 0 and D % 8 == 0:
        onehot = torch.empty((n, V), dtype=torch.bfloat16, device=dy.device)
        dy16 = torch.empty((n, D), dtype=torch.bfloat16, device=dy.device)
        _ffi.call("adt_embed_bwd_operands", _ffi.dptr(tokens), _ffi.dptr(dy), scale, _ffi.dptr(onehot), V, _ffi.dptr(dy16), n, D, V, dp,
                  _ffi.current_stream())
        gemm(onehot, dy16, trans=True, out=dtable)
        return
    dtable.zero_()
    _ffi.call("adt_embed_bwd", _ffi.dptr(tokens), _ffi.dptr(dy), scale, _ffi.dptr(dtable), n, D, V, dp, _ffi.current_stream())


def cross_entropy(logits, labels, ignore_index=1, want_grad=True, grad_dtype=torch.bfloat16):
    """logits fp32 [M, V], labels int64 [M] -> (loss[1] fp32 on device, dlogits bf16 [M, Vpad] | None).
    ``grad_dtype=float32``: the fp32-operand path (fp32 dlogits [M, V], libm exp / log)."""
    assert logits.dtype == torch.float32 and logits.dim() == 2 and logits.stride(1) == 1
    M, V = logits.shape
    labels = labels.reshape(-1).contiguous()
    loss = torch.empty(1, dtype=torch.float32, device=logits.device)
    if grad_dtype == torch.float32:
        dl = torch.zeros((M, V), dtype=torch.float32, device=logits.device) if want_grad else None
        nb = _ffi.load().adt_cross_entropy_workspace_bytes(M)
        ws = _workspace(nb, logits.device)
        _ffi.call("adt_cross_entropy_f32", _ffi.dptr(logits), logits.stride(0), _ffi.dptr(labels), ignore_index, M, V, _ffi.dptr(loss),
                  _p(dl), V, _ffi.dptr(ws), nb, _ffi.current_stream())
        return loss, dl
    Vp = (V + 7) // 8 * 8
    dl = torch.zeros((M, Vp), dtype=torch.bfloat16, device=logits.device) if want_grad else None
    nb = _ffi.load().adt_cross_entropy_workspace_bytes(M)
    ws = _workspace(nb, logits.device)
    _ffi.call("adt_cross_entropy", _ffi.dptr(logits), logits.stride(0), _ffi.dptr(labels), ignore_index, M, V, _ffi.dptr(loss),
              _p(dl), Vp, _ffi.dptr(ws), nb, _ffi.current_stream())
    return loss, (dl[:, :V] if dl is not None else None)


def cast_bf16(x, want=True, want_t=False):
    """fp32 [R, C] -> (bf16 [R, C] | None, bf16 [C, R] | None)."""
    assert x.dtype == torch.float32 and x.is_contiguous()
    R, Cc = (x.shape if x.dim() == 2 else (1, x.numel()))
    y = torch.empty((R, Cc), dtype=torch.bfloat16, device=x.device) if want else None
    yt = torch.empty((Cc, R), dtype=torch.bfloat16, device=x.device) if want_t else None
    _ffi.call("adt_cast_bf16", _ffi.dptr(x), _p(y), _p(yt), R, Cc, _ffi.current_stream())
    return y, yt


class CastTable:
    """Device table for ``adt_cast_bf16_batched``: every (fp32 weight -> bf16 copy + transposed copy) of a model, one launch."""

    def __init__(self, weights):
        import numpy as np
        dev = weights[0].device
        self.src_ptrs = tuple(w.data_ptr() for w in weights)
        self.y = [torch.empty(w.shape, dtype=torch.bfloat16, device=dev) for w in weights]
        self.y_t = [torch.empty((w.shape[1], w.shape[0]), dtype=torch.bfloat16, device=dev) for w in weights]
        rec = np.zeros(len(weights), dtype=np.dtype([("x", "<u8"), ("y", "<u8"), ("yt", "<u8"), ("rows", "<i4"), ("cols", "<i4")]))
        self.max_tiles = 0
        for i, w in enumerate(weights):
            assert w.dtype == torch.float32 and w.dim() == 2 and w.is_contiguous()
            rec[i] = (w.data_ptr(), self.y[i].data_ptr(), self.y_t[i].data_ptr(), w.shape[0], w.shape[1])
            self.max_tiles = max(self.max_tiles, -(-w.shape[0] // 64) * -(-w.shape[1] // 64))
        self.items = torch.from_numpy(rec.view(np.uint8).reshape(-1).copy()).to(dev)
        self.n = len(weights)

    def matches(self, weights) -> bool:
        return self.src_ptrs == tuple(w.data_ptr() for w in weights)

    def run(self):
        _ffi.call("adt_cast_bf16_batched", _ffi.dptr(self.items), self.n, self.max_tiles, _ffi.current_stream())


def grad_norm(g, max_norm, out=None):
    assert g.dtype == torch.float32 and g.is_contiguous()
    if out is None:
        out = torch.empty(2, dtype=torch.float32, device=g.device)
    nb = _ffi.load().adt_grad_norm_workspace_bytes()
    ws = _workspace(nb, g.device)
    _ffi.call("adt_grad_norm", _ffi.dptr(g), g.numel(), max_norm, _ffi.dptr(out), _ffi.dptr(ws), nb, _ffi.current_stream())
    return out


def adamw_step(p, g, m, v, step, lr, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0, norm_and_clip=None, p_bf16=None,
               nodecay=None):
    """``nodecay``: int64 GPU tensor [n, 2] of sorted flat ranges [lo, hi) (multiples of 4) that get no weight decay."""
    for t in (p, g, m, v):
        assert t.dtype == torch.float32 and t.is_contiguous() and t.numel() == p.numel()
    n_nd = 0
    if nodecay is not None:
        assert nodecay.dtype == torch.int64 and nodecay.dim() == 2 and nodecay.shape[1] == 2 and nodecay.is_contiguous()
        n_nd = nodecay.shape[0]
    _ffi.call("adt_adamw_step", _ffi.dptr(p), _ffi.dptr(g), _ffi.dptr(m), _ffi.dptr(v), _p(p_bf16), p.numel(), lr, beta1, beta2,
              eps, weight_decay, step, _p(norm_and_clip), _p(nodecay) if n_nd else None, n_nd, _ffi.current_stream())


def _attn_desc(B, H, Sq, Sk, q, k, v, o, scale, causal, key_len, mask_value, drop=None, head_dim=128):
    d = _ffi.AttnDesc()
    d.batch, d.heads, d.q_len, d.k_len, d.head_dim = B, H, Sq, Sk, head_dim
    d.causal = 1 if causal else 0
    d.ldq, d.ldk, d.ldv, d.ldo = q.stride(0), k.stride(0), v.stride(0), o.stride(0)
    d.scale, d.mask_value = scale, mask_value
    if drop is not None:
        d.drop.p, d.drop.key = drop
    if key_len is not None:
        assert key_len.dtype == torch.int32 and key_len.numel() == B
        d.key_len = _ffi.dptr(key_len)
    d.f32_products = 1 if (q.dtype == torch.float32 and f32_products == "bf16x3") else 0
    return d


class AttnSaved:
    """What ``attn_fwd(save_bits=True)`` hands to ``attn_bwd`` in place of the bare ``lse``: the log-sum-exp rows and the forward's dropout
    keep decisions as bits (``adt_attn_desc.keep_bits``), so that the backward takes the one-kernel path and hashes no mask again."""
    __slots__ = ("lse", "bits")

    def __init__(self, lse, bits):
        self.lse, self.bits = lse, bits


def keep_bits_to_mask(bits, B, H, Sq, Sk):
    """Decode ``AttnSaved.bits`` (layout: csrc/attn_common.h keep_bits_*) into a bool [B, H, Sq, Sk] keep mask -- tests and tools only
    (the backward reads the words as they are)."""
    nq, nk = (Sq + 255) // 256 * 8, (Sk + 255) // 256 * 8
    words = bits.view(torch.int32).view(B * H, nq, nk, 32)
    word_of_key = torch.tensor([2 * ((kk & 3) + 4 * (kk >> 3)) + ((kk >> 2) & 1) for kk in range(32)], device=bits.device)
    w = words[..., word_of_key]                                                            # [bh, slice, block, key]
    bit = (w.unsqueeze(-1) >> torch.arange(32, device=bits.device, dtype=torch.int32)) & 1  # [bh, slice, block, key, query]
    mask = bit.permute(0, 1, 4, 2, 3).reshape(B * H, nq * 32, nk * 32)[:, :Sq, :Sk]
    return mask.reshape(B, H, Sq, Sk).bool()


def attn_fwd(q, k, v, B, H, Sq, Sk, scale, causal=False, key_len=None, mask_value=-1e4, out=None, drop=None, head_dim=128, save_bits=False):
    """q [B*Sq, >=H*128], k/v [B*Sk, >=H*128] bf16 (row-strided views allowed) -> (o [B*Sq, H*128] bf16, lse [B,H,Sq] fp32).
    fp32 tensors take the fp32-operand path (``head_dim`` 16 / 32 / 64 / 128 there; the bf16 kernels are built for 128).
    ``save_bits`` (bf16 path with dropout and more than 256 keys, or ``"force"``: any key count; ``ADT_ATTN_NO_BITS=1`` switches it off for
    A/B runs): the second result is an ``AttnSaved`` carrying the keep bits for ``attn_bwd``."""
    f32 = q.dtype == torch.float32
    for t in (q, k, v):
        assert t.dtype == q.dtype and t.dtype in (torch.bfloat16, torch.float32) and t.dim() == 2 and t.stride(1) == 1
    if out is None:
        out = torch.empty((B * Sq, H * head_dim), dtype=q.dtype, device=q.device)
    lse = torch.empty((B, H, Sq), dtype=torch.float32, device=q.device)
    d = _attn_desc(B, H, Sq, Sk, q, k, v, out, scale, causal, key_len, mask_value, drop, head_dim)
    bits = None
    # (up to 256 keys -- one key block, the decoder's causal self-attention -- the two-kernel backward is as fast and needs no bits)
    if save_bits and not f32 and drop is not None and drop[0] > 0.0 and Sq > 1 and (Sk > 256 or save_bits == "force") and not os.environ.get("ADT_ATTN_NO_BITS"):
        bits = torch.empty(_ffi.load().adt_attn_keep_bits_bytes(C.byref(d)), dtype=torch.uint8, device=q.device)
        d.keep_bits = _ffi.dptr(bits)
    _ffi.call("adt_attn_fwd_f32" if f32 else "adt_attn_fwd", C.byref(d), _ffi.dptr(q), _ffi.dptr(k), _ffi.dptr(v), _ffi.dptr(out), _ffi.dptr(lse),
              _ffi.current_stream())
    return out, (AttnSaved(lse, bits) if bits is not None else lse)


def attn_bwd(q, k, v, o, dout, lse, dq, dk, dv, B, H, Sq, Sk, scale, causal=False, key_len=None, mask_value=-1e4, drop=None,
             bias_grad=None, head_dim=128):
    """Writes dq/dk/dv (bf16 views with the strides of q/k/v).  ``bias_grad`` (fp32 [3 * H * 128], contiguous): also receives the
    column sums of dq | dk | dv, i.e. the gradient of the in-projection bias."""
    assert dq.stride(0) == q.stride(0) and dk.stride(0) == k.stride(0) and dv.stride(0) == v.stride(0)
    assert dout.stride(0) == o.stride(0)
    d = _attn_desc(B, H, Sq, Sk, q, k, v, o, scale, causal, key_len, mask_value, drop, head_dim)
    if isinstance(lse, AttnSaved):
        d.keep_bits = _ffi.dptr(lse.bits)
        lse = lse.lse
    if q.dtype == torch.float32:                              # fp32-operand path; bias gradients by separate column sums
        nb = _ffi.load().adt_attn_bwd_f32_workspace_bytes(C.byref(d))
        ws = _workspace(nb, q.device)
        _ffi.call("adt_attn_bwd_f32", C.byref(d), _ffi.dptr(q), _ffi.dptr(k), _ffi.dptr(v), _ffi.dptr(o), _ffi.dptr(dout),
                  _ffi.dptr(lse), _ffi.dptr(dq), _ffi.dptr(dk), _ffi.dptr(dv), _ffi.dptr(ws), nb, _ffi.current_stream())
        if bias_grad is not None:
            hd = H * head_dim
            assert bias_grad.dtype == torch.float32 and bias_grad.is_contiguous() and bias_grad.numel() == 3 * hd
            colsum(dq[:, :hd], out=bias_grad[:hd])
            bias_grad[hd:2 * hd].zero_()                      # the rows of dS sum to zero: the key-bias gradient vanishes identically
            colsum(dv[:, :hd], out=bias_grad[2 * hd:])
        return
    if bias_grad is not None:
        hd = H * 128
        assert bias_grad.dtype == torch.float32 and bias_grad.is_contiguous() and bias_grad.numel() == 3 * hd
        base = _ffi.dptr(bias_grad)
        d.dq_colsum, d.dk_colsum, d.dv_colsum = base, base + 4 * hd, base + 8 * hd
    nb = _ffi.load().adt_attn_bwd_workspace_bytes(C.byref(d))
    ws = _workspace(nb, q.device)
    _ffi.call("adt_attn_bwd", C.byref(d), _ffi.dptr(q), _ffi.dptr(k), _ffi.dptr(v), _ffi.dptr(o), _ffi.dptr(dout),
              _ffi.dptr(lse), _ffi.dptr(dq), _ffi.dptr(dk), _ffi.dptr(dv), _ffi.dptr(ws), nb, _ffi.current_stream())


def check_attn_bwd() -> None:
    """Raise if a wave of the one-kernel attention backward gave up waiting for a dQ tile since the last check (``adt_attn_bwd_giveups``:
    a pinned host word the kernels count in -- no synchronisation, a microsecond per call).  The trainers call this at the end of every
    optimisation step: an incomplete dQ must end the run, not train on."""
    n = _ffi.load().adt_attn_bwd_giveups(1)
    if n > 0:
        raise RuntimeError(f"adt_attn_bwd: {n} wave(s) of the one-kernel attention backward gave up waiting for a dQ tile in an earlier "
                           "launch (its dQ was incomplete); the step's gradients are not to be trusted")
