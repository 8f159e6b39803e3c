"""ctypes binding of ``libadt_hip.so`` (C ABI: ``include/adt_hip.h``).

The library is built in-tree by ``__graft_entry__.build()`` /
``make -C adt_str_amd/csrc``.  Loading is lazy and fails loudly: there is no
fallback implementation behind these calls.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

# ``ADT_LIB_PATH``: another build of the SAME library (same ABI version, checked at load) -- for same-box A/B runs of two builds of a kernel
_LIB_PATH = os.environ.get("ADT_LIB_PATH") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "libadt_hip.so")
if os.environ.get("ADT_LIB_PATH"):          # said on stderr so that a bench line produced with another build can be traced
    import sys
    print(f"adt_str_amd: ADT_LIB_PATH overrides the in-tree library: {_LIB_PATH}", file=sys.stderr)
_lock = threading.Lock()
_lib = None

ABI_VERSION = 19

i32, i64, f32, ptr = C.c_int32, C.c_int64, C.c_float, C.c_void_p

class Dropout(C.Structure):
    """struct adt_dropout (include/adt_hip.h)."""
    _fields_ = [("p", C.c_float), ("key", C.c_uint32)]


class GemmEpilogue(C.Structure):
    """struct adt_gemm_epilogue (include/adt_hip.h)."""
    _fields_ = [("bias", C.c_void_p), ("gelu_grad_of", C.c_void_p), ("ld_gelu_grad", C.c_int64),
                ("pre_act_out", C.c_void_p), ("ld_pre_act", C.c_int64), ("residual", C.c_void_p),
                ("ld_res", C.c_int64), ("res_row_mod", C.c_int32), ("act", C.c_int32), ("alpha", C.c_float),
                ("out_fp32", C.c_int32), ("aux_bf16_out", C.c_void_p), ("ld_aux", C.c_int64), ("drop", Dropout),
                ("drop_after_residual", C.c_int32), ("colsum_out", C.c_void_p), ("act_grad_mode", C.c_int32),
                ("res_ln_mean", C.c_void_p), ("res_ln_rstd", C.c_void_p), ("res_ln_gamma", C.c_void_p), ("res_ln_beta", C.c_void_p),
                ("side_fp32", C.c_int32)]


class GemmTnItem(C.Structure):
    """struct adt_gemm_tn_item (include/adt_hip.h)."""
    _fields_ = [("A", C.c_void_p), ("lda", C.c_int64), ("B", C.c_void_p), ("ldb", C.c_int64), ("C", C.c_void_p), ("ldc", C.c_int64),
                ("M", C.c_int64), ("N", C.c_int64), ("K", C.c_int64)]


class AttnDesc(C.Structure):
    """struct adt_attn_desc (include/adt_hip.h)."""
    _fields_ = [("batch", C.c_int32), ("heads", C.c_int32), ("q_len", C.c_int32), ("k_len", C.c_int32),
                ("head_dim", C.c_int32), ("causal", C.c_int32), ("ldq", C.c_int64), ("ldk", C.c_int64),
                ("ldv", C.c_int64), ("ldo", C.c_int64), ("scale", C.c_float), ("mask_value", C.c_float),
                ("key_len", C.c_void_p), ("drop", Dropout), ("dq_colsum", C.c_void_p), ("dk_colsum", C.c_void_p),
                ("dv_colsum", C.c_void_p), ("keep_bits", C.c_void_p), ("f32_products", C.c_int32)]


class AffWeights(C.Structure):
    """struct adt_aff_weights (include/adt_hip.h)."""
    _fields_ = [(n, C.c_void_p) for n in ("proj_w", "proj_b", "conv_w", "conv_b", "local_w1", "local_b1", "local_w2", "local_b2",
                                          "global_w1", "global_b1", "global_w2", "global_b2", "ln_gamma", "ln_beta")]


# name -> argtypes; every entry must be declared in include/adt_hip.h (tests check both ways)
SIGNATURES = {
    "adt_version": [],
    "adt_last_error": [],
    "adt_debug_occupy": [i32, i32, i32, ptr],
    "adt_logmel_f32": [ptr, i64, i64, i64, i32, i32, i32, i32, ptr, ptr, ptr, i32, i32, f32, f32, f32, ptr, ptr],
    "adt_mix_workspace_bytes": [i64, i64],
    "adt_gemm_workspace_bytes": [i32, i64, i64, i64],
    "adt_gemm_bf16": [i32, i64, i64, i64, ptr, i64, ptr, i64, ptr, i64, ptr, ptr, C.c_size_t, ptr],
    "adt_gemm_bf16_tn_grouped": [ptr, i32, ptr],
    "adt_ln_gemm_bf16": [i64, i64, i64, ptr, i64, ptr, ptr, f32, ptr, i64, ptr, i64, ptr, ptr, i64, ptr],
    "adt_attn_fwd": [ptr, ptr, ptr, ptr, ptr, ptr, ptr],
    "adt_attn_keep_bits_bytes": [ptr],
    "adt_attn_bwd_workspace_bytes": [ptr],
    "adt_attn_bwd": [ptr, ptr, ptr, ptr, ptr, ptr, ptr, ptr, ptr, ptr, ptr, C.c_size_t, ptr],
    "adt_attn_bwd_giveups": [i32],
    "adt_layernorm_fwd": [ptr, i64, ptr, ptr, f32, ptr, ptr, i64, ptr, ptr, i64, i64, ptr, ptr],
    "adt_layernorm_bwd_workspace_bytes": [i64, i64],
    "adt_layernorm_bwd": [ptr, i64, ptr, i64, ptr, ptr, ptr, ptr, ptr, i64, ptr, ptr, ptr, i64, i64, ptr, ptr, ptr, C.c_size_t, ptr],
    "adt_colsum_workspace_bytes": [i64, i64],
    "adt_gemm_colsum_workspace_bytes": [i64, i64],
    "adt_colsum_bf16": [ptr, i64, i64, i64, ptr, ptr, C.c_size_t, ptr],
    "adt_greedy_step": [ptr, i64, i64, i64, ptr, ptr, i64, ptr, ptr, ptr, ptr, i64, i64, ptr],
    "adt_bilinear_resize_f32": [ptr, i64, i64, i64, ptr, i64, i64, i64, ptr],
    "adt_reduce_queue_begin": [ptr, C.c_size_t, ptr],
    "adt_reduce_queue_flush": [],
    "adt_reduce_queue_end": [i32],
    "adt_wav_probe_batch": [ptr, i32, i32, ptr],
    "adt_wav_decode_batch": [ptr, i32, i32, ptr, ptr, i32, ptr, ptr],
    "adt_copy_files": [ptr, ptr, i32, i32, ptr],
    "adt_embed_pe_fwd": [ptr, ptr, ptr, f32, ptr, ptr, i64, i64, i64, i64, ptr, ptr],
    "adt_embed_bwd": [ptr, ptr, f32, ptr, i64, i64, i64, ptr, ptr],
    "adt_embed_bwd_operands": [ptr, ptr, f32, ptr, i64, ptr, i64, i64, i64, ptr, ptr],
    "adt_cross_entropy_workspace_bytes": [i64],
    "adt_cross_entropy": [ptr, i64, ptr, i64, i64, i64, ptr, ptr, i64, ptr, C.c_size_t, ptr],
    "adt_gemm_f32_workspace_bytes": [i32, i64, i64, i64],
    "adt_gemm_f32": [i32, i64, i64, i64, ptr, i64, ptr, i64, ptr, i64, ptr, ptr, C.c_size_t, ptr],
    "adt_split_bf16x2": [ptr, i64, i64, i64, ptr, i64, i64, i32, ptr],
    "adt_gemm_bf16x3_supported": [i32, i64, i64, i64],
    "adt_gemm_bf16x3_workspace_bytes": [i32, i64, i64, i64],
    "adt_gemm_bf16x3": [i32, i64, i64, i64, ptr, i64, i64, ptr, i64, i64, ptr, i64, ptr, ptr, C.c_size_t, ptr],
    "adt_attn_fwd_f32": [ptr, ptr, ptr, ptr, ptr, ptr, ptr],
    "adt_attn_bwd_f32_workspace_bytes": [ptr],
    "adt_attn_bwd_f32": [ptr, ptr, ptr, ptr, ptr, ptr, ptr, ptr, ptr, ptr, ptr, C.c_size_t, ptr],
    "adt_colsum_f32_workspace_bytes": [i64, i64],
    "adt_colsum_f32": [ptr, i64, i64, i64, ptr, ptr, C.c_size_t, ptr],
    "adt_layernorm_bwd_f32": [ptr, i64, ptr, i64, ptr, ptr, ptr, ptr, ptr, i64, ptr, ptr, ptr, i64, i64, ptr, ptr, ptr, C.c_size_t, ptr],
    "adt_cross_entropy_f32": [ptr, i64, ptr, i64, i64, i64, ptr, ptr, i64, ptr, C.c_size_t, ptr],
    "adt_embed_bwd_operands_f32": [ptr, ptr, f32, ptr, i64, ptr, i64, i64, i64, ptr, ptr],
    "adt_cast_bf16": [ptr, ptr, ptr, i64, i64, ptr],
    "adt_cast_bf16_batched": [ptr, i32, i32, ptr],
    "adt_grad_norm_workspace_bytes": [],
    "adt_grad_norm": [ptr, i64, f32, ptr, ptr, C.c_size_t, ptr],
    "adt_adamw_step": [ptr, ptr, ptr, ptr, ptr, i64, f32, f32, f32, f32, f32, i64, ptr, ptr, i32, ptr],
    "adt_clap_logmel_db_f32": [ptr, ptr, i64, i32, i32, i32, i32, ptr, ptr, ptr, i32, i32, f32, ptr, ptr],
    "adt_clap_logmel_db_ptrs_f32": [ptr, ptr, i64, i32, i32, i32, i32, ptr, ptr, ptr, i32, i32, f32, ptr, ptr],
    "adt_htsat_front_f32": [ptr, i64, i64, i32, i32, i32, i32, ptr, ptr, ptr, ptr],
    "adt_htsat_patch_embed": [ptr, i64, i32, ptr, ptr, ptr, ptr, f32, i32, ptr, ptr, ptr],
    "adt_htsat_fusion_embed_workspace_bytes": [i32, i32],
    "adt_htsat_fusion_embed": [ptr, ptr, i32, ptr, f32, i32, i32, ptr, C.c_size_t, ptr, ptr],
    "adt_window_attn_fwd": [ptr, i64, ptr, i64, ptr, i32, i64, i32, i32, i32, i32, f32, ptr],
    "adt_patch_merge_ln": [ptr, i64, i32, i32, ptr, ptr, f32, ptr, ptr],
    "adt_mean_tokens": [ptr, i64, i32, i32, ptr, ptr, ptr],
    "adt_ln_mean_tokens": [ptr, i64, i32, i32, ptr, ptr, f32, ptr, ptr, ptr],
    "adt_htsat_rowblock_chunk_tiles": [i32, i32],
    "adt_htsat_attn_block": [ptr, i64, i32, i32, i32, i32, ptr, ptr, f32, ptr, ptr, ptr, ptr, i32, f32, ptr],
    "adt_htsat_layer_block": [ptr, i64, i32, i32, i32, i32, f32, ptr, ptr, ptr, ptr, i32, f32, ptr, i32, ptr, ptr, ptr, ptr],
    "adt_htsat_merge_rowblock": [ptr, i64, i32, i32, ptr, ptr, f32, ptr, i32, ptr, ptr, i64, ptr],
    "adt_htsat_rowblock": [i32, ptr, i64, i32, ptr, i64, ptr, ptr, f32, ptr, i32, ptr, ptr, ptr, i64, ptr],
    "adt_l2_normalize": [ptr, i64, i32, ptr, ptr],
    "adt_cosine_argmax_f32": [ptr, i64, ptr, i64, i64, i64, f32, ptr, ptr, ptr, ptr],
    "adt_resample_f32": [ptr, i64, i64, i64, ptr, ptr, i32, i32, i32, i32, ptr, i64, i64, ptr],
    "adt_mix_render_f32": [ptr, ptr, i64, ptr, i64, ptr, ptr, ptr, i64, i64, ptr, i64, ptr, C.c_size_t, ptr],
    "adt_mix_render_fx_f32": [ptr, ptr, i64, ptr, i64, ptr, ptr, ptr, i64, i64, ptr, i32, ptr, i64, ptr, C.c_size_t, ptr],
}
_RESTYPES = {"adt_last_error": C.c_char_p}
_RESTYPES.update({n: C.c_size_t for n in ("adt_mix_workspace_bytes", "adt_gemm_workspace_bytes", "adt_attn_bwd_workspace_bytes", "adt_attn_keep_bits_bytes",
                                          "adt_layernorm_bwd_workspace_bytes", "adt_colsum_workspace_bytes", "adt_gemm_colsum_workspace_bytes",
                                          "adt_cross_entropy_workspace_bytes", "adt_grad_norm_workspace_bytes",
                                          "adt_htsat_fusion_embed_workspace_bytes",
                                          "adt_attn_bwd_f32_workspace_bytes", "adt_colsum_f32_workspace_bytes", "adt_gemm_f32_workspace_bytes",
                                          "adt_gemm_bf16x3_workspace_bytes")})


class AdtError(RuntimeError):
    """A libadt_hip call returned a negative ADT_E* code."""

    def __init__(self, fn: str, code: int, msg: str):
        super().__init__(f"{fn} failed with code {code}: {msg}")
        self.code = code


def lib_path() -> str:
    return _LIB_PATH


# Entry points without device code (csrc/errors.cpp, csrc/wav_io.cpp).  ``ADT_HOST_ONLY_LIB=<path>`` makes ``load()`` bind THESE from a host-only
# build of the same sources -- the AddressSanitizer harness of the CPU box (``make -C adt_str_amd/csrc asan``, tests/test_sanitize_cpu.py).
# It is not a fallback: every other entry point is absent from such a library and calling one raises.
HOST_ONLY = ("adt_version", "adt_last_error", "adt_wav_probe_batch", "adt_wav_decode_batch", "adt_copy_files")


def load() -> C.CDLL:
    """Load (once) and return the library; raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is None and os.environ.get("ADT_HOST_ONLY_LIB"):
            lib = C.CDLL(os.environ["ADT_HOST_ONLY_LIB"])
            for name in HOST_ONLY:
                fn = getattr(lib, name)
                fn.argtypes = SIGNATURES[name]
                fn.restype = _RESTYPES.get(name, C.c_int)
            if lib.adt_version() != ABI_VERSION:
                raise RuntimeError("host-only library has another ABI version; rebuild it (make -C adt_str_amd/csrc asan)")
            _lib = lib
        if _lib is None:
            if not os.path.exists(_LIB_PATH):
                raise RuntimeError(
                    f"{_LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                    "(or `make -C adt_str_amd/csrc`). adt_str_amd has no fallback path.")
            lib = C.CDLL(_LIB_PATH)
            for name, argtypes in SIGNATURES.items():
                fn = getattr(lib, name)
                fn.argtypes = argtypes
                fn.restype = _RESTYPES.get(name, C.c_int)
            v = lib.adt_version()
            if v != ABI_VERSION:
                raise RuntimeError(f"libadt_hip.so ABI version {v} != expected {ABI_VERSION}; rebuild it")
            _lib = lib
    return _lib


def call(name: str, *args) -> None:
    """Invoke an ``int``-returning entry point and raise ``AdtError`` on failure."""
    lib = load()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        raise AdtError(name, rc, lib.adt_last_error().decode("utf-8", "replace"))


def dptr(t) -> int:
    """Device pointer of a CUDA(HIP) tensor; refuses anything else."""
    if not t.is_cuda:
        raise RuntimeError("adt_str_amd kernels take GPU tensors only (there is no CPU path)")
    return t.data_ptr()


def current_stream() -> int:
    import torch
    return torch.cuda.current_stream().cuda_stream
