"""The C-ABI library loads and exports exactly what include/adt_hip.h declares
(no compute calls: runs without a GPU)."""
import ctypes as C
import os
import re

import pytest

from adt_str_amd import _ffi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    text = open(os.path.join(ROOT, "include", "adt_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(adt_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(_ffi.lib_path()):
        import __graft_entry__
        __graft_entry__.build()
    return C.CDLL(_ffi.lib_path())


def test_header_and_binding_agree():
    assert declared_functions() == sorted(_ffi.SIGNATURES)


def test_library_exports_every_declared_symbol(lib):
    for name in declared_functions():
        assert hasattr(lib, name), f"{name} declared in adt_hip.h but not exported by libadt_hip.so"


def test_version_and_error_string(lib):
    lib.adt_version.restype = C.c_int
    lib.adt_last_error.restype = C.c_char_p
    assert lib.adt_version() == _ffi.ABI_VERSION
    assert isinstance(lib.adt_last_error(), bytes)


def test_argument_validation_needs_no_gpu(lib):
    """Null pointers / bad shapes are rejected before any HIP call."""
    fn = lib.adt_logmel_f32
    fn.restype = C.c_int
    fn.argtypes = _ffi.SIGNATURES["adt_logmel_f32"]
    assert fn(None, 1, 4096, 4096, 2048, 160, 7, 1, None, None, None, 128, 0, 1e-10, -23.0, 12.0, None, None) == -1
    one = C.c_void_p(16)
    assert fn(one, 1, 4096, 4096, 1024, 160, 7, 1, one, one, one, 128, 0, 1e-10, -23.0, 12.0, one, None) == -2
    assert b"n_fft" in lib.adt_last_error()


def test_round6_entry_points_reject_bad_arguments_without_a_gpu(lib):
    """adt_htsat_layer_block / adt_ln_mean_tokens / adt_clap_logmel_db_ptrs_f32: null pointers, unsupported shapes and the missing bf16 bias
    table of the C = 192 layer come back as error codes (-1 invalid argument, -2 shape) before any HIP call."""
    lib.adt_last_error.restype = C.c_char_p
    one = C.c_void_p(16)

    def bind(name):
        fn = getattr(lib, name)
        fn.restype = C.c_int
        fn.argtypes = _ffi.SIGNATURES[name]
        return fn

    layer = bind("adt_htsat_layer_block")
    assert layer(None, 1, 16, 384, 16, 0, 1e-5, one, one, one, one, 1, 0.2, one, 48, one, one, None, None) == -1
    assert layer(one, 1, 16, 768, 32, 0, 1e-5, one, one, one, one, 1, 0.2, one, 96, one, one, None, None) == -2
    assert b"C = 96 / 192 / 384" in lib.adt_last_error()
    assert layer(one, 1, 16, 384, 16, 0, 1e-5, one, one, one, one, 1, 0.2, one, 47, one, one, None, None) == -2          # 4C hidden units: C / 8 tiles
    assert layer(one, 1, 32, 192, 8, 0, 1e-5, one, one, one, one, 1, 0.2, one, 24, one, one, None, None) == -1           # C = 192 needs the bf16 bias table
    assert b"bf16" in lib.adt_last_error()
    assert layer(one, 1, 32, 192, 8, 0, 1e-5, one, one, one, one, 3, 0.2, one, 24, one, one, one, None) == -1            # n_bias_windows: 1 or (R/8)^2
    lnm = bind("adt_ln_mean_tokens")
    assert lnm(None, 1, 64, 768, one, one, 1e-5, one, None, None) == -1
    assert lnm(one, 1, 64, 770, one, one, 1e-5, one, None, None) == -2
    assert lnm(one, 1, 64, 2048, one, one, 1e-5, one, None, None) == -2
    assert lnm(one, 0, 64, 768, one, one, 1e-5, one, None, None) == 0                                                   # empty batch: nothing to do
    ptrs = bind("adt_clap_logmel_db_ptrs_f32")
    assert ptrs(None, one, 1, 480000, 1024, 480, 1001, one, one, one, 64, 100, 1e-10, one, None) == -1
    assert ptrs(one, one, 1, 480000, 2048, 480, 1001, one, one, one, 64, 100, 1e-10, one, None) == -2
    attn = bind("adt_htsat_attn_block")
    assert attn(one, 1, 16, 384, 16, 0, one, None, 1e-5, one, one, one, one, 1, 0.2, None) == -1                         # gamma without beta
    assert b"both LayerNorm parameters or neither" in lib.adt_last_error()


def test_product_has_no_cpu_path():
    import torch
    from adt_str_amd.frontend import ComputeMelSpectrogram
    m = ComputeMelSpectrogram(16000, 2048, 0.01, 128)
    with pytest.raises(RuntimeError, match="GPU tensors only"):
        m(torch.zeros(1, 8000))
