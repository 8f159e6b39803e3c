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


def test_product_has_no_cpu_path():
    import torch
    from adt_str_amd.frontend import ComputeMelSpectrogram
    m = ComputeMelSpectrogram(16000, 2048, 0.01, 128)
    with pytest.raises(RuntimeError, match="GPU tensors only"):
        m(torch.zeros(1, 8000))
