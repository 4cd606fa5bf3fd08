"""The C-ABI library loads and exports every symbol include/mugd.h declares (no compute here)."""
import ctypes
import os

import pytest

import __graft_entry__ as ge
from conftest import load_build_module


def test_hip_library_exports_every_declared_symbol():
    lib = load_build_module().build(verbose=False)          # hipcc cross-compiles for gfx950 without a GPU
    declared = ge.declared_symbols()
    assert len(declared) >= 25
    exported = ge.exported_symbols(lib)
    assert [s for s in declared if s not in exported] == []


def test_python_binding_covers_the_header():
    from mug import _native
    assert sorted(_native.EXPORTS) == ge.declared_symbols()


def test_emulated_build_exports_the_same_abi():
    lib = load_build_module().build_emulated(verbose=False)
    assert [s for s in ge.declared_symbols() if s not in ge.exported_symbols(lib)] == []
    dll = ctypes.CDLL(lib)
    dll.mugd_version.restype = ctypes.c_char_p
    assert b"mugd" in dll.mugd_version()


def test_product_loader_has_no_cpu_fallback():
    """Without a GPU the product entry point must fail loudly instead of computing somewhere else."""
    import torch
    from mug import _native
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(_native.MugdError):
        _native.Lib()
