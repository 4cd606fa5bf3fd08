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


def test_error_paths_return_status_codes_not_exceptions():
    """Nothing throws across the C boundary: bad arguments come back as negative mugd_status values with a message
    (include/mugd.h conventions), here through the emulated build (same host code)."""
    import torch
    from conftest import emu_lib
    from mug import _native
    lib = emu_lib()
    dll = lib.dll
    assert dll.mugd_set_conv_tiling(lib.ctx, 3, 0) == -2 and b"wk" in dll.mugd_last_error(lib.ctx)
    assert dll.mugd_set_conv_tiling(lib.ctx, 0, 24) == -2
    assert dll.mugd_unet_forward(None, None, None, None, 0, None, 0, None, 0, 0) == -2
    x = torch.zeros(1, 24, 8)                                   # 24 channels: not a multiple of the 16-channel K-chunk
    w = torch.zeros(8, 24, 1)
    with pytest.raises(_native.MugdError, match="multiple of 16"):
        lib.op_conv1d(x, w)
    with pytest.raises(_native.MugdError, match="divisible"):
        lib.op_group_norm(torch.zeros(1, 30, 8), torch.ones(30), torch.zeros(30), 4, 0)
    cfg = dict(in_channels=16, model_channels=32, out_channels=16, num_res_blocks=1, attention_resolutions=[2], channel_mult=[1, 2],
               num_heads=2, context_dim=32, audio_channels=[32, 32], s4_layer=False)
    net = lib.unet(cfg)                                          # no parameters registered
    with pytest.raises(_native.MugdError, match="missing parameter"):
        net.forward(torch.zeros(1, 16, 8), torch.zeros(1, dtype=torch.long), torch.zeros(1, 32, 3),
                    [torch.zeros(1, 32, 8), torch.zeros(1, 32, 4)])
    with pytest.raises(_native.MugdError, match="divisible"):   # latent length must be a multiple of 2^(levels-1)
        net.forward(torch.zeros(1, 16, 7), torch.zeros(1, dtype=torch.long), torch.zeros(1, 32, 3),
                    [torch.zeros(1, 32, 7), torch.zeros(1, 32, 3)])
