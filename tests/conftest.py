import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "mug-diffusion_amd")
for p in (PKG, ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run through gpurun)")


def pytest_collection_modifyitems(config, items):
    """Every `gpu`-marked item skips when no GPU is visible to torch (a plain `pytest tests` in the authoring container stays
    green whatever a test forgets to call); on a GPU box nothing is skipped here."""
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="needs a real MI355X (no GPU visible to torch)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


_libs = {}


def load_build_module():
    import importlib.util
    spec = importlib.util.spec_from_file_location("mugd_build", os.path.join(PKG, "build.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    return b


def emu_lib():
    """tests/emu/libmugd_emu.so: the kernel sources compiled against the CPU emulation of HIP
    (test infrastructure; lets the GPU-less container exercise kernel indexing logic)."""
    if "emu" not in _libs:
        path = load_build_module().build_emulated(verbose=False)
        from mug._native import Lib
        _libs["emu"] = Lib(path=path, device="cpu")
    return _libs["emu"]


def real_lib():
    """The HIP build on the GPU.  Without a GPU the `gpu`-marked tests SKIP (a plain `pytest tests` in the authoring
    container stays green); on a GPU box a missing / unloadable library is an error, never a skip."""
    if "gpu" not in _libs:
        import torch
        if not torch.cuda.is_available():
            pytest.skip("needs a real MI355X (no GPU visible to torch)")
        from mug._native import get_lib
        _libs["gpu"] = get_lib()
    return _libs["gpu"]


@pytest.fixture(params=["emu", pytest.param("gpu", marks=pytest.mark.gpu)])
def lib(request):
    """The C-ABI library under test: the emulated build on CPU, or the real HIP build on the GPU."""
    return emu_lib() if request.param == "emu" else real_lib()


@pytest.fixture
def gpu_lib():
    return real_lib()
