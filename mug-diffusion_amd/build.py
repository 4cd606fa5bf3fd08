"""Builds libmugd.so (HIP, gfx950) in-tree with hipcc.  `python build.py` or build.build().

`build_emulated()` compiles the same sources against tests/emu (a functional CPU emulation
of the HIP subset the kernels use) into tests/emu/libmugd_emu.so -- TEST INFRASTRUCTURE for
the GPU-less authoring container; the product never loads it.
"""
import concurrent.futures
import glob
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libmugd.so")
TL_DIR = os.path.join(ROOT, "tests", "tl")
TL_LIB = os.path.join(TL_DIR, "libmugd_tl.so")
EMU_DIR = os.path.join(ROOT, "tests", "emu")
EMU_LIB = os.path.join(EMU_DIR, "libmugd_emu.so")


def _sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _deps():
    return _sources() + glob.glob(os.path.join(CSRC, "*.h")) + [os.path.join(ROOT, "include", "mugd.h")]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _run(cmd):
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("command failed: %s\n%s" % (" ".join(cmd), r.stdout))
    return r.stdout


def build_variant(name, defines, verbose=True):
    """DEVELOPMENT A/B builds: the same sources with extra -D flags into tests/var/<name>/libmugd.so (git-ignored; loaded through
    MUGD_LIB_PATH by the GPU probes, never by the product)."""
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    outdir = os.path.join(ROOT, "tests", "var", name)
    objdir = os.path.join(outdir, "build")
    os.makedirs(objdir, exist_ok=True)
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast", "-Wno-unused-result"] + ["-D" + d for d in defines]

    def cc(src):
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        _run([hipcc] + flags + ["-c", src, "-o", obj])
        return obj
    with concurrent.futures.ThreadPoolExecutor(max_workers=8) as ex:
        objs = list(ex.map(cc, _sources()))
    lib = os.path.join(outdir, "libmugd.so")
    _run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", lib])
    if verbose:
        print("built", lib)
    return lib


def build(force=False, verbose=True, timeline=False):
    """hipcc --offload-arch=gfx950 for every csrc/*.hip, linked into libmugd.so next to this file.

    timeline=True builds the DEVELOPMENT variant instead (same sources with -DMUGD_TL: per-wave phase stamps in the
    conv_gemm kernels, csrc/common.h) into tests/tl/libmugd_tl.so; only tests/gpu_timeline.py loads it."""
    LIB = TL_LIB if timeline else globals()["LIB"]
    if not force and not _stale(LIB, _deps()):
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: libmugd.so cannot be built (there is no CPU fallback)")
    objdir = os.path.join(TL_DIR, "build") if timeline else os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast", "-Wno-unused-result"]
    if timeline:
        flags.append("-DMUGD_TL=1")

    def cc(src):
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        if force or _stale(obj, [src] + glob.glob(os.path.join(CSRC, "*.h")) + [os.path.join(ROOT, "include", "mugd.h")]):
            _run([hipcc] + flags + ["-c", src, "-o", obj])
        return obj

    with concurrent.futures.ThreadPoolExecutor(max_workers=8) as ex:
        objs = list(ex.map(cc, _sources()))
    tmp = LIB + ".tmp.%d" % os.getpid()              # never expose a half-linked library to a concurrent loader
    _run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", tmp])
    os.replace(tmp, LIB)
    if verbose:
        print("built", LIB)
    return LIB


def build_emulated(force=False, verbose=True):
    deps = _deps() + glob.glob(os.path.join(EMU_DIR, "*.cpp")) + glob.glob(os.path.join(EMU_DIR, "include", "hip", "*.h"))
    if not force and not _stale(EMU_LIB, deps):
        return EMU_LIB
    cxx = "/opt/rocm/lib/llvm/bin/clang++"
    if not os.path.exists(cxx):
        cxx = shutil.which("clang++") or shutil.which("g++")
    objdir = os.path.join(EMU_DIR, "build")
    os.makedirs(objdir, exist_ok=True)
    flags = ["-std=c++17", "-O2", "-g", "-fPIC", "-I", os.path.join(EMU_DIR, "include"), "-Wno-unused-value"]

    def cc(src):
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        _run([cxx] + flags + ["-x", "c++", "-c", src, "-o", obj])
        return obj

    with concurrent.futures.ThreadPoolExecutor(max_workers=8) as ex:
        objs = list(ex.map(cc, _sources() + [os.path.join(EMU_DIR, "emu_runtime.cpp")]))
    _run([cxx, "-shared", "-fPIC"] + objs + ["-o", EMU_LIB])
    if verbose:
        print("built", EMU_LIB)
    return EMU_LIB


if __name__ == "__main__":
    if "--variant" in sys.argv:
        i = sys.argv.index("--variant")
        build_variant(sys.argv[i + 1], sys.argv[i + 2:])
    elif "--emu" in sys.argv:
        build_emulated(force="--force" in sys.argv)
    elif "--tl" in sys.argv:
        build(force="--force" in sys.argv, timeline=True)
    else:
        build(force="--force" in sys.argv)
