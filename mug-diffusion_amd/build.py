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


# ---------------------------------------------------------------------------------------------------------------------------
# Register-pressure guard (round 6): a hot kernel that starts spilling gets slower without failing anything, so the build reads every
# kernel's resource record out of the code objects (llvm-readelf --notes: .private_segment_fixed_size = scratch bytes per lane) and FAILS
# when a kernel has scratch that this table does not allow.  An entry is (regex over the demangled name, max bytes, why it is tolerated).
# ---------------------------------------------------------------------------------------------------------------------------
SCRATCH_ALLOWED = [
    # conv_gemm_kernel<WK, DUAL, KIND, NIT, WT, TN, MS>
    (r"conv_gemm_kernel<1, (true|false), 0, 1, (float|unsigned short), 16, 0>", 192,
     "one-wave 16-wide tiles: only launches with a single K chunk AND <= 128 tiles pick them (never on the shipped networks' hot paths)"),
    (r"conv_gemm_kernel<\d, true, 2, 9, float, 32, 0>", 400,
     "gated epilogue over generic windows: a GLU / GEGLU projection at a length that is not a multiple of 4 (cold path, kept for completeness)"),
    (r"conv_gemm_kernel<4, false, 1, 1, float, 32, 0>", 64,
     "dilated ResnetBlock convs of the wave encoder / VAE at 4 waves per tile: run-time transform + 4 halo loads per lane"),
    (r"wgrad_mfma_kernel", 256, "weight gradient of the fp32 (parity) training mode; the bf16 mode of record runs twgrad_bf16_kernel"),
]


def kernel_resources(obj):
    """[(demangled kernel name, vgprs, sgprs, scratch bytes, LDS bytes)] of the gfx950 code object inside a hipcc object file."""
    import re
    import tempfile
    bindir = "/opt/rocm/lib/llvm/bin"
    with tempfile.TemporaryDirectory() as d:
        shutil.copy(obj, os.path.join(d, "o.o"))
        subprocess.run([os.path.join(bindir, "llvm-objdump"), "--offloading", "o.o"], cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        cos = [f for f in os.listdir(d) if "gfx950" in f]
        if not cos:
            return []
        notes = subprocess.run([os.path.join(bindir, "llvm-readelf"), "--notes", os.path.join(d, cos[0])], stdout=subprocess.PIPE, text=True).stdout
    rows = []
    for k in re.split(r"\n\s+- \.agpr_count", notes)[1:]:
        def g(f):
            m = re.search(r"\.%s:\s+(\d+)" % f, k)
            return int(m.group(1)) if m else 0
        rows.append([re.search(r"\.name:\s+(\S+)", k).group(1), g("vgpr_count"), g("sgpr_count"), g("private_segment_fixed_size"), g("group_segment_fixed_size")])
    names = subprocess.run(["c++filt"], input="\n".join(r[0] for r in rows), stdout=subprocess.PIPE, text=True).stdout.split("\n")
    for r, n in zip(rows, names):
        r[0] = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return [tuple(r) for r in rows]


def check_scratch(objs, verbose=True):
    """Raises if a kernel of the library uses more scratch than SCRATCH_ALLOWED grants (everything unlisted: none)."""
    import re
    bad, seen = [], 0
    for obj in sorted(objs):
        for name, vgpr, sgpr, scratch, lds in kernel_resources(obj):
            seen += 1
            if scratch == 0:
                continue
            limit = 0
            for pat, mx, _why in SCRATCH_ALLOWED:
                if re.fullmatch(pat, name.split("(")[0].strip()):
                    limit = max(limit, mx)
            if scratch > limit:
                bad.append("%s: %d bytes of scratch per lane (%d VGPRs; allowed: %d) in %s" % (name, scratch, vgpr, limit, os.path.basename(obj)))
    if bad and os.environ.get("MUGD_GUARD") == "warn":       # development: experiments may link what the guard would refuse
        print("register-pressure guard (MUGD_GUARD=warn):\n  " + "\n  ".join(bad))
        return
    if bad:
        raise RuntimeError("register-pressure guard: kernels spill beyond what build.py: SCRATCH_ALLOWED tolerates\n  " + "\n  ".join(bad))
    if verbose:
        print("register-pressure guard: %d kernels checked, none spills beyond its allowance" % seen)


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


def build_rev(rev, name, verbose=True, defines=()):
    """DEVELOPMENT A/B: the library as of git revision `rev` (its own csrc/ and include/, extracted with `git archive`) into
    tests/var/<name>/libmugd.so -- the "before" arm of same-box comparisons (tests/gpu_run.sh ab:<name>; the binding loads it with
    MUGD_LIB_LENIENT=1 because an older library lacks the newer entry points)."""
    import tarfile
    import io
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    outdir = os.path.join(ROOT, "tests", "var", name)
    src = os.path.join(outdir, "src")
    shutil.rmtree(src, ignore_errors=True)
    os.makedirs(src)
    blob = subprocess.run(["git", "-C", ROOT, "archive", rev, "mug-diffusion_amd/csrc", "include"], stdout=subprocess.PIPE, check=True).stdout
    tarfile.open(fileobj=io.BytesIO(blob)).extractall(src)
    objdir = os.path.join(outdir, "build")
    os.makedirs(objdir, exist_ok=True)
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast", "-Wno-unused-result"] + ["-D" + d for d in defines]

    def cc(s):
        obj = os.path.join(objdir, os.path.basename(s) + ".o")
        _run([hipcc] + flags + ["-c", s, "-o", obj])
        return obj
    with concurrent.futures.ThreadPoolExecutor(max_workers=8) as ex:
        objs = list(ex.map(cc, sorted(glob.glob(os.path.join(src, "mug-diffusion_amd", "csrc", "*.hip")))))
    lib = os.path.join(outdir, "libmugd.so")
    _run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", lib])
    shutil.rmtree(src, ignore_errors=True)
    shutil.rmtree(objdir, ignore_errors=True)
    if verbose:
        print("built", lib, "from", rev)
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
    if not timeline:
        check_scratch(objs, verbose)                 # fails the build when a hot kernel starts spilling
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
    flags = ["-std=c++17", "-O2", "-fPIC", "-I", os.path.join(EMU_DIR, "include"), "-Wno-unused-value"]

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
    if "--rev" in sys.argv:                           # python build.py --rev <git revision> <name>
        i = sys.argv.index("--rev")
        build_rev(sys.argv[i + 1], sys.argv[i + 2], defines=sys.argv[i + 3:])
    elif "--variant" in sys.argv:
        i = sys.argv.index("--variant")
        build_variant(sys.argv[i + 1], sys.argv[i + 2:])
    elif "--resources" in sys.argv:                   # table of every kernel's registers / scratch / LDS (the guard's input)
        for obj in sorted(glob.glob(os.path.join(HERE, "build", "*.hip.o"))):
            for row in kernel_resources(obj):
                print("%-22s vgpr %3d sgpr %3d scratch %4d lds %6d  %s" % (os.path.basename(obj), row[1], row[2], row[3], row[4], row[0][:140]))
    elif "--emu" in sys.argv:
        build_emulated(force="--force" in sys.argv)
    elif "--tl" in sys.argv:
        build(force="--force" in sys.argv, timeline=True)
    else:
        build(force="--force" in sys.argv)
