"""Development tool (run through gpurun): conv_gemm launch-shape micro-benchmark, hot vs cold weights, tile widths, K-splits.
python tests/gpu_convbench.py [--pmc]  (with --pmc: a single configuration, few launches, for rocprofv3 --pmc passes)"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "mug-diffusion_amd"))

from mug._native import get_lib  # noqa: E402

SHAPES = [  # name, B, C, T, M, taps, norm, gated
    ("res.l3 K=4608", 4, 1536, 64, 512, 3, 1, 0),
    ("res.l3 K=1536", 4, 512, 64, 512, 3, 1, 0),
    ("res.l2 K=4224", 4, 1408, 128, 384, 3, 1, 0),
    ("res.l1 K=3456", 4, 1152, 256, 256, 3, 1, 0),
    ("res.l0 K=1920", 4, 640, 512, 128, 3, 1, 0),
    ("ff1.l3 geglu", 4, 512, 64, 4096, 1, 2, 1),
    ("ff1.l1 geglu", 4, 256, 256, 2048, 1, 2, 1),
    ("ff2.l3", 4, 2048, 64, 512, 1, 0, 0),
    ("qkv.l3", 4, 512, 64, 1536, 1, 2, 0),
    ("proj.l3 1x1", 4, 512, 64, 512, 1, 0, 0),
    ("proj.l1 1x1", 4, 256, 256, 256, 1, 0, 0),
]


# (name, B, C, T, M, taps, norm, gated): the distinct hot layer shapes of the shipped U-Net at batch 4
SWEEP = [
    ("res3 c1 K4608", 4, 1536, 64, 512, 3, 1, 0), ("res3 c2 K1536", 4, 512, 64, 512, 3, 1, 0),
    ("res2 c1 K4224", 4, 1408, 128, 384, 3, 1, 0), ("res2 c2 K1152", 4, 384, 128, 384, 3, 1, 0),
    ("res1 c1 K3456", 4, 1152, 256, 256, 3, 1, 0), ("res1 c2 K768", 4, 256, 256, 256, 3, 1, 0),
    ("res0 c1 K1920", 4, 640, 512, 128, 3, 1, 0), ("res0 c2 K384", 4, 128, 512, 128, 3, 1, 0),
    ("qkv3", 4, 512, 64, 1536, 1, 2, 0), ("qkv2", 4, 384, 128, 1152, 1, 2, 0), ("qkv1", 4, 256, 256, 768, 1, 2, 0),
    ("ff1.3", 4, 512, 64, 4096, 1, 2, 1), ("ff1.2", 4, 384, 128, 3072, 1, 2, 1), ("ff1.1", 4, 256, 256, 2048, 1, 2, 1),
    ("ff2.3", 4, 2048, 64, 512, 1, 0, 0), ("ff2.2", 4, 1536, 128, 384, 1, 0, 0), ("ff2.1", 4, 1024, 256, 256, 1, 0, 0),
    ("proj3", 4, 512, 64, 512, 1, 0, 0), ("proj2", 4, 384, 128, 384, 1, 0, 0), ("proj1", 4, 256, 256, 256, 1, 0, 0),
    ("toq3 ln", 4, 512, 64, 512, 1, 2, 0), ("toq1 ln", 4, 256, 256, 256, 1, 2, 0),
    ("s4glu3", 4, 512, 64, 1024, 1, 0, 1), ("s4glu0", 4, 128, 512, 256, 1, 0, 1),
    ("s4out3 K1536", 4, 512, 64, 512, 3, 0, 0), ("s4out0 K384", 4, 128, 512, 128, 3, 0, 0),
]


def sweep(lib):
    print("%-16s %7s %6s | best (tn,wk) us | auto us | all: tn32 wk1/2/4/8 ; tn16 wk1/2/4/8" % ("shape", "GFLOP", "tiles"))
    for name, B, C, T, M, taps, norm, gated in SWEEP:
        gf = 2.0 * M * C * taps * T * B / 1e9
        Mo = M // 2 if gated else M
        tiles = ((T + 31) // 32) * ((Mo + 31) // 32) * B
        copies = max(1, int(300e6 / (M * C * taps * 4)))
        nch = C // 16
        res = {}
        for tn in (32, 16):
            for wk in (1, 2, 4, 8):
                if wk > nch:
                    continue
                res[(tn, wk)] = lib.dev_bench_conv(B, C, T, M, taps, norm, bool(gated), wk=wk, tn=tn, copies=copies, iters=100)
        auto = lib.dev_bench_conv(B, C, T, M, taps, norm, bool(gated), wk=0, tn=0, copies=copies, iters=100)
        best = min(res, key=res.get)
        row = " ".join("%5.1f" % res.get((32, w), float("nan")) for w in (1, 2, 4, 8)) + " ; " + " ".join("%5.1f" % res.get((16, w), float("nan")) for w in (1, 2, 4, 8))
        print("%-16s %7.3f %6d | (%2d,%d) %5.1f | %5.1f | %s" % (name, gf, tiles, best[0], best[1], res[best], auto, row), flush=True)


def compare(lib):
    """one line per hot shape: the launch the host rules pick (wk = 0, tn = 0), cold weights -- run under two libraries (MUGD_LIB_PATH) and diff"""
    for name, B, C, T, M, taps, norm, gated in SWEEP:
        copies = max(1, int(300e6 / (M * C * taps * 4)))
        us = min(lib.dev_bench_conv(B, C, T, M, taps, norm, bool(gated), wk=0, tn=0, copies=copies, iters=200) for _ in range(3))
        print("%-16s %6.2f us" % (name, us), flush=True)


def forms(lib):
    """round 6: the M-split geometries (waves per workgroup x K-slices; conv_body.h: MS) against the host's own choice on the tall-M launches of
    the batch-4 step (gated projections, q/k/v) -- wk = 0x100 | waves << 4 | kslices forces a form where it exists (include/mugd.h)."""
    cand = [("auto", 0), ("K-split wk2", 2), ("K-split wk4", 4), ("2x1", 0x121), ("4x1", 0x141), ("8x1", 0x181), ("2x2", 0x142), ("4x2", 0x182)]
    print("%-16s %7s | %s" % ("shape", "GFLOP", "  ".join("%11s" % n for n, _ in cand)))
    for name, B, C, T, M, taps, norm, gated in SWEEP:
        if not (name.startswith("ff1") or name.startswith("qkv") or name.startswith("s4glu")):
            continue
        for Bb in (B, 2 * B):
            gf = 2.0 * M * C * taps * T * Bb / 1e9
            copies = max(1, int(300e6 / (M * C * taps * 4)))
            row = []
            for _, wk in cand:
                row.append(min(lib.dev_bench_conv(Bb, C, T, M, taps, norm, bool(gated), wk=wk, tn=32, copies=copies, iters=200) for _ in range(2)))
            print("%-16s %7.3f | %s" % ("%s B%d" % (name, Bb), gf, "  ".join("%8.2f us" % u for u in row)), flush=True)


def icache(lib):
    """round 6: what a launch pays for starting with COLD code.  Per hot shape: back-to-back launches of one kernel (its code stays in the
    instruction caches) against the same launches each preceded by a kernel that walks 128 KB of straight-line code on every CU
    (MUGD_BENCH_THRASH, csrc/k_misc.hip) -- minus that kernel's own time."""
    print("%-16s | warm us | cold us (thrash + conv - thrash alone) | penalty us" % "shape")
    for name, B, C, T, M, taps, norm, gated in SWEEP:
        copies = max(1, int(300e6 / (M * C * taps * 4)))
        r = {}
        for mode in ("0", "1", "2"):
            os.environ["MUGD_BENCH_THRASH"] = mode
            r[mode] = min(lib.dev_bench_conv(B, C, T, M, taps, norm, bool(gated), wk=0, tn=0, copies=copies, iters=100) for _ in range(3))
        os.environ["MUGD_BENCH_THRASH"] = "0"
        print("%-16s | %7.2f | %7.2f (%6.2f - %6.2f) | %+6.2f" % (name, r["0"], r["1"] - r["2"], r["1"], r["2"], r["1"] - r["2"] - r["0"]), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--icache", action="store_true")
    ap.add_argument("--forms", action="store_true")
    ap.add_argument("--sweep", action="store_true")
    ap.add_argument("--compare", action="store_true")
    ap.add_argument("--pmc", action="store_true")
    ap.add_argument("--shape", type=int, default=0)
    ap.add_argument("--tn", type=int, default=0)
    ap.add_argument("--wk", type=int, default=0)
    ap.add_argument("--hot", action="store_true", help="--pmc: one weight copy (L2-resident) instead of a fresh copy per launch")
    ap.add_argument("--B", type=int, default=0, help="--pmc: override the batch")
    a = ap.parse_args()
    if a.compare:
        compare(get_lib())
        return
    lib = get_lib()
    if a.icache:
        return icache(lib)
    if a.forms:
        return forms(lib)
    if a.sweep:
        return sweep(lib)
    if a.pmc:
        name, B, C, T, M, taps, norm, gated = SHAPES[a.shape]
        B = a.B or B
        copies = 1 if a.hot else max(1, int(300e6 / (M * C * taps * 4)))
        us = lib.dev_bench_conv(B, C, T, M, taps, norm, bool(gated), wk=a.wk, tn=a.tn, copies=copies, iters=20)
        print("%s B=%d tn=%d wk=%d %s: %.2f us" % (name, B, a.tn, a.wk, "hot" if a.hot else "cold", us))
        return
    print("%-16s %8s | %s" % ("shape", "GFLOP", "us per launch (TF/s): hot tn32 | hot tn16 | cold tn32 wk8/wk4 | cold tn16 wk8/wk4 | cold no-norm auto"))
    for name, B, C, T, M, taps, norm, gated in SHAPES:
        gf = 2.0 * M * C * taps * T * B / 1e9
        copies = max(1, int(300e6 / (M * C * taps * 4)))
        r = []
        for tn in (32, 16):
            r.append(lib.dev_bench_conv(B, C, T, M, taps, norm, bool(gated), wk=0, tn=tn, copies=1, iters=200))
        for tn in (32, 16):
            for wk in (8, 4):
                r.append(lib.dev_bench_conv(B, C, T, M, taps, norm, bool(gated), wk=wk, tn=tn, copies=copies, iters=200))
        r.append(lib.dev_bench_conv(B, C, T, M, taps, 0, bool(gated), wk=0, tn=0, copies=copies, iters=200))
        print("%-16s %8.3f | " % (name, gf) + " | ".join("%6.1f (%5.1f)" % (u, gf / u * 1e3) for u in r), flush=True)


if __name__ == "__main__":
    main()
