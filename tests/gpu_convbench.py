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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pmc", action="store_true")
    ap.add_argument("--shape", type=int, default=0)
    ap.add_argument("--tn", type=int, default=0)
    ap.add_argument("--wk", type=int, default=0)
    a = ap.parse_args()
    lib = get_lib()
    if a.pmc:
        name, B, C, T, M, taps, norm, gated = SHAPES[a.shape]
        copies = max(1, int(300e6 / (M * C * taps * 4)))
        us = lib.dev_bench_conv(B, C, T, M, taps, norm, bool(gated), wk=a.wk, tn=a.tn, copies=copies, iters=20)
        print("%s tn=%d wk=%d cold: %.2f us" % (name, a.tn, a.wk, us))
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
