"""Development tool (run through gpurun): per-launch PHASE TIMELINE of the conv_gemm kernels inside one U-Net evaluation.

Loads tests/tl/libmugd_tl.so -- the product sources compiled with -DMUGD_TL (`python mug-diffusion_amd/build.py --tl`),
in which every wave of conv_gemm (either tile width) stamps s_memtime at: kernel entry, side operands requested, first chunk
parked, K loop done, K-split combine done, output stored, statistics done (csrc/common.h).  Writes one CSV row per launch
(median / max cycles of every phase over the launch's waves, span and start skew from the device-global 100 MHz counter)
and a short text summary.

python tests/gpu_timeline.py --z 512 --B 4 --out gpurun_out/tl_z512_b4.csv [--raw-op 37]
"""
import argparse
import csv
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "mug-diffusion_amd"))

import torch  # noqa: E402

from oracle import cases, weights  # noqa: E402
from mug._native import Lib  # noqa: E402

TL_LIB = os.path.join(ROOT, "tests", "tl", "libmugd_tl.so")


def summarise(path, out):
    rows = list(csv.DictReader(open(path)))
    rows = [r for r in rows if r["op"] != "op"]
    f = lambda r, k: float(r[k])
    n = len(rows)
    mhz = sorted(f(r, "mhz") for r in rows)[n // 2]
    tot = {k: sum(f(r, k) for r in rows) for k in ("setup_med", "first_med", "loop_med", "combine_med", "store_med", "tail_med", "total_med", "total_max")}
    span = sum(f(r, "span_ns") for r in rows)
    print("conv launches: %d   median shader clock %.0f MHz" % (n, mhz), file=out)
    print("sum of kernel spans (first wave in -> last wave out, 100 MHz counter): %.1f us" % (span / 1e3), file=out)
    print("sum over launches of the MEDIAN wave's phases, in us at the median clock (share of the median wave's life):", file=out)
    # round 6: the first segment's operand ring is issued BEFORE the wave waits for / reduces the statistics, so "setup" now ends with the reduce +
    # barrier and contains the ring issue; "first" is what is left until chunk 0 is parked
    for k, name in (("setup_med", "entry -> statistics reduced (kernarg, index math, side + operand requests, sums, barrier)"),
                    ("first_med", "-> first chunk parked (chunk 0 arrives, transform, LDS store)"),
                    ("loop_med", "-> K loop done"), ("combine_med", "-> K-split combine done (2 barriers + LDS)"),
                    ("store_med", "-> outputs stored"), ("tail_med", "-> row / column statistics done")):
        print("  %-66s %8.1f us  %5.1f%%" % (name, tot[k] / mhz, 100 * tot[k] / tot["total_med"]), file=out)
    print("  %-66s %8.1f us" % ("median wave total", tot["total_med"] / mhz), file=out)
    print("  %-66s %8.1f us" % ("slowest wave total", tot["total_max"] / mhz), file=out)
    if "f_issue" in rows[0]:
        fi, fa, fp = (sum(f(r, k) for r in rows) / mhz for k in ("f_issue", "f_arrive", "f_park"))
        print("  in detail: side operands requested -> operand ring issued %.1f us; statistics done -> chunk 0's window arrived %.1f us; transformed + parked %.1f us" % (fi, fa, fp), file=out)
    skew = sum(f(r, "start_skew_ns") for r in rows)
    print("sum of start skews (last wave's entry - first wave's entry): %.1f us" % (skew / 1e3), file=out)
    # MFMA-ideal loop time: cycles per chunk if the SIMD's matrix pipe were the only limit
    print("\nper launch (us at the median clock): span | setup (= statistics requested + epilogue operands requested + [ring issue: see 'issue'] + wait for the sums behind it + reduce / barrier) first loop combine store tail | cyc/chunk | label", file=out)
    for r in rows:
        print("%4s tn%-2s wk%s blk%-5s %6.2f | %5.2f (%4.2f %4.2f %4.2f %4.2f) %5.2f %6.2f %5.2f %5.2f %5.2f | %6.0f | %s" % (
            r["op"], r["tn"], r["wk"], r["blocks"], f(r, "span_ns") / 1e3, f(r, "setup_med") / mhz,
            f(r, "su_issue") / mhz, f(r, "su_side") / mhz, f(r, "su_wait") / mhz, f(r, "su_reduce") / mhz, f(r, "first_med") / mhz,
            f(r, "loop_med") / mhz, f(r, "combine_med") / mhz, f(r, "store_med") / mhz, f(r, "tail_med") / mhz,
            f(r, "cyc_per_chunk"), ("[first: issue %.2f arrive %.2f park %.2f] " % (f(r, "f_issue") / mhz, f(r, "f_arrive") / mhz, f(r, "f_park") / mhz) if "f_issue" in r else "") + r["label"]), file=out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--z", type=int, default=512)
    ap.add_argument("--B", type=int, default=4)
    ap.add_argument("--out", default="gpurun_out/timeline.csv")
    ap.add_argument("--raw-op", type=int, default=-2, help="also dump every wave's raw record of this op index (-1: all)")
    ap.add_argument("--lib", default=TL_LIB, help="another -DMUGD_TL build (build.py --rev <rev> <name> MUGD_TL=1): same-box comparisons")
    a = ap.parse_args()
    if not os.path.exists(a.lib):
        raise SystemExit("build the timeline library first: python mug-diffusion_amd/build.py --tl")
    if a.lib != TL_LIB:
        os.environ["MUGD_LIB_PATH"] = a.lib; os.environ["MUGD_LIB_LENIENT"] = "1"
    lib = Lib(path=a.lib)
    lib.dll.raw.mugd_dev_timeline.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_int]
    lib.dll.raw.mugd_dev_timeline.restype = C.c_int
    case = cases.FULL
    man = weights.load_manifest(os.path.join(cases.GOLDEN, case["manifest"]))
    sd = weights.set_s4_lengths(weights.make_state_dict(man, 0), case["unet"], a.z)
    unet = lib.unet(case["unet"]); unet.set_params(sd, "model.unet_model.")
    dev = lib.device
    x = cases.x_T(1, a.B, a.z).to(dev)
    t = torch.full((a.B,), 501, dtype=torch.long, device=dev)
    c = cases.context(case, 1, a.B).to(dev)
    w = [m.to(dev) for m in cases.audio_maps(case, 1, 1, a.z)]
    for _ in range(3):
        unet.forward(x, t, c, w)
    torch.cuda.synchronize()
    os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
    for p in (a.out, a.out + ".raw"):
        if os.path.exists(p):
            os.remove(p)
    raw = (a.out + ".raw").encode() if (a.raw_op >= -1 or os.environ.get("MUGD_TL_RAW_LABEL")) else None
    lib.check(lib.dll.mugd_dev_timeline(unet.h, a.out.encode(), raw, a.raw_op))
    with open(a.out.replace(".csv", "") + ".txt", "w") as f:
        summarise(a.out, f)
    summarise(a.out, sys.stdout)


if __name__ == "__main__":
    main()
