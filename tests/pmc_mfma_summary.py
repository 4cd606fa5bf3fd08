"""Development tool: per-kernel matrix-pipe utilisation from one rocprofv3 --pmc pass
(SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVES) over tests/gpu_unet_once.py.
    python tests/pmc_mfma_summary.py <dir with *counter_collection.csv> [out.txt]
MFMA utilisation = (SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs) / (SQ_BUSY_CYCLES / 32 shader engines): the fraction of the kernel's
cycles an average SIMD's matrix pipe was busy (same normalisation as profiles/r1_pmc_conv_res3.txt)."""
import collections
import csv
import glob
import os
import re
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*$", "", name)[:48]


def main():
    d = sys.argv[1]
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for r in csv.DictReader(f):
                agg[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    lines = ["%-48s %7s %14s %14s %12s %9s %9s" % ("kernel", "launches", "MFMA busy/SIMD", "kernel cycles", "VALU issue", "MFMA util", "VALU util")]
    tot_m = tot_c = 0.0
    for name, cs in sorted(agg.items(), key=lambda kv: -sum(kv[1].get("SQ_BUSY_CYCLES", [0]))):
        if "SQ_BUSY_CYCLES" not in cs:
            continue
        n = len(cs["SQ_BUSY_CYCLES"])
        cyc = sum(cs["SQ_BUSY_CYCLES"]) / 32.0
        mfma = sum(cs.get("SQ_VALU_MFMA_BUSY_CYCLES", [0])) / 1024.0
        valu = sum(cs.get("SQ_ACTIVE_INST_VALU", [0])) * 4 / 1024.0
        if "conv_gemm" in name:
            tot_m += mfma
            tot_c += cyc
        lines.append("%-48s %7d %14.0f %14.0f %12.0f %8.1f%% %8.1f%%" % (name, n, mfma / n, cyc / n, valu / n, 100 * mfma / max(cyc, 1), 100 * valu / max(cyc, 1)))
    if tot_c:
        lines.append("all conv_gemm kernels: matrix pipe busy %.1f%% of kernel cycles" % (100 * tot_m / tot_c))
    text = "\n".join(lines)
    print(text)
    if len(sys.argv) > 2:
        with open(sys.argv[2], "w") as f:
            f.write(text + "\n")


if __name__ == "__main__":
    main()
