"""Development tool: mean per launch of every counter in a rocprofv3 --pmc output directory, per kernel (conv_gemm variants merged).
    python tests/pmc_generic_summary.py <dir> [out.txt]"""
import collections
import csv
import glob
import os
import re
import sys


def short(name):
    if "conv_gemm" in name:
        return "conv_gemm (all variants)"
    name = re.sub(r"\(anonymous namespace\)::", "", re.sub(r"^void ", "", name))
    return re.sub(r"\(.*$", "", name)[:40]


def main():
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for path in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for r in csv.DictReader(f):
                agg[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    counters = sorted({c for cs in agg.values() for c in cs})
    lines = ["%-40s %8s " % ("kernel", "launches") + " ".join("%22s" % c for c in counters)]
    for name, cs in sorted(agg.items(), key=lambda kv: -len(next(iter(kv[1].values())))):
        n = len(next(iter(cs.values())))
        lines.append("%-40s %8d " % (name, n) + " ".join("%22.1f" % (sum(cs.get(c, [0])) / max(len(cs.get(c, [0])), 1)) for c in counters))
    text = "\n".join(lines)
    print(text)
    if len(sys.argv) > 2:
        with open(sys.argv[2], "w") as f:
            f.write(text + "\n")


if __name__ == "__main__":
    main()
