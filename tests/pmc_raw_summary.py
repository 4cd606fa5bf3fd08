"""Development tool: raw per-kernel sums of every counter found in rocprofv3 --pmc output directories.
    python tests/pmc_raw_summary.py <dir> [name filter]"""
import collections
import csv
import glob
import os
import re
import sys


def main():
    d = sys.argv[1]
    filt = sys.argv[2] if len(sys.argv) > 2 else ""
    agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for r in csv.DictReader(f):
                name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
                name = re.sub(r"^void ", "", name)
                name = re.sub(r"\(.*$", "", name)[:60]
                if filt and filt not in name:
                    continue
                e = agg[name][r["Counter_Name"]]
                e[0] += float(r["Counter_Value"]); e[1] += 1
    for name, cs in agg.items():
        print(name)
        for c, (v, n) in sorted(cs.items()):
            print("    %-34s %16.0f total  %14.1f per launch  (%d launches)" % (c, v, v / max(n, 1), n))


if __name__ == "__main__":
    main()
