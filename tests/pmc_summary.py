"""Development tool: summarise rocprofv3 --pmc passes of bench.py into profiles/conv_traffic.json.
python tests/pmc_summary.py <dir with *_counter_collection.csv files> [out.json]
FETCH_SIZE / WRITE_SIZE are in KiB-like units of 1024 B on this rocprofv3; FETCH_SIZE is doubled for the conv_gemm kernels
(16 B/lane coalesced streams are tallied at half size on gfx950: MI355X_MICROARCH.md, HBM section)."""
import collections
import csv
import glob
import json
import os
import sys


def main():
    d = sys.argv[1]
    out = sys.argv[2] if len(sys.argv) > 2 else None
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for r in csv.DictReader(f):
                name = r["Kernel_Name"]
                kind = "conv_gemm" if "conv_gemm" in name else name.split("(")[0][-40:]
                agg[kind][r["Counter_Name"]].append(float(r["Counter_Value"]))
    res = {}
    for kind, cs in agg.items():
        res[kind] = {c: {"launches": len(v), "mean": sum(v) / len(v)} for c, v in cs.items()}
    conv = res.get("conv_gemm", {})
    summary = {"per_kernel": res}
    if "FETCH_SIZE" in conv:
        fetch = conv["FETCH_SIZE"]["mean"] * 1024 * 2.0
        write = conv.get("WRITE_SIZE", {}).get("mean", 0.0) * 1024
        summary.update({"hbm_bytes_per_launch": fetch + write, "fetch_bytes_per_launch_corrected_x2": fetch, "write_bytes_per_launch": write,
                        "note": "mean over all conv_gemm launches of the profiled bench run; FETCH_SIZE*1024*2 (gfx950 wide-load correction) + WRITE_SIZE*1024"})
    print(json.dumps(summary, indent=1)[:3000])
    if out:
        with open(out, "w") as f:
            json.dump(summary, f, indent=1)


if __name__ == "__main__":
    main()
