"""Development tool: per-grid-shape totals of wgrad_mfma_kernel dispatches from a rocprofv3 --kernel-trace CSV
(python tests/pp_wgrad_trace.py <..._kernel_trace.csv>): which layers the weight-gradient time goes to."""
import csv, sys, collections
f = sys.argv[1]
rows = list(csv.DictReader(open(f)))
agg = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    if "wgrad_mfma" not in r["Kernel_Name"]: continue
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    key = (r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"])
    agg[key][0] += 1; agg[key][1] += d
tot = sum(v[1] for v in agg.values())
print("wgrad total us", tot)
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:25]:
    print(k, "calls", v[0], "total_us %.0f" % v[1], "avg_us %.1f" % (v[1] / v[0]))
