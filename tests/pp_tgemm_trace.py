"""Development tool: per-grid-shape totals of the bf16 training GEMM dispatches (tconv / twgrad) from a rocprofv3 --kernel-trace CSV
(python tests/pp_tgemm_trace.py <..._kernel_trace.csv>): which layers the GEMM time goes to."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
for pat in ("tconv_bf16_kernel<1", "tconv_bf16_kernel<3", "twgrad_bf16_kernel<1", "twgrad_bf16_kernel<3", "attn_bwd", "gn_silu_bwd"):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in rows:
        if pat not in r["Kernel_Name"]:
            continue
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        key = (r["Kernel_Name"].split("(")[0][-28:], r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"])
        agg[key][0] += 1
        agg[key][1] += d
    tot = sum(v[1] for v in agg.values())
    print("== %s: total %.0f us over %d dispatches" % (pat, tot, sum(v[0] for v in agg.values())))
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
        print("   grid %-28s calls %4d  total_us %8.0f  avg_us %7.1f" % (k[1:], v[0], v[1], v[1] / v[0]))

# duration histograms (power-of-two buckets) of the kernels whose grid does not tell the layer size apart
import math
for pat in ("gn_silu_bwd_kernel", "group_norm_kernel", "s4_conv_train_bwd", "s4_conv_train_fwd", "s4_kernel_gen", "ln_bwd", "bias_grad_kernel", "interleave_parity"):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in rows:
        if pat not in r["Kernel_Name"]:
            continue
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        k = int(math.floor(math.log2(max(d, 1.0))))
        agg[k][0] += 1
        agg[k][1] += d
    print("== %s: total %.0f us over %d dispatches; by duration" % (pat, sum(v[1] for v in agg.values()), sum(v[0] for v in agg.values())))
    for k, v in sorted(agg.items()):
        print("   %6d..%-6d us  calls %4d  total_us %8.0f" % (2 ** k, 2 ** (k + 1), v[0], v[1]))
