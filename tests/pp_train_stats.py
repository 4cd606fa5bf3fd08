"""Per-step summary of a rocprofv3 kernel_stats.csv of tests/gpu_train_probe.py (argv: csv, steps): kernel classes, then the top kernels."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 3.0
top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
rows = [r for r in rows if not ('at::native' in r['Name'] or 'copyBuffer' in r['Name'] or 'fillBuffer' in r['Name'])]
def cls(n):
    for key, pats in (('tconv', ['tconv']), ('twgrad', ['twgrad_bf16']), ('attn_bwd', ['attn_bwd']), ('attn_fwd', ['attention_kernel']), ('gn', ['gn_', 'group_norm']),
                      ('ln', ['ln_', 'layer_norm']), ('s4', ['s4_']), ('pack', ['tpack']), ('reduce', ['twgrad_reduce', 'treduce', 'batch_reduce', 'attn_tables_reduce']),
                      ('bias_grad', ['bias_grad']), ('adamw', ['adamw'])):
        if any(p in n for p in pats):
            return key
    return 'other'
acc, calls = {}, {}
for r in rows:
    k = cls(r['Name'])
    acc[k] = acc.get(k, 0) + float(r['TotalDurationNs']) / 1e6 / steps
    calls[k] = calls.get(k, 0) + int(r['Calls']) / steps
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print("%-10s %6.2f ms/step %6d launches/step" % (k, v, calls[k]))
print("total      %6.2f ms/step %6d launches/step" % (sum(acc.values()), sum(calls.values())))
rows.sort(key=lambda r: -float(r['TotalDurationNs']))
for r in rows[:top]:
    print("%-92s %5d %8.1f us  %6.2f ms/step" % (r['Name'].replace('(anonymous namespace)::', '').replace('void ', '')[:92], int(r['Calls']) // steps,
                                                float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / 1e6 / steps))
