"""Development tool: digest of a rocprofv3 --kernel-trace CSV of tests/gpu_train_probe.py -- the LAST complete training step
(between the last two adamw_chunks_kernel dispatches): busy time by kernel class, idle gaps, runtime copies / fills in the step.
python tests/pp_train_trace.py <dir with *_kernel_trace.csv> [out.txt]"""
import csv
import glob
import os
import re
import sys


def main():
    d = sys.argv[1]
    f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "")) for r in rows))
    marks = [i for i, e in enumerate(ev) if "adamw_chunks_kernel" in e[2]]
    lo, hi = marks[-2] + 1, marks[-1] + 1
    step = ev[lo:hi]
    t0, t1 = step[0][0], step[-1][1]
    out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout
    print("last step: %d dispatches over %.2f ms (first start -> last end)" % (len(step), (t1 - t0) / 1e6), file=out)
    # union of busy intervals (two streams overlap)
    busy, cur_s, cur_e = 0, None, None
    for s, e, _, _ in step:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    print("device busy (union of dispatch intervals) %.2f ms, idle %.2f ms; sum of durations %.2f ms" % (busy / 1e6, (t1 - t0 - busy) / 1e6, sum(e - s for s, e, _, _ in step) / 1e6), file=out)
    agg = {}
    for s, e, n, q in step:
        n = n.replace("void ", "", 1).replace("(anonymous namespace)::", "")
        k = re.split(r"[<(]", n)[0]
        if k.startswith("at::native"):
            k = "torch:" + re.sub(r".*?([A-Za-z_]+Functor|reduce_kernel|copy_kernel|direct_copy).*", r"\1", n)[:40]
        a = agg.setdefault(k, [0, 0])
        a[0] += 1; a[1] += e - s
    for k, v in sorted(agg.items(), key=lambda x: -x[1][1]):
        print("  %-44s %5d launches %8.3f ms  %7.1f us avg" % (k, v[0], v[1] / 1e6, v[1] / 1e3 / v[0]), file=out)
    # gaps: idle time in front of each dispatch, by the class of the dispatch that follows
    gaps, cur_e = {}, step[0][0]
    for s, e, n, q in step:
        if s > cur_e:
            k = re.split(r"[<(]", n.replace("void ", "", 1).replace("(anonymous namespace)::", ""))[0][:44]
            g = gaps.setdefault(k, [0, 0]); g[0] += 1; g[1] += s - cur_e
        cur_e = max(cur_e, e)
    print("idle gaps by the dispatch that ends them:", file=out)
    for k, v in sorted(gaps.items(), key=lambda x: -x[1][1])[:15]:
        print("  %-44s %5d gaps %8.3f ms  %6.2f us avg" % (k, v[0], v[1] / 1e6, v[1] / 1e3 / v[0]), file=out)


if __name__ == "__main__":
    main()
