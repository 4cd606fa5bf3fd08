"""Development tool: inter-kernel gaps of a rocprofv3 --kernel-trace CSV (python tests/pp_kernel_gaps.py <kernel_trace.csv> [label]).
Kernels are ordered by start time; gap = start[i+1] - end[i].  Prints: kernel count, busy time, gap totals, the gap histogram and
the largest gap classes by the PRECEDING kernel's name -- the evidence behind "graph replay vs eager launches" (round-2 verdict #4)."""
import collections
import csv
import sys


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    label = sys.argv[2] if len(sys.argv) > 2 else sys.argv[1]
    ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
    # keep the last 60 % of the dispatches: warm-up calls and one-off packing kernels are in front
    ev = ev[int(len(ev) * 0.4):]
    busy = sum(e - s for s, e, _ in ev)
    span = ev[-1][1] - ev[0][0]
    gaps = [(ev[i + 1][0] - ev[i][1], ev[i][2], ev[i + 1][2]) for i in range(len(ev) - 1)]
    pos = [g for g in gaps if g[0] > 0]
    print("== %s: %d kernels, span %.2f ms, busy %.2f ms (%.1f %%), sum of positive gaps %.2f ms, overlapping pairs %d"
          % (label, len(ev), span / 1e6, busy / 1e6, 100.0 * busy / span, sum(g[0] for g in pos) / 1e6, len(gaps) - len(pos)))
    hist = collections.Counter()
    for g, _, _ in gaps:
        b = "<0" if g < 0 else "0-0.5us" if g < 500 else "0.5-1us" if g < 1000 else "1-2us" if g < 2000 else "2-5us" if g < 5000 else "5-20us" if g < 20000 else ">20us"
        hist[b] += 1
    print("   gap histogram:", {k: hist[k] for k in ("<0", "0-0.5us", "0.5-1us", "1-2us", "2-5us", "5-20us", ">20us") if hist[k]})
    sg = sorted(g[0] for g in gaps)
    print("   gap median %.2f us, mean %.2f us, p90 %.2f us" % (sg[len(sg) // 2] / 1e3, sum(sg) / len(sg) / 1e3, sg[int(len(sg) * 0.9)] / 1e3))
    by = collections.defaultdict(lambda: [0, 0])
    for g, a, b in gaps:
        k = (a.split("(")[0][-60:], b.split("(")[0][-60:])
        by[k][0] += 1
        by[k][1] += g
    for k, v in sorted(by.items(), key=lambda kv: -kv[1][1])[:12]:
        print("   %8.1f us total  %6d x %6.2f us   after %s -> before %s" % (v[1] / 1e3, v[0], v[1] / v[0] / 1e3, k[0], k[1]))
    durs = collections.defaultdict(lambda: [0, 0])
    for s, e, n in ev:
        durs[n.split("(")[0][-70:]][0] += 1
        durs[n.split("(")[0][-70:]][1] += e - s
    for k, v in sorted(durs.items(), key=lambda kv: -kv[1][1])[:8]:
        print("   kernel %-72s %6d x %7.2f us" % (k, v[0], v[1] / v[0] / 1e3))


if __name__ == "__main__":
    main()
