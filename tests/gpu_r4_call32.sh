#!/bin/bash
# the M-split form chosen per launch by the host rule (default) against MUGD_CONV_WIDE=0, batch 16 / 8 / 4
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
L=gpurun_out/r4_wide_rule_ab.log
: > $L
for B in 16 8 4; do
if [ $B = 4 ]; then seq="auto 0"; else seq="auto 0 auto 0"; fi
for v in $seq; do
  if [ $v = auto ]; then unset MUGD_CONV_WIDE; else export MUGD_CONV_WIDE=0; fi
  echo "== B=$B MUGD_CONV_WIDE=$v" >> $L
  timeout 200 python tests/gpu_probe.py --B $B --quick --reps 2 2>&1 | grep -E "ddim_eager|ddim_graph|total \(event" >> $L
done
done
cat $L
