#!/bin/bash
# staged ring fill: chunks requested before the first park = 1 (default build) | 2 | 4 (the old order); KARG_WARM in all three
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
L=gpurun_out/r4_ringhead_ab.log
: > $L
for B in 4 8; do
for v in head1 head4 head2 head1 head4 head2; do
  if [ $v = head1 ]; then unset MUGD_LIB_PATH; else export MUGD_LIB_PATH=$PWD/tests/var/$v/libmugd.so; fi
  echo "== B=$B lib=$v" >> $L
  timeout 300 python tests/gpu_probe.py --B $B --quick 2>&1 | grep -E "ms/step|total \(event" >> $L
done
done
unset MUGD_LIB_PATH
cat $L
