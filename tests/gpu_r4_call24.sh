#!/bin/bash
# KARG_WARM (touch all kernarg lines at kernel entry) A/B
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
L=gpurun_out/r4_kargwarm_ab.log
: > $L
for B in 4 8; do
for v in warm nowarm warm nowarm; do
  if [ $v = nowarm ]; then export MUGD_LIB_PATH=$PWD/tests/var/nowarm/libmugd.so; else unset MUGD_LIB_PATH; fi
  echo "== B=$B lib=$v" >> $L
  timeout 300 python tests/gpu_probe.py --B $B --quick 2>&1 | grep -E "ms/step|total \(event" >> $L
done
done
unset MUGD_LIB_PATH
cat $L
