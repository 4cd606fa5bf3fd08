#!/bin/bash
# A/B/C: L2-warming touches at kernel entry (0 none | 1 weights | 2 weights + windows), one box
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
L=gpurun_out/r4_touch_ab.log
: > $L
( timeout 600 python -m pytest tests/test_ops.py tests/test_nets.py -x -q -m gpu -p no:cacheprovider 2>&1 | tail -3 ) >> $L 2>&1
for B in 4 8; do
  for v in touch0 touch1 touch2 touch0 touch1 touch2; do
    if [ $v = touch1 ]; then unset MUGD_LIB_PATH; else export MUGD_LIB_PATH=$PWD/tests/var/$v/libmugd.so; fi
    echo "== B=$B lib=$v" >> $L
    timeout 300 python tests/gpu_probe.py --B $B --quick 2>&1 | grep -E "ms/step" >> $L
  done
done
unset MUGD_LIB_PATH
cat $L
