#!/bin/bash
# A/B: pair-plane H3 windows (new) vs HEAD (tests/var/head), one box
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
L=gpurun_out/r4_pair_ab.log
: > $L
( timeout 600 python -m pytest tests/test_ops.py tests/test_nets.py -x -q -m gpu -p no:cacheprovider 2>&1 | tail -5 ) >> $L 2>&1
for B in 4 8 16; do
  for v in new head new head; do
    if [ $v = head ]; then export MUGD_LIB_PATH=$PWD/tests/var/head/libmugd.so; else unset MUGD_LIB_PATH; fi
    echo "== B=$B lib=$v" >> $L
    timeout 300 python tests/gpu_probe.py --B $B --quick 2>&1 | grep -E "ms/step|unet fwd|ddim|clock" >> $L
  done
done
unset MUGD_LIB_PATH
cat $L
