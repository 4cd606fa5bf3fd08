#!/bin/bash
# code-size A/B: the same plain launches through the 93 KB kernel and a 19 KB single-path build of it (tests/var/slim)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
L=gpurun_out/r4_codesize_ab.log
: > $L
for v in full slim full slim; do
  if [ $v = slim ]; then export MUGD_LIB_PATH=$PWD/tests/var/slim/libmugd.so; else unset MUGD_LIB_PATH; fi
  echo "== lib=$v" >> $L
  timeout 200 python tests/gpu_convbench.py --sweep --plain 2>&1 | grep -v amdgpu.ids >> $L
done
cat $L
