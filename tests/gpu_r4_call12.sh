#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for tall in 0 1 auto; do
  if [ $tall = auto ]; then unset MUGD_CONV_TALL; else export MUGD_CONV_TALL=$tall; fi
  for B in 16 8 4; do
    timeout 200 python tests/gpu_probe.py --B $B --quick > gpurun_out/r4_h3tall_${tall}_b$B.txt 2>&1
    echo "tall=$tall B=$B: $(grep -E 'ddim_graph' gpurun_out/r4_h3tall_${tall}_b$B.txt | cut -c1-90) | $(grep -E '  conv_gemm  ' gpurun_out/r4_h3tall_${tall}_b$B.txt | cut -c1-100)"
  done
done
