#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_ops.py -x -q -m gpu -p no:cacheprovider > gpurun_out/r4_t13.log 2>&1; tail -2 gpurun_out/r4_t13.log
timeout 300 python -m pytest tests/test_nets.py -x -q -m gpu -p no:cacheprovider -k "unet_forward or ddim or headline" > gpurun_out/r4_t13b.log 2>&1; tail -2 gpurun_out/r4_t13b.log
for B in 4 8 16; do
  timeout 200 python tests/gpu_probe.py --B $B --quick > gpurun_out/r4_h3c16_b$B.txt 2>&1
  echo "B=$B: $(grep -E 'ddim_graph' gpurun_out/r4_h3c16_b$B.txt | cut -c1-90) | $(grep -E '  conv_gemm  ' gpurun_out/r4_h3c16_b$B.txt | cut -c1-100)"
done
timeout 200 python tests/gpu_probe.py --B 4 > gpurun_out/r4_h3c16_b4_full.txt 2>&1
grep -E 'ddim_eager|ddim_graph|vae decode|wave encode|CFG' gpurun_out/r4_h3c16_b4_full.txt
