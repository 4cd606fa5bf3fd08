#!/bin/bash
# round-3 GPU call 1: XCD-local barrier price, shipped training step vs the reference fixture, graph vs eager kernel gaps
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 120 tests/bin/xcd_barrier > gpurun_out/r3_xcd_barrier.txt 2>&1; echo "xcd_barrier rc $?" >> gpurun_out/r3_xcd_barrier.txt
timeout 900 python -m pytest tests/test_train.py -x -q -m gpu -k "shipped_training_step_vs_reference" -s > gpurun_out/r3_train_parity_fp32.log 2>&1; echo "rc $?" >> gpurun_out/r3_train_parity_fp32.log
for mode in graph eager graph_all; do
  rm -rf /tmp/gve_$mode
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/gve_$mode -- python $GRAFT_REPO_ROOT/tests/gpu_graph_vs_eager.py --mode $mode) > gpurun_out/r3_gve_$mode.log 2>&1
  f=$(find /tmp/gve_$mode -name "*kernel_trace.csv" | head -1)
  python tests/pp_kernel_gaps.py "$f" $mode >> gpurun_out/r3_graph_vs_eager.txt 2>&1
  grep "ddim_" gpurun_out/r3_gve_$mode.log >> gpurun_out/r3_graph_vs_eager.txt
done
# un-profiled A/B of the three modes in one process order (graph, eager, graph_all, graph)
for mode in graph eager graph_all graph eager; do timeout 200 python tests/gpu_graph_vs_eager.py --mode $mode 2>&1 | grep ddim_ >> gpurun_out/r3_graph_vs_eager.txt; done
tail -5 gpurun_out/r3_xcd_barrier.txt; tail -30 gpurun_out/r3_graph_vs_eager.txt; tail -5 gpurun_out/r3_train_parity_fp32.log
