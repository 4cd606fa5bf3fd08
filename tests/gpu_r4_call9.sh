#!/bin/bash
# round 4, call 9: conv_gemm K loop, cold vs hot weights: memory / LDS latency (level counters), MFMA+VALU co-execution
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp
for mode in cold hot; do
rm -rf /tmp/pmc9
flag=""; [ $mode = hot ] && flag="--hot"
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INST_LEVEL_SMEM SQ_INSTS_SMEM SQ_IFETCH SQ_IFETCH_LEVEL"; do
  i=$((i+1))
  timeout 120 rocprofv3 --pmc $set --output-format csv -d /tmp/pmc9/p$i -- python $GRAFT_REPO_ROOT/tests/gpu_convbench.py --pmc --shape 3 --tn 32 --wk 8 $flag > $GRAFT_REPO_ROOT/gpurun_out/r4_pmc9_${mode}_$i.log 2>&1
  grep "res.l1" $GRAFT_REPO_ROOT/gpurun_out/r4_pmc9_${mode}_$i.log
done
python $GRAFT_REPO_ROOT/tests/pmc_raw_summary.py /tmp/pmc9 conv_gemm > $GRAFT_REPO_ROOT/gpurun_out/r4_pmc9_${mode}_summary.txt
echo "== $mode"; cat $GRAFT_REPO_ROOT/gpurun_out/r4_pmc9_${mode}_summary.txt
done
