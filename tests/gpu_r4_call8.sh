#!/bin/bash
# round 4, call 8: where do the conv_gemm K-loop cycles go?  stall / issue counters on one long-K launch shape (hot weights)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp
rocprofv3 -L 2>/dev/null | grep -oE "\bSQ_[A-Z0-9_]+" | sort -u > $GRAFT_REPO_ROOT/gpurun_out/r4_sq_counters.txt
wc -l $GRAFT_REPO_ROOT/gpurun_out/r4_sq_counters.txt
rm -rf /tmp/pmc8
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_VALU_MFMA_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_VMEM_RD" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_IFETCH"; do
  i=$((i+1))
  timeout 120 rocprofv3 --pmc $set --output-format csv -d /tmp/pmc8/p$i -- python $GRAFT_REPO_ROOT/tests/gpu_convbench.py --pmc --shape 3 --tn 32 --wk 8 > $GRAFT_REPO_ROOT/gpurun_out/r4_pmc8_$i.log 2>&1
  tail -1 $GRAFT_REPO_ROOT/gpurun_out/r4_pmc8_$i.log
done
python $GRAFT_REPO_ROOT/tests/pmc_raw_summary.py /tmp/pmc8 conv_gemm > $GRAFT_REPO_ROOT/gpurun_out/r4_pmc8_summary.txt
cat $GRAFT_REPO_ROOT/gpurun_out/r4_pmc8_summary.txt
