#!/bin/bash
# development tool: executed instructions per wave of single conv_gemm launch shapes (tests/gpu_convbench.py --pmc --shape I), in-tree library vs tests/var/$1
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
OLD=$1; shift
out=gpurun_out/shape_pmc.txt; : > $out
for shape in "$@"; do for v in new $OLD; do
  for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM" "SQ_WAVES SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_VMEM" "SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY"; do
    d=/tmp/spmc_${shape}_${v}_$(echo $set | tr ' ' '_'); rm -rf $d
    if [ $v = new ]; then (cd /tmp && timeout 120 rocprofv3 --pmc $set --output-format csv -d $d -- python $OLDPWD/tests/gpu_convbench.py --pmc --shape $shape) > /dev/null 2>&1
    else (cd /tmp && MUGD_LIB_PATH=$OLDPWD/tests/var/$v/libmugd.so timeout 120 rocprofv3 --pmc $set --output-format csv -d $d -- python $OLDPWD/tests/gpu_convbench.py --pmc --shape $shape) > /dev/null 2>&1; fi
    echo "== shape $shape lib $v" >> $out
    python tests/pmc_generic_summary.py $d 2>/dev/null | grep -E "^kernel|conv_gemm" >> $out
  done
done; done
cat $out
