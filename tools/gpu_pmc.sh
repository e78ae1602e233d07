#!/bin/bash
# PMC passes (one counter group per run, --kernel-trace only — never combined with other trace domains) over the
# isolated trailing-update kernel (tools/gemm_bench.py).  Output: gpurun_out/pmc/<tag>/..., counters list in
# gpurun_out/pmc/counters.txt.
set -u
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc
mkdir -p $OUT
cd /tmp
timeout 120 rocprofv3 -L > $OUT/counters.txt 2>&1
ARGS="${PMC_ARGS:---shapes 32768x32768x2048 --reps 2 --lower 1}"
CMD="${PMC_CMD:-python $GRAFT_REPO_ROOT/tools/gemm_bench.py $ARGS}"
i=0
while IFS= read -r grp; do
  [ -z "$grp" ] && continue
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $grp -d $OUT/p$i -o pmc --output-format csv -- $CMD > $OUT/p$i.log 2>&1
  echo "pass $i [$grp] rc=$?"
done <<< "${PMC_GROUPS:-SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE
SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F64
FETCH_SIZE
WRITE_SIZE
SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS
TCC_HIT_sum TCC_MISS_sum}"
find $OUT -name "*.csv" | head -40
