#!/bin/bash
# HBM traffic of the bench's dominant kernel: separate --pmc passes (FETCH_SIZE, WRITE_SIZE) over `bench.py --steps 1`,
# kernel-trace only.  Output under gpurun_out/pmc_bench/.
set -u
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_bench
mkdir -p $OUT
cd /tmp
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES"; do
  i=$((i+1))
  timeout 900 rocprofv3 --kernel-trace --pmc $grp -d $OUT/p$i -o pmc --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-check > $OUT/p$i.log 2>&1
  echo "pass $i [$grp] rc=$?"
done
