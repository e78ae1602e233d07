#!/bin/bash
# Counter evidence of a round (ROUND=r6 bash tools/gpu_pmc_configs.sh): per-kernel SQ MFMA-busy / wait / LDS-stall counters and the effective clock
# (GRBM_GUI_ACTIVE / kernel wall time: MI355X_MICROARCH.md "DVFS give-back") for C5 (fp32 VFE GEMMs), C2 and C3 (leaf, in-panel update,
# both fp64 GEMM kernels) and a pure-MFMA reference kernel in the same pass; separate --pmc passes, kernel-trace only.
#   -> gpurun_out/${ROUND:-r6}/pmc/pmc_sq_summary.json (tools/pmc_summary2.py)
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${ROUND:-r6}/pmc
rm -rf $OUT; mkdir -p $OUT
cd /tmp
i=0
for g in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $g -d $OUT/pmc_c5_g$i -o pmc --output-format csv -- python $R/tools/c5_profile.py reps=2 mfma_ref=1 > $OUT/c5_g$i.log 2>&1; echo "c5 g$i rc=$?"
  timeout 300 rocprofv3 --kernel-trace --pmc $g -d $OUT/pmc_c2_g$i -o pmc --output-format csv -- python $R/tools/trace_fit.py 16384 mfma_ref=1 > $OUT/c2_g$i.log 2>&1; echo "c2 g$i rc=$?"
  timeout 300 rocprofv3 --kernel-trace --pmc $g -d $OUT/pmc_c3_g$i -o pmc --output-format csv -- python $R/tools/trace_fit.py 32768 mfma_ref=1 > $OUT/c3_g$i.log 2>&1; echo "c3 g$i rc=$?"
done
python $R/tools/pmc_summary2.py $OUT > $OUT/pmc_sq_summary.json 2> $OUT/pmc_sq_summary.err
python $R/tools/pmc_clock.py $OUT/pmc_sq_summary.json | tee $OUT/pmc_sq_table.txt
find $OUT -name "*.csv" -size +1M -delete; find $OUT -name "*.db" -delete; du -sh $OUT
