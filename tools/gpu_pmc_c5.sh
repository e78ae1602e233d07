#!/bin/bash
# SQ wait / LDS counters of the fp32 VFE GEMMs (C5) and of the fp64 trailing update (C3), separate passes
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_sq
rm -rf $OUT; mkdir -p $OUT
cd /tmp
i=0
for g in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $g -d $OUT/pmc_c5_g$i -o pmc --output-format csv -- python $R/tools/sweep_r2.py C5only16 > $OUT/c5_g$i.log 2>&1; echo "c5 g$i rc=$?"
  timeout 300 rocprofv3 --kernel-trace --pmc $g -d $OUT/pmc_c3_g$i -o pmc --output-format csv -- python $R/tools/trace_fit.py 32768 > $OUT/c3_g$i.log 2>&1; echo "c3 g$i rc=$?"
done
python $R/tools/pmc_summary2.py $OUT > $OUT/pmc_sq_summary.json 2> $OUT/pmc_sq_summary.err
python - <<PY
import json
d=json.load(open("$OUT/pmc_sq_summary.json"))
for tag,v in d.items():
    for k,c in list(v.items())[:3]:
        print(tag, k[:36], c['dispatches'], round(c['total_us']/1e3,1),'ms', {n:float('%.4g'%x['sum']) for n,x in c.items() if isinstance(x,dict)})
PY
tail -3 $OUT/c5_g3.log
find $OUT -name "*.csv" -size +1M -delete; find $OUT -name "*.db" -delete; du -sh $OUT
