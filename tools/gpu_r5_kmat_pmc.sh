#!/bin/bash
# Round 5: counters of kmat_kernel<double> (the Gram phase sits at 4.6 TB/s while its store pattern alone reaches 5.7): where do its wave cycles go?
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r5/kmat; rm -rf $OUT; mkdir -p $OUT; cd /tmp
i=0
for g in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INST_CYCLES_VMEM_WR GRBM_GUI_ACTIVE" "SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_CVT SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES" "TCC_EA0_WRREQ TCC_EA0_WRREQ_64B TCC_EA0_WRREQ_STALL"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $g -d $OUT/p$i -o pmc --output-format csv -- python $R/tools/trace_fit.py 32768 > $OUT/p$i.log 2>&1; echo "pass $i rc=$?"
done
python - <<PY
import csv, glob, collections
for i in (1,2,3):
    cc=glob.glob("$OUT/p%d/**/*counter_collection.csv"%i, recursive=True); kt=glob.glob("$OUT/p%d/**/*kernel_trace.csv"%i, recursive=True)
    dur={}
    for f in kt:
        for r in csv.DictReader(open(f)): dur[r["Dispatch_Id"]]=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3
    agg=collections.defaultdict(float); ids=set()
    for f in cc:
        for r in csv.DictReader(open(f)):
            if "kmat_kernel<double" in r["Kernel_Name"]:
                agg[r["Counter_Name"]]+=float(r["Counter_Value"]); ids.add(r["Dispatch_Id"])
    us=sum(dur.get(d,0) for d in ids)
    print("pass",i,"kmat dispatches",len(ids),"avg us",us/max(len(ids),1), dict(agg))
PY
find $OUT -name "*.csv" -size +1M -delete; find $OUT -name "*.db" -delete
