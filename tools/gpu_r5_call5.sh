#!/bin/bash
# Round 5, GPU call 5: Gram assembly on the compact lower grid (kmat_compact) A/B, kernel table of value + gradient at C2 / C4.
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r5; mkdir -p $OUT; cd $R
t0=$(date +%s)
timeout 300 python tools/r5_sweep.py pair:16384,65536 -- base kmat_compact=1 base kmat_compact=1 "kmat_compact=1,kmat_nt=1" > $OUT/sweep5.jsonl 2> $OUT/sweep5.err; echo "sweep rc=$? ($(( $(date +%s) - t0 )) s)"; python - <<PY
import json
for ln in open("$OUT/sweep5.jsonl"):
    d=json.loads(ln)
    if "case" in d: print(d["case"], d["setting"], round(d["ms_med"],2), "assemble", round(d["assemble_ms"],3), "logpdf", d["logpdf"])
PY
GPMI_PARAMS=kmat_compact=1 GPMI_TEST_EXPECT=kmat_compact=1 timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_random.py -x -q -m gpu > $OUT/pytest_call5_compact.log 2>&1; echo "pytest compact rc=$?"; tail -3 $OUT/pytest_call5_compact.log
cd /tmp
for n in 16384 65536; do
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/st_g$n -o t -- python $R/tools/r5_next.py $n grad -- base > $OUT/gradstats_$n.log 2>&1; echo "gradstats $n rc=$?"
  f=$(find $OUT/st_g$n -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/grad${n}_kernel_stats.csv && head -12 $OUT/grad${n}_kernel_stats.csv | cut -c1-170
  rm -rf $OUT/st_g$n
done
echo "all done ($(( $(date +%s) - t0 )) s)"
