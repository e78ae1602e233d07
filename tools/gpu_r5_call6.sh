#!/bin/bash
# Round 5, GPU call 6: the row-by-row Gram kernel (one instance per dimension bucket, few registers): parity files + timings (assemble phase, pairs, C5).
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r5; mkdir -p $OUT; cd $R
t0=$(date +%s)
timeout 900 python -m pytest tests/test_gpu_units.py tests/test_gpu_parity.py tests/test_gpu_random.py tests/test_gpu_api.py -x -q -m gpu > $OUT/pytest_call6.log 2>&1; echo "pytest rc=$? ($(( $(date +%s) - t0 )) s)"; tail -4 $OUT/pytest_call6.log
timeout 300 python tools/r5_sweep.py pair:4096,16384,32768,65536 c5 -- base base > $OUT/sweep6.jsonl 2> $OUT/sweep6.err; echo "sweep rc=$?"; python - <<PY
import json
for ln in open("$OUT/sweep6.jsonl"):
    d=json.loads(ln)
    if "case" in d: print(d["case"], round(d["ms_med"],3), "assemble", d.get("assemble_ms"), d.get("frac", d.get("frac_fp32")))
PY
echo "all done ($(( $(date +%s) - t0 )) s)"
