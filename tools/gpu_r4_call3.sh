#!/bin/bash
# Round 4, call 3: the settled register-resident leaf (early pivot, MFMA accumulators in VGPRs): kernel check, unit/parity tests, sweep
set -u
export TMPDIR=/tmp
O=gpurun_out/r4
mkdir -p $O
timeout 60 tools/bin/leaf_stamps 16384 0 128 | tail -3; timeout 60 tools/bin/leaf_stamps 16384 0 64 | tail -3; timeout 60 tools/bin/leaf_stamps 16384 1 64 | tail -3
timeout 120 tools/bin/leaf_check > $O/leaf_check.log 2>&1; echo "leaf_check rc=$?"; tail -4 $O/leaf_check.log | cut -c1-230
timeout 300 python -m pytest tests/test_gpu_units.py tests/test_gpu_parity.py -x -q -m gpu --timeout 250 > $O/pytest_call3a.log 2>&1; echo "a rc=$?"; tail -3 $O/pytest_call3a.log | cut -c1-300
timeout 300 python tools/leaf_sweep_r4.py > $O/leaf_sweep2.jsonl 2> $O/leaf_sweep2.err; echo "sweep rc=$?"; python - <<'PY'
import json
for l in open("gpurun_out/r4/leaf_sweep2.jsonl"):
    r = json.loads(l); print(r["case"], r["setting"], "%.2f ms" % r["ms_min"], "potrf %.2f" % r["potrf_ms"], "%.1f TF" % r["tflops"], "lp %.1e a %.1e" % (r["logpdf_rel_vs_v1"], r["alpha_rel_vs_v1"]))
PY
