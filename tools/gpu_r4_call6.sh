#!/bin/bash
# Round 4, call 6: the 128-column register-resident leaf in the engine: unit / parity tests, then the sweep (64- vs 128-column leaves)
set -u
export TMPDIR=/tmp
O=gpurun_out/r4
mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_units.py tests/test_gpu_parity.py tests/test_gpu_api.py -x -q -m gpu --timeout 250 > $O/pytest_call6a.log 2>&1; echo "a rc=$?"; tail -3 $O/pytest_call6a.log | cut -c1-300
timeout 400 python tools/leaf_sweep_r4.py > $O/leaf_sweep3.jsonl 2> $O/leaf_sweep3.err; echo "sweep rc=$?"; tail -2 $O/leaf_sweep3.err; python - <<'PY'
import json
for l in open("gpurun_out/r4/leaf_sweep3.jsonl"):
    r = json.loads(l); print(r["case"], r["setting"], "%.2f ms" % r["ms_min"], "potrf %.2f" % r["potrf_ms"], "%.1f TF" % r["tflops"], "lp %.1e a %.1e" % (r["logpdf_rel_vs_v1"], r["alpha_rel_vs_v1"]))
PY
