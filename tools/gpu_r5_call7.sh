#!/bin/bash
# Round 5, GPU call 7: batched build of the inverse diagonal blocks — parity (api / parity files), first-call vs warm predictive call, value + gradient with / without the blocks.
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r5; mkdir -p $OUT; cd $R
t0=$(date +%s)
timeout 900 python -m pytest tests/test_gpu_api.py tests/test_gpu_parity.py tests/test_gpu_units.py -x -q -m gpu > $OUT/pytest_call7.log 2>&1; echo "pytest rc=$? ($(( $(date +%s) - t0 )) s)"; tail -4 $OUT/pytest_call7.log
python - <<'PY' > $OUT/dib_first_call.jsonl 2> $OUT/dib_first_call.err
import json, sys, time
import numpy as np
sys.path.insert(0, ".")
import abstractgps_jl_amd as agp
ctx = agp.default_context(0)
for n in (16384, 65536):
    rng = np.random.default_rng(2)
    x = rng.standard_normal((n, 3)); y = np.sin(x.sum(1)) + 0.1 * rng.standard_normal(n)
    xs = rng.standard_normal((4096, 3))
    fx = agp.GP(agp.SqExponentialKernel(), ctx=ctx)(agp.RowVecs(x), 0.01)
    for dib in (0, 2048):
        ctx.set_param("dib_nb", dib)
        post = agp.posterior(fx, y)
        post.mean_and_var(agp.RowVecs(xs)); post.data.C.free()          # allocations of this size are in the cache now
        post = agp.posterior(fx, y)
        ts = []
        for _ in range(3):
            t0 = time.perf_counter(); post.mean_and_var(agp.RowVecs(xs)); ts.append((time.perf_counter() - t0) * 1e3)
        print(json.dumps({"n": n, "dib_nb": dib, "mean_and_var_4096_ms_first_then_warm": [round(t, 2) for t in ts]}), flush=True)
        post.data.C.free()
    ctx.set_param("dib_nb", 2048)
    ctx.trim()
PY
echo "first-call rc=$?"; cat $OUT/dib_first_call.jsonl
timeout 300 python tools/r5_next.py 16384 grad -- dib_nb=0 dib_nb=2048 dib_nb=0 dib_nb=2048 > $OUT/next7_c2.jsonl 2> $OUT/next7_c2.err; echo "next c2 rc=$?"; cut -c1-150 $OUT/next7_c2.jsonl
timeout 400 python tools/r5_next.py 65536 grad -- dib_nb=0 dib_nb=2048 > $OUT/next7_c4.jsonl 2> $OUT/next7_c4.err; echo "next c4 rc=$? ($(( $(date +%s) - t0 )) s)"; cut -c1-150 $OUT/next7_c4.jsonl
echo "all done ($(( $(date +%s) - t0 )) s)"
