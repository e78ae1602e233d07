#!/bin/bash
# Round 4, call 5: deterministic mode, the in-suite fresh-context stress, the bench line with digest check / kmat roofline / next rows / rocSOLVER comparator,
# XCD order on / off at C4, C5 and M = 4096 on the new leaf.
set -u
export TMPDIR=/tmp
O=gpurun_out/r4
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_api.py tests/test_gpu_multi.py -q -m gpu --timeout 800 -k "deterministic or fresh_context or truly_concurrent or self_check" --durations=5 > $O/pytest_call5a.log 2>&1; echo "a rc=$?"; tail -12 $O/pytest_call5a.log | cut -c1-300
timeout 600 python bench.py --steps 5 --warmup 1 > $O/bench_c4.json 2> $O/bench_c4.err; echo "bench rc=$?"; tail -2 $O/bench_c4.err | cut -c1-300; python - <<'PY'
import json
l = [x for x in open("gpurun_out/r4/bench_c4.json") if x.startswith("{")]
if l:
    r = json.loads(l[-1])
    print("value", r["value"], "ms", r["ms_per_step"], "frac", r["roofline"]["frac"], "kernel_frac", r["roofline"]["kernel_frac"])
    print("kmat", r["roofline"].get("kmat"))
    print({k: v for k, v in r.items() if k.startswith("check")})
    oc = r.get("other_configs", {})
    for k in ("C2", "C3", "C5"):
        if k in oc: print(k, {kk: oc[k][kk] for kk in oc[k] if kk in ("ms_per_step", "ms_min", "frac", "frac_fp32")})
    print("next", json.dumps(oc.get("next"))[:1500])
    print("comparator", r.get("comparator_rocsolver_dpotrf"))
PY
timeout 400 python tools/r4_misc.py > $O/misc.jsonl 2> $O/misc.err; echo "misc rc=$?"; cat $O/misc.jsonl | cut -c1-250; tail -2 $O/misc.err
