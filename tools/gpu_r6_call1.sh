#!/bin/bash
# round 6, call 1: the GPU suite on the first tree of the round (new full-size value tests of the next rows, fp32 inverse-block test) + the bench line
# with the measured cpu baseline and the checks on the timed next rows
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu -x --durations=15 > $O/pytest_call1.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_call1.log
tail -25 $O/pytest_call1.log
timeout 900 python bench.py --steps 10 --warmup 1 > $O/bench_call1.json 2> $O/bench_call1.err; echo "bench rc=$?"
tail -30 $O/bench_call1.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r6/bench_call1.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel_frac'])
print({k:(v['ms_per_step'], v.get('frac', v.get('frac_fp32'))) for k,v in d['other_configs'].items() if k!='next'})
for k,v in d['other_configs']['next'].items():
    if k=='value_and_gradient':
        for kk,vv in v.items(): print(kk, vv['ms'], vv['frac'], vv.get('check'))
    else: print(k, v['ms'], v['frac'], v.get('check'))
cb=d['cpu_baseline']; print(cb['value'], cb['sample'], cb.get('engine_vs_this_run'), cb['extrapolated']['pair_s'])
PY
