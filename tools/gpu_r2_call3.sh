#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q --no-header -rA --tb=short -p no:cacheprovider --timeout 900 > gpurun_out/pytest_gpu.log 2>&1; echo "tests rc=$?"; grep -E "passed|failed|FAILED|ERROR" gpurun_out/pytest_gpu.log | tail -15
timeout 120 python bench.py --selftest --vranks 4 --grid 2x2 2>&1 | tail -2
for A in "" "--vranks 1" "--vranks 1 --nb 2048" "--vranks 2" "--vranks 4 --grid 2x2" "--vranks 4" "--vranks 8"; do
  echo "== bench $A"
  timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline $A 2>&1 | tail -1 | python -c "
import sys,json
l=sys.stdin.read().strip()
try:
    r=json.loads(l); print(round(r['ms_per_step'],1),'ms', r['config']['parallelism'], 'frac',round(r['roofline']['frac'],4),'kernel',round(r['roofline']['kernel_achieved'],1),'launches',r['roofline']['launches_per_step'],'resid',r.get('check_residual_max'), 'logpdf', r['logpdf'])
except Exception as e: print('PARSE FAIL', l[-400:])
"
done
