#!/bin/bash
# Round 3, GPU call 8 (final check): 220 fresh-context first fits in the shipped configuration (queues primed, self-check on) with
# stream-K and the high-priority comm stream ON in the rank contexts; the repaired tests; smoke and the bench line as the driver runs them.
export TMPDIR=/tmp
mkdir -p gpurun_out/r3
O=gpurun_out/r3
timeout 500 python tools/multi_fresh_stress.py 110 dims=1 check=0 grids=8x1,4x2 verify=1 sk=1 prio=1 > $O/ff7_shipped_220.log 2>&1; tail -1 $O/ff7_shipped_220.log
timeout 300 python -m pytest tests/test_gpu_api.py tests/test_gpu_multi.py -q --timeout 300 -k "append or self_check or sqmahal" > $O/pytest_call8.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_call8.log
python -c 'import __graft_entry__ as g; g.smoke()' 2>&1 | tail -2
timeout 600 python bench.py > $O/bench_final.json 2> $O/bench_final.err; echo "bench rc=$?"; python -c "
import json; d=json.loads(open('$O/bench_final.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic_detail']['source'], {k:round(v['ms_per_step'],2) for k,v in d['other_configs'].items()})"
