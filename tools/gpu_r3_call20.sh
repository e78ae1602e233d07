#!/bin/bash
# Round 3, call 20 (short): smoke() and a short bench line (C4) on the final tree
set -u
export TMPDIR=/tmp
O=gpurun_out/r3
mkdir -p $O
timeout 60 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke_final2.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke_final2.log | cut -c1-200
timeout 80 python bench.py --steps 4 --warmup 1 --no-cpu-baseline > $O/bench_final2.json 2> $O/bench_final2.err; echo "bench rc=$?"
python -c "
import json; d=json.loads(open('$O/bench_final2.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel_frac'], {k:(round(v['ms_per_step'],2)) for k,v in d.get('other_configs',{}).items()})"
