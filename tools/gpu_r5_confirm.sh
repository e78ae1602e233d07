#!/bin/bash
# Confirmation call for a tree whose kernels and exact path are those of the last full artefact run (tools/gpu_r5_final.sh): the whole GPU suite as the
# driver runs it, smoke(), and the bench line with default flags (it replays profiles/r5/pmc_bench_summary.json for the counter fields).
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r5/confirm; rm -rf $OUT; mkdir -p $OUT; cd $R
t0=$(date +%s)
timeout 900 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu_final.log 2>&1; echo "pytest rc=$? ($(( $(date +%s) - t0 )) s)"; tail -3 $OUT/pytest_gpu_final.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke_final.log 2>&1; echo "smoke rc=$?"; tail -2 $OUT/smoke_final.log
timeout 600 python bench.py > $OUT/bench_c4_confirm.json 2> $OUT/bench_c4_confirm_progress.log; echo "bench rc=$? ($(( $(date +%s) - t0 )) s)"
python -c "
import json; d=json.loads(open('$OUT/bench_c4_confirm.json').read().strip().splitlines()[-1]); rf=d['roofline']; print(d['ms_per_step'], rf['frac'], rf['kernel_frac'], {k:(round(v['ms_per_step'],2), round(v.get('frac', v.get('frac_fp32')),3)) for k,v in d.get('other_configs',{}).items() if k != 'next'}, d['cpu_baseline']['value'], d.get('check_vs_oracle_digest')); nx=d['other_configs']['next']; print({k:(round(v['ms'],1), round(v['frac'],3)) for k,v in nx.items() if k!='value_and_gradient'}, {k:(round(v['ms'],1), round(v['frac'],3)) for k,v in nx['value_and_gradient'].items()})"
