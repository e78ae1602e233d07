#!/bin/bash
# Final validation of the tree + the artefacts committed under profiles/r2: what the driver runs at round end (smoke, GPU tests,
# default bench), rocprofv3 kernel stats of the bench and of C2 / C3 / C5, full-size parity (C4, C5) against the CPU oracle on this
# box's host cores, the seeded random sweep, and the in-library multi-device driver with virtual ranks.
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/final
rm -rf $OUT; mkdir -p $OUT
cd $R
timeout 600 python __graft_entry__.py --smoke > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log | cut -c1-300
timeout 1500 python -m pytest tests -m gpu -q --no-header -rA --tb=short -p no:cacheprovider --timeout 900 > $OUT/pytest_gpu.log 2>&1; echo "tests rc=$?"; grep -E "passed|failed" $OUT/pytest_gpu.log | tail -3
timeout 400 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; cut -c1-700 $OUT/bench.json
cd /tmp
stats() { tag=$1; shift
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$tag -o p -- "$@" > $OUT/$tag.log 2>&1
  echo "stats $tag rc=$?"
  f=$(find $OUT/$tag -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $OUT/${tag}_kernel_stats.csv && head -4 $f | cut -c1-160
  rm -rf $OUT/$tag
}
stats bench_c4 python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-check
stats C2 python $R/tools/bench_configs.py C2
stats C3 python $R/tools/bench_configs.py C3
stats C5 python $R/tools/sweep_r2.py C5only16
cd $R
timeout 900 python tools/fullsize_parity.py --configs C4,C5 --out $OUT/fullsize_parity.jsonl > $OUT/fullsize.log 2>&1; echo "fullsize rc=$?"; cut -c1-420 $OUT/fullsize_parity.jsonl
GPU_MAX_HW_QUEUES=16 timeout 900 python tools/random_sweep2.py 150 > $OUT/random_sweep2.log 2>&1; echo "sweep rc=$?"; tail -3 $OUT/random_sweep2.log
for A in "--vranks 1" "--vranks 1 --nb 2048" "--vranks 2" "--vranks 4 --grid 2x2" "--vranks 8"; do
  timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline $A 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$A', round(d['ms_per_step'], 1), 'ms frac', round(d['roofline']['frac'], 4), d['config'].get('parallelism'), 'logpdf', d.get('logpdf'))"
done 2>&1 | tee $OUT/multi_virtual.txt
du -sh $OUT
