#!/bin/bash
# full GPU test-suite + default bench + mid-size / VFE timings with the defaults of the tree
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/check
rm -rf $OUT; mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest.log
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; cat $OUT/bench.json
timeout 600 python tools/bench_configs.py C2 C3 > $OUT/configs.jsonl 2>&1; tail -4 $OUT/configs.jsonl | cut -c1-400
timeout 300 python tools/sweep_r2.py C5only16 2>&1 | tail -2
