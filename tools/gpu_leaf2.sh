#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/leaf2
rm -rf $OUT; mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest.log
timeout 600 python tools/bench_configs.py C2 C3 2>&1 | cut -c1-260
timeout 300 python tools/sweep_r2.py C5only16 2>&1 | tail -1
timeout 300 python tools/trace_fit.py 4096 2>&1 | tail -2
timeout 300 python tools/trace_fit.py 65536 2>&1 | tail -2
