#!/bin/bash
# quick check of an engine change: kernel unit tests + small/mid timings
set -u
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_units.py -x -q 2>&1 | tail -3
timeout 300 python tools/trace_fit.py 4096 2>&1 | tail -1
timeout 300 python tools/trace_fit.py 16384 2>&1 | tail -1
timeout 300 python tools/trace_fit.py 32768 2>&1 | tail -1
timeout 300 python tools/sweep_r2.py C5only16 2>&1 | tail -1 | cut -c1-200
