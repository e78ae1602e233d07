#!/bin/bash
# what the driver runs at round end (smoke, GPU tests, default bench) + the launcher-form check
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python __graft_entry__.py --smoke > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/smoke.log
timeout 1500 python -m pytest tests -m gpu -q --no-header -rA --tb=short -p no:cacheprovider --timeout 900 > gpurun_out/pytest_gpu.log 2>&1; echo "tests rc=$?"; grep -E "passed|failed|FAILED|ERROR" gpurun_out/pytest_gpu.log | tail -8
timeout 400 python bench.py > gpurun_out/bench.log 2>&1; echo "bench rc=$?"; tail -1 gpurun_out/bench.log | cut -c1-1200
