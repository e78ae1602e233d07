#!/bin/bash
# Round 2, GPU call 1: smoke, full GPU test-suite, full-size value parity (oracle on the host cores), bench, sweeps.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/fullsize_parity.jsonl
timeout 600 python __graft_entry__.py --smoke > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -3 gpurun_out/smoke.log
timeout 1000 python -m pytest tests -m gpu -q --no-header -rA --tb=short -p no:cacheprovider --timeout 400 > gpurun_out/pytest_gpu.log 2>&1; echo "tests rc=$?"; tail -15 gpurun_out/pytest_gpu.log
timeout 700 python tools/fullsize_parity.py > gpurun_out/fullsize_parity.log 2>&1; echo "fullsize rc=$?"; cut -c1-600 gpurun_out/fullsize_parity.log | tail -8
timeout 300 python bench.py --steps 3 --warmup 1 > gpurun_out/bench.log 2>&1; echo "bench rc=$?"; tail -2 gpurun_out/bench.log | cut -c1-1500
timeout 420 python tools/sweep_r2.py > gpurun_out/sweep_r2.log 2>&1; echo "sweep rc=$?"; cat gpurun_out/sweep_r2.log
nproc; free -g | head -2
