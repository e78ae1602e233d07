#!/bin/bash
# Round 4, call 4: the whole GPU suite on the register-resident leaf (timings per test file for the 1 200 s budget)
set -u
export TMPDIR=/tmp
O=gpurun_out/r4
mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu --timeout 600 --durations=25 > $O/pytest_gpu_call4.log 2>&1; echo "rc=$?"; tail -45 $O/pytest_gpu_call4.log | cut -c1-200
