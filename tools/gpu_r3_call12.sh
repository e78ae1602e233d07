#!/bin/bash
# Round 3, call 12 (short): the passes over the distributed factor added late in the round (C \ B, sequential update on the pieces)
set -u
export TMPDIR=/tmp
O=gpurun_out/r3
mkdir -p $O
timeout 100 python tools/multi_update_diag.py 1x1 2x2 2x3 > $O/multi_update_diag.log 2>&1; echo "diag rc=$?"; grep "grid\|Error\|error" $O/multi_update_diag.log | cut -c1-900
timeout 170 python -m pytest tests/test_gpu_multi.py -x -q --timeout 120 -k "sequential or predictive_variance_on_the_distributed_factor" > $O/pytest_call12.log 2>&1; echo "pytest rc=$?"; tail -15 $O/pytest_call12.log | cut -c1-400
