#!/bin/bash
# Round 4, call 2: the register-resident leaf — kernel-level check against the round-3 leaf, the unit / parity tests on it, and what it buys at C2 / C3 / small N.
set -u
export TMPDIR=/tmp
O=gpurun_out/r4
mkdir -p $O
timeout 120 tools/bin/leaf_check > $O/leaf_check.log 2>&1; echo "leaf_check rc=$?"; cat $O/leaf_check.log | cut -c1-260
timeout 300 python -m pytest tests/test_gpu_units.py tests/test_gpu_parity.py -x -q -m gpu --timeout 250 > $O/pytest_call2a.log 2>&1; echo "a rc=$?"; tail -5 $O/pytest_call2a.log | cut -c1-300
timeout 300 python tools/leaf_sweep_r4.py > $O/leaf_sweep.jsonl 2> $O/leaf_sweep.err; echo "sweep rc=$?"; cut -c1-330 $O/leaf_sweep.jsonl; tail -3 $O/leaf_sweep.err
