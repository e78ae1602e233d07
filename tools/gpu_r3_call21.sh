#!/bin/bash
# Round 3, call 21 (short): the quick members of the full-size test file on the final tree
set -u
export TMPDIR=/tmp
O=gpurun_out/r3
mkdir -p $O
timeout 75 python -m pytest tests/test_gpu_fullsize.py -x -q --timeout 70 -k "c2 or fp32_vs_fp64 or committed" > $O/pytest_call21.log 2>&1; echo "rc=$?"; tail -5 $O/pytest_call21.log | cut -c1-300
