#!/bin/bash
# Round 3, call 22 (short): C3 value parity and the C4 full-size test on the final tree
set -u
export TMPDIR=/tmp
O=gpurun_out/r3
mkdir -p $O
timeout 52 python -m pytest tests/test_gpu_fullsize.py -x -q --timeout 50 -k "c3_full_size_values or c4_full_size" > $O/pytest_call22.log 2>&1; echo "rc=$?"; tail -5 $O/pytest_call22.log | cut -c1-300
