#!/bin/bash
# Round 3, call 18 (short): the API and random-sweep test files on the tree after the variant removal
set -u
export TMPDIR=/tmp
O=gpurun_out/r3
mkdir -p $O
timeout 95 python -m pytest tests/test_gpu_api.py tests/test_gpu_random.py -x -q --timeout 80 > $O/pytest_call18.log 2>&1; echo "rc=$?"; tail -5 $O/pytest_call18.log | cut -c1-300
