#!/bin/bash
# Round 3, call 19 (short): the multi-device test files on the tree after the variant removal
set -u
export TMPDIR=/tmp
O=gpurun_out/r3
mkdir -p $O
timeout 100 python -m pytest tests/test_gpu_multi.py tests/test_gpu_multi_rccl.py -x -q --timeout 90 -k "not concurrent" > $O/pytest_call19.log 2>&1; echo "rc=$?"; tail -5 $O/pytest_call19.log | cut -c1-300
