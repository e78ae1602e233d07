#!/bin/bash
# Round 3, call 13 (short): C.U' xi on the pieces + the whole multi-device test files after the late changes (event kinds, gather)
set -u
export TMPDIR=/tmp
O=gpurun_out/r3
mkdir -p $O
timeout 120 python -m pytest tests/test_gpu_multi.py -x -q --timeout 100 -k "sequential" > $O/pytest_call13a.log 2>&1; echo "pytest a rc=$?"; tail -12 $O/pytest_call13a.log | cut -c1-500
timeout 170 python -m pytest tests/test_gpu_multi.py tests/test_gpu_multi_rccl.py -x -q --timeout 150 -k "not sequential and not concurrent" > $O/pytest_call13b.log 2>&1; echo "pytest b rc=$?"; tail -8 $O/pytest_call13b.log | cut -c1-400
