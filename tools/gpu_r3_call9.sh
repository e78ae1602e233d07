#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r3
O=gpurun_out/r3
timeout 900 python -m pytest tests/test_gpu_multi.py tests/test_gpu_multi_rccl.py -q --timeout 300 -k "not concurrent" > $O/pytest_call9.log 2>&1; echo "pytest rc=$?"; tail -25 $O/pytest_call9.log | cut -c1-300
