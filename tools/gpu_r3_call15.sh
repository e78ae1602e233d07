#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r3
mkdir -p $O
timeout 150 python -m pytest tests/test_gpu_api.py -x -q --timeout 100 -k "conformance or sqmahal" > $O/pytest_call15.log 2>&1; echo "pytest rc=$?"; tail -30 $O/pytest_call15.log | cut -c1-600
