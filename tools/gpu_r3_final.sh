#!/bin/bash
# Round 3, final check of the tree as the driver will run it: the whole GPU suite, smoke.
export TMPDIR=/tmp
mkdir -p gpurun_out/r3
O=gpurun_out/r3
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest_gpu_final.log 2>&1; echo "pytest rc=$?"; tail -6 $O/pytest_gpu_final.log | cut -c1-300
python -c 'import __graft_entry__ as g; g.smoke()' 2>&1 | tail -1
