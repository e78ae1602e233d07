#!/bin/bash
# Round 3, call 17 (short): after the removal of the off-by-default kernel variants — the unit tests of the building blocks, the
# single-device parity file, and the C2 pair as a speed check
set -u
export TMPDIR=/tmp
O=gpurun_out/r3
mkdir -p $O
timeout 70 python -m pytest tests/test_gpu_units.py -x -q --timeout 60 > $O/pytest_call17a.log 2>&1; echo "units rc=$?"; tail -4 $O/pytest_call17a.log | cut -c1-300
timeout 80 python -m pytest tests/test_gpu_parity.py -x -q --timeout 60 > $O/pytest_call17b.log 2>&1; echo "parity rc=$?"; tail -4 $O/pytest_call17b.log | cut -c1-300
timeout 30 python tools/bench_configs.py C2 2>&1 | grep '"config"' | cut -c1-260
