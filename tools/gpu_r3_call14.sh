#!/bin/bash
# Round 3, call 14 (short): third stand-alone probe — small pageable host-to-device uploads seen by the kernel behind them
set -u
O=gpurun_out/r3
mkdir -p $O
L=$O/hip_h2d_repro.log
: > $L
run() { echo "--- $*" >> $L; timeout 40 env "$@" >> $L 2>&1; echo "   rc=$?" >> $L; }
run GPU_MAX_HW_QUEUES=16 tools/bin/hip_h2d_repro 8 100 24 16384 0
run GPU_MAX_HW_QUEUES=16 tools/bin/hip_h2d_repro 8 100 24 16384 1
run GPU_MAX_HW_QUEUES=16 tools/bin/hip_h2d_repro 8 60 24 262144 1
run GPU_MAX_HW_QUEUES=16 tools/bin/hip_h2d_repro 12 60 24 4096 1
grep "uploads seen\|rc=\|held\|holds" $L | cut -c1-260 | head -30
