#!/bin/bash
# Round 3, call 11 (short): the second stale-argument probe (large by-value arguments, events, memops, fresh queues).
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3
mkdir -p $O
L=$O/hip_kernarg_repro2.log
: > $L
run() { echo "--- $*" >> $L; timeout 45 env "$@" >> $L 2>&1; echo "   rc=$?" >> $L; }
run GPU_MAX_HW_QUEUES=16 tools/bin/hip_kernarg_repro2 8 100 48 256 1 1 1
run GPU_MAX_HW_QUEUES=16 tools/bin/hip_kernarg_repro2 8 100 48 480 1 1 1
run GPU_MAX_HW_QUEUES=16 tools/bin/hip_kernarg_repro2 8 100 48 480 0 0 0
run GPU_MAX_HW_QUEUES=16 tools/bin/hip_kernarg_repro2 12 80 48 128 1 1 1
run GPU_MAX_HW_QUEUES=16 HIP_FORCE_DEV_KERNARG=0 tools/bin/hip_kernarg_repro2 8 100 48 480 1 1 1
run GPU_MAX_HW_QUEUES=32 tools/bin/hip_kernarg_repro2 8 100 48 480 1 1 1
grep -c STALE $L; grep "launches wrong\|rc=\|held\|holds" $L | cut -c1-240 | head -40
