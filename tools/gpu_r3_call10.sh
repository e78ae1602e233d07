#!/bin/bash
# Round 3, call 10 (short): the second stale-argument probe (large by-value arguments, events, memops, fresh queues) under the failing
# regime's queue count, and the C5 kernel statistics + phase split of the final engine.
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3
mkdir -p $O
L=$O/hip_kernarg_repro2.log
: > $L
run() { echo "--- $*" >> $L; ( eval "timeout 45 $*" ) >> $L 2>&1; echo "   rc=$?" >> $L; }
run "GPU_MAX_HW_QUEUES=16 tools/bin/hip_kernarg_repro2 8 100 48 256 1 1 1"
run "GPU_MAX_HW_QUEUES=16 tools/bin/hip_kernarg_repro2 8 100 48 480 1 1 1"
run "GPU_MAX_HW_QUEUES=16 tools/bin/hip_kernarg_repro2 8 100 48 480 0 0 0"
run "GPU_MAX_HW_QUEUES=16 tools/bin/hip_kernarg_repro2 12 80 48 128 1 1 1"
run "GPU_MAX_HW_QUEUES=16 HIP_FORCE_DEV_KERNARG=0 tools/bin/hip_kernarg_repro2 8 100 48 480 1 1 1"
run "GPU_MAX_HW_QUEUES=32 tools/bin/hip_kernarg_repro2 8 100 48 480 1 1 1"
grep -c STALE $L; grep "launches wrong\|rc=" $L | cut -c1-220
cd /tmp
timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c5 -o s -- python $R/tools/c5_profile.py reps=3 > $O/c5_profile.log 2>&1; echo "c5 rc=$?"
f=$(find $O/stats_c5 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/C5_kernel_stats.csv && head -12 $O/C5_kernel_stats.csv | cut -c1-160
rm -rf $O/stats_c5
grep '"config"' $O/c5_profile.log
