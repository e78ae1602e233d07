#!/bin/bash
# Round 3, GPU call 3: stand-alone event-order probe, selective host-side waits in the multi-device driver, the new vector solves.
export TMPDIR=/tmp
mkdir -p gpurun_out/r3
O=gpurun_out/r3
R=tools/bin/hip_event_repro
{
for cfg in "16 8 400 32768 0" "16 8 400 32768 1" "32 8 400 32768 0" "4 8 400 32768 0" "16 12 300 32768 0" "16 4 400 32768 0" "16 8 1500 4096 0" "16 8 200 262144 0"; do
  set -- $cfg
  GPU_MAX_HW_QUEUES=$1 timeout 120 $R $2 $3 $4 $5; echo "   rc=$?"
done
} > $O/hip_event_repro.log 2>&1
cat $O/hip_event_repro.log
S="timeout 300 python tools/multi_fresh_stress.py"
$S 10 dims=1 check=0 dsync=32   > $O/ff2_dsync32_arrived.log 2>&1; tail -1 $O/ff2_dsync32_arrived.log
$S 10 dims=1 check=0 dsync=64   > $O/ff2_dsync64_peer.log 2>&1; tail -1 $O/ff2_dsync64_peer.log
$S 10 dims=1 check=0 dsync=128  > $O/ff2_dsync128_reuse.log 2>&1; tail -1 $O/ff2_dsync128_reuse.log
$S 10 dims=1 check=0            > $O/ff2_check0.log 2>&1; tail -1 $O/ff2_check0.log
$S 10 dims=1 check=0 hwq=8      > $O/ff2_hwq8.log 2>&1; tail -1 $O/ff2_hwq8.log
$S 10 dims=1 check=0 sk=0 prio=0 > $O/ff2_sk0_prio0.log 2>&1; tail -1 $O/ff2_sk0_prio0.log
grep -h WRONG $O/ff2_*.log | cut -c1-330 | head -20
timeout 600 python -m pytest tests/test_gpu_units.py tests/test_gpu_parity.py tests/test_gpu_api.py -q -x --timeout 300 > $O/pytest_call3.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest_call3.log
timeout 300 python tools/sweep_r3.py trsv C2 C4 > $O/sweep_trsv.jsonl 2> $O/sweep_trsv.err; echo "sweep trsv rc=$?"; cat $O/sweep_trsv.jsonl
