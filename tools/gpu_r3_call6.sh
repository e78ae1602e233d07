#!/bin/bash
# Round 3, GPU call 6: kernel-argument probe, residual first-fit rate with every queue created up front, the self-check, full GPU suite.
export TMPDIR=/tmp
mkdir -p gpurun_out/r3
O=gpurun_out/r3
{
GPU_MAX_HW_QUEUES=16 timeout 120 tools/bin/hip_kernarg_repro 8 150 48; echo "   rc=$?"
GPU_MAX_HW_QUEUES=16 HIP_FORCE_DEV_KERNARG=0 timeout 120 tools/bin/hip_kernarg_repro 8 150 48; echo "   rc=$?"
GPU_MAX_HW_QUEUES=16 timeout 120 tools/bin/hip_kernarg_repro 16 100 48; echo "   rc=$?"
} > $O/hip_kernarg_repro.log 2>&1
cat $O/hip_kernarg_repro.log | tail -20
S="timeout 300 python tools/multi_fresh_stress.py 60 dims=1 check=0 grids=8x1,4x2"
$S > $O/ff5_primed_raw.log 2>&1; tail -1 $O/ff5_primed_raw.log
$S verify=1 > $O/ff5_primed_verify.log 2>&1; tail -1 $O/ff5_primed_verify.log
GPMI_MULTI_PRIME=0 $S verify=1 > $O/ff5_unprimed_verify.log 2>&1; tail -1 $O/ff5_unprimed_verify.log
timeout 1500 python -m pytest tests -q -m gpu --timeout 600 > $O/pytest_gpu_full.log 2>&1; echo "pytest rc=$?"; tail -8 $O/pytest_gpu_full.log
