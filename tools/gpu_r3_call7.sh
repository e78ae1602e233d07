#!/bin/bash
# Round 3, GPU call 7: the every-row self-check against lazily created queues (the 20 % regime), the corrected kernel-argument probe,
# the multi-device tests, then the measurement artefacts (tools/gpu_r3_prof.sh).
export TMPDIR=/tmp
mkdir -p gpurun_out/r3
O=gpurun_out/r3
S="timeout 300 python tools/multi_fresh_stress.py 60 dims=1 check=0 grids=8x1,4x2"
GPMI_MULTI_PRIME=0 $S verify=1 > $O/ff6_unprimed_verify_allrows.log 2>&1; tail -1 $O/ff6_unprimed_verify_allrows.log; grep -c WRONG $O/ff6_unprimed_verify_allrows.log
{
GPU_MAX_HW_QUEUES=16 timeout 100 tools/bin/hip_kernarg_repro 8 150 48; echo "   rc=$?"
GPU_MAX_HW_QUEUES=16 HIP_FORCE_DEV_KERNARG=0 timeout 100 tools/bin/hip_kernarg_repro 8 150 48; echo "   rc=$?"
} > $O/hip_kernarg_repro.log 2>&1
tail -6 $O/hip_kernarg_repro.log | cut -c1-250
timeout 600 python -m pytest tests/test_gpu_multi.py tests/test_gpu_multi_rccl.py tests/test_gpu_api.py -q --timeout 300 -k "not concurrent" > $O/pytest_call7.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest_call7.log
bash tools/gpu_r3_prof.sh
