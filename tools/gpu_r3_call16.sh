#!/bin/bash
# Round 3, call 16 (short): VFE row statistics from the Y GEMM's epilogue ("vfe_fuse_stats") — C5 with / without, the VFE tests with
# it switched on for every ctx, and the C2 pair as a check that the shared GEMM kernel did not move.
# (Record of an experiment: the parameter was not kept — no gain, profiles/r3/c5_fuse_stats_{off,on}.log — so this script no longer runs as is.)
set -u
export TMPDIR=/tmp
O=gpurun_out/r3
mkdir -p $O
timeout 60 python tools/c5_profile.py reps=4 > $O/c5_fuse0.log 2>&1; grep '"config"' $O/c5_fuse0.log | cut -c1-400
timeout 60 python tools/c5_profile.py reps=4 vfe_fuse_stats=1 > $O/c5_fuse1.log 2>&1; grep '"config"' $O/c5_fuse1.log | cut -c1-400
GPMI_VFE_FUSE_STATS=1 timeout 150 python -m pytest tests/test_gpu_api.py tests/test_gpu_parity.py -x -q --timeout 100 -k "vfe or VFE or elbo or approx or dtc or sparse" > $O/pytest_call16.log 2>&1; echo "pytest rc=$?"; tail -6 $O/pytest_call16.log | cut -c1-400
timeout 60 python tools/bench_configs.py C2 2>&1 | grep '"config"' | cut -c1-300
