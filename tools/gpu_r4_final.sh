#!/bin/bash
# Round-4 end-of-round GPU call: the whole GPU suite as the driver runs it, smoke(), then the measurement artefacts (tools/gpu_r4_prof.sh).
set -u
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r4
cd $R
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r4/pytest_gpu_final.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r4/pytest_gpu_final.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r4/smoke_final.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/r4/smoke_final.log
bash tools/gpu_r4_prof.sh
