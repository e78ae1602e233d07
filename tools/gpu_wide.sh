#!/bin/bash
set -u
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_units.py -m gpu -q --no-header -k "wide" --tb=short -p no:cacheprovider 2>&1 | tail -15
for P in "gemm_wide=0" "gemm_wide=2"; do
  timeout 300 python tools/gemm_bench.py --params $P --shapes 16384x16384x2048,32768x32768x2048,8192x8192x1024,4096x4096x2048 --lower 1 --reps 4 2>&1 | grep -v amdgpu | grep '"m"'
done
python - <<'PY'
import sys; sys.path.insert(0,'.')
from tools.sweep_r2 import vfe, exact
import abstractgps_jl_amd as agp
for w in (0, 1, 2):
    vfe({"vfe_chunk": 16384, "gemm_wide": w}, reps=2)
for w in (0, 1):
    exact("C3", 32768, 8, 3, agp.Matern32Kernel() @ agp.ScaleTransform(0.5), {"gemm_wide": w, "gemm_wide_min": 256}, reps=2)
    exact("C2", 16384, 3, 2, agp.SqExponentialKernel(), {"gemm_wide": w, "gemm_wide_min": 256}, reps=3)
PY
