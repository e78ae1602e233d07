#!/bin/bash
# launcher form of the bench with virtual ranks (rank 0 drives, rank 1 idles at the barriers), RCCL API check on one device,
# and the "fewer GPUs than requested" exit
set -u
export TMPDIR=/tmp
echo "== torchrun 2 ranks, 2 virtual ranks on GPU 0, N=16384"
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --vranks 2 --npoints 16384 --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | grep -v amdgpu.ids | tail -3 | cut -c1-700
echo "== --gpus 2 on a 1-GPU box must fail loudly"
timeout 120 python bench.py --gpus 2 --n 8192 --steps 1 --warmup 0 --no-cpu-baseline 2>&1 | tail -1; echo "rc=${PIPESTATUS[0]}"
echo "== RCCL API check: ncclCommInitAll over one device"
GPMI_COMM=rccl timeout 200 python - <<'PY' 2>&1 | grep -v amdgpu.ids | tail -12
import sys; sys.path.insert(0, '.')
import numpy as np
import abstractgps_jl_amd as agp
from oracle import gp_oracle as o
x, y = o.synth_inputs(3000, 3, 1)
ctx = agp.Context(devices=[0], nb=256)
print(ctx.multi_info())
f = agp.GP(agp.SqExponentialKernel(), ctx=ctx)
post = agp.posterior(f(agp.RowVecs(x), 0.01), y)
lp, op = o.logpdf_and_posterior(o.FiniteGP(o.GP(o.Kernel(o.SE)), x, 0.01), y)
print("logpdf rel", abs(float(post.logpdf_value) - lp) / abs(lp), "alpha rel", np.linalg.norm(post.data.alpha - op.alpha) / np.linalg.norm(op.alpha))
ctx.close()
PY
