"""C5 (VFE, N = 262 144, M = 4 096, fp32 streamed / fp64 M×M side) alone, for rocprofv3 --kernel-trace --stats: `reps` fits, the
phase split of the last one (assemble = K_zz + chol + inv(L_z) prelude; potrf = streamed pass + Λ_ε side) and optional parameter
overrides NAME=VALUE on the command line.    python tools/c5_profile.py [reps=3] [dtype=f32|f64] [vfe_chunk=...] [vfe_ks=...] [vfe_overlap=0|1]"""
import json
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import abstractgps_jl_amd as agp  # noqa: E402

reps, params, mfma_ref, dt = 3, {}, 0, np.float32
for a in sys.argv[1:]:
    k, v = a.split("=")
    if k == "reps":
        reps = int(v)
    elif k == "mfma_ref":
        mfma_ref = int(v)
    elif k == "dtype":
        dt = np.float64 if v in ("f64", "float64") else np.float32
    else:
        params[k] = int(v)
n, m, d = 262144, 4096, 3
rng = np.random.default_rng(5)
X = (rng.uniform(0, 1, (n, d)) * 4).astype(dt)
y = (np.sin(X.sum(1)) + 0.3 * rng.standard_normal(n)).astype(dt)
z = X[rng.permutation(n)[:m]].copy()
ctx = agp.default_context()
for k, v in params.items():
    ctx.set_param(k, v)
f = agp.GP(agp.SqExponentialKernel())
fx = f(agp.RowVecs(X), dt(0.1))
approx = agp.VFE(f(agp.RowVecs(z), 1e-4))
ts = []
for _ in range(reps):
    t0 = time.perf_counter()
    post = agp.posterior(approx, fx, y)
    ts.append((time.perf_counter() - t0) * 1e3)
    tm = ctx.timings()
    elbo = float(post.objective)
    del post
flops = 2.0 * n * m * m + 2.0 * m**3 / 3
if mfma_ref:  # the pure-MFMA reference kernel inside the same profiled process (normalises SQ_VALU_MFMA_BUSY_CYCLES / GRBM_GUI_ACTIVE)
    import ctypes as C

    v = C.c_double()
    ctx.lib.gp_bench_mfma_f32(ctx.handle, 0, 20000, C.byref(v))
    ctx.lib.gp_bench_mfma_f64(ctx.handle, 20000, C.byref(v))
print(json.dumps({"config": "C5", "params": params, "fit_ms": [round(t, 2) for t in ts], "best_ms": min(ts), "frac_fp32": flops / (min(ts) * 1e-3) / 157.3e12,
                  "phases_last": {k: (round(v, 3) if isinstance(v, float) else v) for k, v in tm.items()}, "elbo": elbo}), flush=True)
