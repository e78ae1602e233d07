"""C5 (VFE, N = 262 144, M = 4 096, fp32) A/B inside ONE process: the chunk GEMMs on one / two streams ("vfe_dual") × the prelude's inv(L_z) by 64-wide leaves /
batched inverse diagonal blocks ("vfe_inv_nb"), alternated `rounds` times; per configuration one warm-up fit, then the median of 5, with the phase split
(prelude / streamed pass / M×M side after it) of the last fit and the ELBO (must agree between configurations to fp32 accumulation noise).
    python tools/c5_ab.py [rounds=3] [extra NAME=VALUE ctx parameters]"""
import json
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import abstractgps_jl_amd as agp  # noqa: E402

rounds, extra = 3, {}
for a in sys.argv[1:]:
    k, v = a.split("=")
    if k == "rounds":
        rounds = int(v)
    else:
        extra[k] = int(v)
n, m, d = 262144, 4096, 3
rng = np.random.default_rng(5)
X = (rng.uniform(0, 1, (n, d)) * 4).astype(np.float32)
y = (np.sin(X.sum(1)) + 0.3 * rng.standard_normal(n)).astype(np.float32)
z = X[rng.permutation(n)[:m]].copy()
ctx = agp.default_context()
for k, v in extra.items():
    ctx.set_param(k, v)
f = agp.GP(agp.SqExponentialKernel())
fx = f(agp.RowVecs(X), np.float32(0.1))
approx = agp.VFE(f(agp.RowVecs(z), 1e-4))
flops = 2.0 * n * m * m + 2.0 * m**3 / 3
for rnd in range(rounds):
    for dual, inv in ((0, 0), (1, 0), (0, 512), (1, 512)):
        ctx.set_param("vfe_dual", dual)
        ctx.set_param("vfe_inv_nb", inv)
        ts = []
        for i in range(6):
            t0 = time.perf_counter()
            post = agp.posterior(approx, fx, y)
            dt = (time.perf_counter() - t0) * 1e3
            if i:
                ts.append(dt)
            tm = ctx.timings()
            elbo = float(post.objective)
            del post
        med = float(np.median(ts))
        print(json.dumps({"round": rnd, "vfe_dual": dual, "vfe_inv_nb": inv, "ms_median": round(med, 3), "ms_min": round(min(ts), 3),
                          "frac_fp32": flops / (med * 1e-3) / 157.3e12, "prelude_ms": round(tm["assemble_ms"], 3), "stream_ms": round(tm["potrf_ms"], 3),
                          "mxm_side_ms": round(tm["solve_ms"], 3), "elbo": elbo}), flush=True)
ctx.set_param("vfe_dual", 0)
ctx.set_param("vfe_inv_nb", 512)
