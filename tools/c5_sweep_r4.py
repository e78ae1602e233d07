"""Round 4: C5 (VFE fp32, N = 262 144, M = 4 096) on the final engine: default, stream-K on the M×M side, chunk size — ms per posterior + ELBO (median of 5)."""
import json
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import abstractgps_jl_amd as agp  # noqa: E402

ctx = agp.default_context(0)
rng = np.random.default_rng(5)
n, m, d = 262144, 4096, 3
X = (rng.uniform(0, 1, (n, d)) * 4).astype(np.float32)
y = (np.sin(X.sum(1)) + 0.3 * rng.standard_normal(n)).astype(np.float32)
z = X[rng.permutation(n)[:m]].copy()
f = agp.GP(agp.SqExponentialKernel(), ctx=ctx)
fx = f(agp.RowVecs(X), np.float32(0.1))
approx = agp.VFE(f(agp.RowVecs(z), 1e-4))
DEF = {"vfe_sk": 0, "vfe_chunk": 16384, "updk_max_k": 512, "upd128": 1}
for tag, params in (("default", {}), ("vfe_sk", {"vfe_sk": 1}), ("chunk32768", {"vfe_chunk": 32768}), ("no_updk", {"updk_max_k": 0, "upd128": 0}), ("default_again", {})):
    for k, v in {**DEF, **params}.items():
        ctx.set_param(k, v)
    ts, obj = [], None
    for rep in range(6):
        t0 = time.perf_counter()
        p = agp.posterior(approx, fx, y)
        obj = float(p.objective)
        ts.append(time.perf_counter() - t0)
        del p
    print(json.dumps({"setting": tag, "ms_med": float(np.median(ts[1:])) * 1e3, "ms_min": min(ts[1:]) * 1e3, "elbo": obj}), flush=True)
for k, v in DEF.items():
    ctx.set_param(k, v)
