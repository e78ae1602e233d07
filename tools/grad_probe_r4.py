"""Round 4: value + gradient at C4, twice per setting, with the engine's phase timings (is the +1 s against mid-round in the factorisation, in C^-1, or outside the kernels?)."""
import json
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import abstractgps_jl_amd as agp  # noqa: E402

ctx = agp.default_context(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
rng = np.random.default_rng(4)
x = rng.standard_normal((n, 3)); y = np.sin(x.sum(1)) + 0.1 * rng.standard_normal(n)
fx = agp.GP(agp.SqExponentialKernel() @ agp.ScaleTransform(1.0), ctx=ctx)(agp.RowVecs(x), 0.01)
for tag, params in (("default", {}), ("pool_cap_192GB", {"pool_cap_mb": 196608}), ("no_updk", {"updk_max_k": 0, "upd128": 0, "pool_cap_mb": 196608})):
    for k, v in params.items():
        ctx.set_param(k, v)
    for rep in range(3):
        t0 = time.perf_counter()
        lp, g = agp.logpdf_and_grad(fx, y)
        dt = time.perf_counter() - t0
        tm = ctx.timings()
        print(json.dumps({"setting": tag, "rep": rep, "ms": round(dt * 1e3, 1), "logpdf": float(lp), "timings": {k: round(v, 1) for k, v in tm.items() if isinstance(v, float) and v}}), flush=True)
