"""sk_min_k sweep (stream-K only for k ranges at least this long) at N = 4 096 / 8 192 / C2: pair ms (min of 5)."""
import json, sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import abstractgps_jl_amd as agp
ctx = agp.default_context(0)
for name, n in (("N4096", 4096), ("N8192", 8192), ("C2", 16384)):
    rng = np.random.default_rng(2); X = rng.standard_normal((n, 3)); y = np.sin(X.sum(1)) + 0.1 * rng.standard_normal(n)
    fx = agp.GP(agp.SqExponentialKernel(), ctx=ctx)(agp.RowVecs(X), 0.01)
    for k in (0, 256, 512, 1024, 4096):
        ctx.set_param("sk_min_k", k)
        ts = []
        for _ in range(6):
            t0 = time.perf_counter(); p = agp.posterior(fx, y); ts.append(time.perf_counter() - t0); p.data.C.free()
        print(json.dumps({"case": name, "sk_min_k": k, "ms_min": round(min(ts[1:]) * 1e3, 3), "ms_med": round(float(np.median(ts[1:])) * 1e3, 3)}), flush=True)
ctx.set_param("sk_min_k", 0)
