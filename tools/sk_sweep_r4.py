"""sk_max_tiles sweep (which launches take the persistent stream-K GEMM) at N = 8 192 / C2 / C3: pair ms (min of 5)."""
import json, sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import abstractgps_jl_amd as agp
ctx = agp.default_context(0)
for name, n, d, seed, kern in (("N8192", 8192, 3, 8, agp.SqExponentialKernel()), ("C2", 16384, 3, 2, agp.SqExponentialKernel()),
                               ("C3", 32768, 8, 3, agp.Matern32Kernel() @ agp.ScaleTransform(0.5))):
    rng = np.random.default_rng(seed); X = rng.standard_normal((n, d)); y = np.sin(X.sum(1)) + 0.1 * rng.standard_normal(n)
    fx = agp.GP(kern, ctx=ctx)(agp.RowVecs(X), 0.01)
    for skt in (4096, 2048, 1024, 512, 256, 8192, 16384):
        ctx.set_param("sk_max_tiles", skt)
        ts = []
        for _ in range(6):
            t0 = time.perf_counter(); p = agp.posterior(fx, y); ts.append(time.perf_counter() - t0); p.data.C.free()
        print(json.dumps({"case": name, "sk_max_tiles": skt, "ms_min": min(ts[1:]) * 1e3, "ms_med": float(np.median(ts[1:])) * 1e3}), flush=True)
ctx.set_param("sk_max_tiles", 4096)
