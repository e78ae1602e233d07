"""Round 4 side measurements on one MI355X: (a) C4 pair time with the XCD-aware super-tile order on / off; (b) C5 (VFE fp32) with the new leaf;
(c) M = 4 096 fp64 Cholesky alone.  One JSON line each."""
import json
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import abstractgps_jl_amd as agp  # noqa: E402

ctx = agp.default_context(0)


def synth(n, d, seed):
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((n, d))
    return X, np.sin(X.sum(1)) + 0.1 * rng.standard_normal(n)


def pair_times(n, d, seed, reps, params):
    for k, v in params.items():
        ctx.set_param(k, v)
    x, y = synth(n, d, seed)
    fx = agp.GP(agp.SqExponentialKernel(), ctx=ctx)(agp.RowVecs(x), 0.01)
    ts, pot = [], []
    for _ in range(reps + 1):
        t0 = time.perf_counter()
        p = agp.posterior(fx, y)
        ts.append(time.perf_counter() - t0)
        pot.append(ctx.timings()["potrf_ms"])
        p.data.C.free()
    return min(ts[1:]) * 1e3, float(np.median(ts[1:])) * 1e3, min(pot[1:])


what = sys.argv[1:] or ["xcd", "c5", "m4096"]
if "m4096" in what:
    mn, md, pt = pair_times(4096, 3, 9, 6, {})
    print(json.dumps({"case": "N4096 pair", "ms_min": mn, "ms_med": md, "potrf_ms": pt}), flush=True)
if "c5" in what:
    rng = np.random.default_rng(5)
    n, m, d = 262144, 4096, 3
    X = (rng.uniform(0, 1, (n, d)) * 4).astype(np.float32)
    y = (np.sin(X.sum(1)) + 0.3 * rng.standard_normal(n)).astype(np.float32)
    z = X[rng.permutation(n)[:m]].copy()
    f = agp.GP(agp.SqExponentialKernel(), ctx=ctx)
    fx = f(agp.RowVecs(X), np.float32(0.1))
    approx = agp.VFE(f(agp.RowVecs(z), 1e-4))
    for tag, params in (("leaf_v2", {"leaf_v2": 1}), ("leaf_v1", {"leaf_v2": 0}), ("leaf_v2 again", {"leaf_v2": 1})):
        for k, v in params.items():
            ctx.set_param(k, v)
        ts = []
        for _ in range(6):
            t0 = time.perf_counter()
            p = agp.posterior(approx, fx, y)
            ts.append(time.perf_counter() - t0)
            obj = float(p.objective)
            del p
        flops = 2.0 * n * m * m + 2.0 * m**3 / 3
        print(json.dumps({"case": "C5", "setting": tag, "ms_min": min(ts[1:]) * 1e3, "ms_med": float(np.median(ts[1:])) * 1e3, "elbo": obj,
                          "frac_fp32": flops / min(ts[1:]) / 1e12 / 157.3}), flush=True)
    ctx.trim()
if "xcd" in what:
    for tag, params in (("xcd off", {"xcd_swizzle": 0}), ("xcd on", {"xcd_swizzle": 1}), ("xcd off again", {"xcd_swizzle": 0})):
        mn, md, pt = pair_times(65536, 3, 4, 3, params)
        print(json.dumps({"case": "C4 pair", "setting": tag, "ms_min": mn, "ms_med": md, "potrf_ms": pt}), flush=True)
    ctx.set_param("xcd_swizzle", 0)
    ctx.trim()
