"""Vector solves as one persistent launch ("trsv_persist") against two launches per 256-column block, alternated inside ONE process: the (logpdf, posterior) pair at
N = 4 096 / 16 384 (C2) / 32 768 (C3's size, SE D = 3) / 65 536 (C4) and the C5 sparse fit; one warm-up, then the median of `reps` pairs per setting and round.
    python tools/sweep_ab.py [rounds=2]"""
import json
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import abstractgps_jl_amd as agp  # noqa: E402

rounds = 2
for a in sys.argv[1:]:
    k, v = a.split("=")
    if k == "rounds":
        rounds = int(v)
ctx = agp.default_context()


def pair_times(n, reps):
    rng = np.random.default_rng(n)
    X = rng.standard_normal((n, 3))
    y = np.sin(X.sum(1)) + 0.1 * rng.standard_normal(n)
    fx = agp.GP(agp.SqExponentialKernel(), ctx=ctx)(agp.RowVecs(X), 0.01)

    def run():
        ts, lp, a0 = [], None, None
        for i in range(reps + 1):
            t0 = time.perf_counter()
            p = agp.posterior(fx, y)
            dt = (time.perf_counter() - t0) * 1e3
            lp, a0 = float(p.logpdf_value), float(p.data.alpha[0])
            p.data.C.free()
            if i:
                ts.append(dt)
        return float(np.median(ts)), lp, a0

    return run


def c5_times(reps):
    n, m, d = 262144, 4096, 3
    rng = np.random.default_rng(5)
    X = (rng.uniform(0, 1, (n, d)) * 4).astype(np.float32)
    y = (np.sin(X.sum(1)) + 0.3 * rng.standard_normal(n)).astype(np.float32)
    z = X[rng.permutation(n)[:m]].copy()
    f = agp.GP(agp.SqExponentialKernel(), ctx=ctx)
    fx = f(agp.RowVecs(X), np.float32(0.1))
    approx = agp.VFE(f(agp.RowVecs(z), 1e-4))

    def run():
        ts, obj = [], None
        for i in range(reps + 1):
            t0 = time.perf_counter()
            p = agp.posterior(approx, fx, y)
            dt = (time.perf_counter() - t0) * 1e3
            obj = float(p.objective)
            del p
            if i:
                ts.append(dt)
        return float(np.median(ts)), obj, 0.0

    return run


cases = [("N4096", pair_times(4096, 15)), ("C2", pair_times(16384, 9)), ("N32768", pair_times(32768, 5)), ("C4", pair_times(65536, 3)), ("C5", c5_times(5))]
for name, run in cases:
    for rnd in range(rounds):
        for persist in (0, 1):
            ctx.set_param("trsv_persist", persist)
            ms, v, a0 = run()
            print(json.dumps({"case": name, "round": rnd, "trsv_persist": persist, "ms_median": round(ms, 4), "value": v, "alpha0": a0}), flush=True)
    ctx.trim()
ctx.set_param("trsv_persist", 1)
