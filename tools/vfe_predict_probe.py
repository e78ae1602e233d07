"""VFE predictive marginals at C5 (N = 262 144, M = 4 096) with and without the cached inverse diagonal blocks of Lz / Ld ("dib_nb" 2048 / 0):
first call on a fresh handle (builds the cache) and the median of the next three, plus the max difference between the two answers."""
import json, sys, time
import numpy as np
sys.path.insert(0, ".")
import abstractgps_jl_amd as agp

n, m, d = 262144, 4096, 3
rng = np.random.default_rng(5)
X = (rng.uniform(0, 1, (n, d)) * 4).astype(np.float32)
y = (np.sin(X.sum(1)) + 0.3 * rng.standard_normal(n)).astype(np.float32)
z = X[rng.permutation(n)[:m]].copy()
xs = (rng.uniform(0, 1, (4096, d)) * 4).astype(np.float32)
f = agp.GP(agp.SqExponentialKernel())
fx = f(agp.RowVecs(X), np.float32(0.1))
approx = agp.VFE(f(agp.RowVecs(z), 1e-4))
ctx = agp.default_context()
out = {}
ans = {}
for nb in (0, 2048, 0, 2048):
    ctx.set_param("dib_nb", nb)
    post = agp.posterior(approx, fx, y)
    ts = []
    for _ in range(4):
        t0 = time.perf_counter()
        mp, vp = post.mean_and_var(agp.RowVecs(xs))
        ts.append((time.perf_counter() - t0) * 1e3)
    out.setdefault(f"dib_nb={nb}", []).append({"first_ms": ts[0], "later_ms_median": float(np.median(ts[1:]))})
    ans[nb] = (np.asarray(mp, dtype=np.float64), np.asarray(vp, dtype=np.float64))
    del post
ctx.set_param("dib_nb", 2048)
out["max_abs_diff_mean"] = float(np.abs(ans[0][0] - ans[2048][0]).max())
out["max_abs_diff_var"] = float(np.abs(ans[0][1] - ans[2048][1]).max())
out["var_min"] = float(ans[2048][1].min())
print(json.dumps(out), flush=True)
