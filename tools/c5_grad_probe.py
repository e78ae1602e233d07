"""C5-size value + gradient of the sparse objective: N = 262 144, M = 4 096, D = 3, fp32 and fp64 handles — ms per fit, ms per
gradient (with its three phases), and the gradient along a random direction in (variance, scale, noise, z) against a central difference of fp64 fits.
    python tools/c5_grad_probe.py [reps=3]"""
import sys, time, json
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import abstractgps_jl_amd as agp

reps = int(dict(a.split("=") for a in sys.argv[1:] if "=" in a).get("reps", 3))
n, m, d = 262144, 4096, 3
rng = np.random.default_rng(5)
X = rng.normal(size=(n, d))
y = np.sin(X.sum(1)) + 0.1 * rng.normal(size=n)
Z = X[rng.choice(n, m, replace=False)].copy()
ctx = agp.default_context(0)
out = {"n": n, "m": m}
for dt in (np.float32, np.float64):
    Xd, Zd, yd = X.astype(dt), Z.astype(dt), y.astype(dt)
    f = agp.GP(1.0 * agp.SqExponentialKernel() @ agp.ScaleTransform(1.0))
    fx, a = f(agp.RowVecs(Xd), dt(0.1)), agp.VFE(f(agp.RowVecs(Zd), 1e-4))
    post = agp.posterior(a, fx, yd)
    post.objective_grad()
    tf, tg, ph = [], [], None
    for _ in range(reps):   # fits, then gradients of the resident posterior (a gradient right behind an fp32 fit has measured 10–20 ms slower than the steady pass)
        del post      # (its blocks go back to the cache outside the timed statement)
        t0 = time.perf_counter(); post = agp.posterior(a, fx, yd); tf.append((time.perf_counter() - t0) * 1e3)
    for _ in range(reps):
        t1 = time.perf_counter(); g = post.objective_grad(); tg.append((time.perf_counter() - t1) * 1e3)
        ph = ctx.timings()
    key = "f32" if dt is np.float32 else "f64"
    out[key] = {"fit_ms": float(np.median(tf)), "grad_ms": float(np.median(tg)), "grad_phases_ms": {k: ph[k] for k in ("assemble_ms", "potrf_ms", "solve_ms", "total_ms")},
                "objective": float(post.objective), "variance": g["variance"], "scale": g["scale"], "noise": g["noise"]}
    if dt is np.float64:
        g64 = g
    else:
        g32 = g
# directional derivative in fp64
dZ = rng.normal(size=Z.shape); dirs = np.array([0.3, -0.2, 0.05]); h = 1e-5
def obj(var, sc, s2, Zc):
    f = agp.GP(var * agp.SqExponentialKernel() @ agp.ScaleTransform(sc))
    return float(agp.approx_log_evidence(agp.VFE(f(agp.RowVecs(Zc), 1e-4)), f(agp.RowVecs(X), s2), y))
f = agp.GP(1.0 * agp.SqExponentialKernel() @ agp.ScaleTransform(1.0))
val, g = agp.elbo_and_grad(agp.VFE(f(agp.RowVecs(Z), 1e-4)), f(agp.RowVecs(X), 0.1), y)
fd = (obj(1 + h * dirs[0], 1 + h * dirs[1], 0.1 + h * dirs[2], Z + h * dZ) - obj(1 - h * dirs[0], 1 - h * dirs[1], 0.1 - h * dirs[2], Z - h * dZ)) / (2 * h)
an = g["variance"] * dirs[0] + g["scale"] * dirs[1] + g["noise"] * dirs[2] + float(np.sum(g["z"] * dZ))
out["directional_fp64"] = {"analytic": an, "central_difference": fd, "rel": abs(an - fd) / abs(fd)}
out["f32_vs_f64"] = {k: float(abs(g32[k] - g64[k]) / max(abs(g64[k]), 1.0)) for k in ("variance", "scale", "noise")}
out["f32_vs_f64"]["y"] = float(np.max(np.abs(g32["y"] - g64["y"])) / np.max(np.abs(g64["y"])))
print(json.dumps(out))
