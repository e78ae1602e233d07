"""Timing + size-independent parity properties of the BASELINE.json configs other than the bench workload
(C2, C3 incl. the ARD variant, C5) on one MI355X.  One JSON line per config (kept under profiles/)."""
import json
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import abstractgps_jl_amd as agp  # noqa: E402
from oracle import gp_oracle as o  # noqa: E402


def exact(tag, n, d, seed, kernel, okernel, sigma2=0.01, reps=3, oracle_check=False):
    x, y = o.synth_inputs(n, d, seed)
    ctx = agp.default_context(0)
    f = agp.GP(kernel)
    wrap = (lambda a: a) if d == 1 else agp.RowVecs
    fx = f(wrap(x), sigma2)
    best = None
    for _ in range(reps):
        t0 = time.perf_counter()
        post = agp.posterior(fx, y)
        dt = time.perf_counter() - t0
        tm = ctx.timings()
        if best is None or dt < best[0]:
            best = (dt, tm)
        if _ < reps - 1:
            post.data.C.free()
    dt, tm = best
    alpha = post.data.alpha
    idx = np.linspace(0, n - 1, 512).astype(int)
    m_tr = post.mean(wrap(x[idx]))                       # independent device path: K α = δ − σ² α
    resid = float(np.max(np.abs(m_tr - (y[idx] - sigma2 * alpha[idx]))))
    Krows = o.kernelmatrix(okernel, x[idx[:64]], x)              # host recomputation of 64 rows of (K + σ²I) α = δ
    resid_host = float(np.max(np.abs(Krows @ alpha + sigma2 * alpha[idx[:64]] - y[idx[:64]])))
    xs = x[:min(n, 1024)] + 0.05
    t0 = time.perf_counter()
    mp, vp = post.mean_and_var(wrap(xs))
    t_pred = time.perf_counter() - t0
    out = {"config": tag, "n": n, "d": d, "pair_ms": dt * 1e3, "points_per_s": n / dt,
           "pair_tflops": (n**3 / 3 + 3 * n**2) / dt / 1e12, "phases_ms": {k: round(tm[k], 3) for k in ("assemble_ms", "potrf_ms", "solve_ms")},
           "logpdf": float(post.logpdf_value), "resid_device": resid, "resid_host_rows": resid_host,
           "mean_and_var_1024_ms": t_pred * 1e3, "var_min": float(vp.min())}
    if oracle_check:
        lp, _ = o.logpdf_and_posterior(o.FiniteGP(o.GP(okernel), x, sigma2), y)
        out["oracle_logpdf_rel"] = abs(float(post.logpdf_value) - lp) / abs(lp)
    print(json.dumps(out), flush=True)
    post.data.C.free()


def vfe(n=262144, m=4096, d=3, reps=2):
    rng = np.random.default_rng(5)
    X = (rng.uniform(0, 1, (n, d)) * 4).astype(np.float32)
    y = (np.sin(X.sum(1)) + 0.3 * rng.standard_normal(n)).astype(np.float32)
    z = X[rng.permutation(n)[:m]].copy()
    xs = (rng.uniform(0, 1, (4096, d)) * 4).astype(np.float32)
    f = agp.GP(agp.SqExponentialKernel())
    fx = f(agp.RowVecs(X), np.float32(0.1))
    approx = agp.VFE(f(agp.RowVecs(z), 1e-4))
    best = None
    for _ in range(reps):
        t0 = time.perf_counter()
        post = agp.posterior(approx, fx, y)
        t_fit = time.perf_counter() - t0
        t0 = time.perf_counter()
        mp, vp = post.mean_and_var(agp.RowVecs(xs))
        t_pred = time.perf_counter() - t0
        if best is None or t_fit < best[0]:
            best = (t_fit, t_pred)
    t_fit, t_pred = best
    flops = 2.0 * n * m * m + 2.0 * m**3 / 3
    # parity at this size against the fp64 oracle on a 32 768-point subset would change the problem; instead the
    # property test: ELBO(fp32 inputs, fp64 arithmetic on the device) vs fp32 path
    e32 = float(post.objective)
    e64 = float(agp.elbo(agp.VFE(f(agp.RowVecs(z.astype(np.float64)), 1e-4)), f(agp.RowVecs(X.astype(np.float64)), 0.1),
                         y.astype(np.float64)))
    print(json.dumps({"config": "C5 VFE fp32", "n": n, "m": m, "fit_ms": t_fit * 1e3, "points_per_s": n / t_fit,
                      "fit_tflops_fp32": flops / t_fit / 1e12, "predict_4096_ms": t_pred * 1e3, "elbo_fp32": e32,
                      "elbo_fp64_device": e64, "elbo_rel_diff": abs(e32 - e64) / abs(e64),
                      "pred_mean_absmax": float(np.abs(mp).max()), "pred_var_min": float(vp.min())}), flush=True)


if __name__ == "__main__":
    which = sys.argv[1:] or ["C1", "C2", "C3", "C3ard", "C5"]
    if "C1" in which:
        exact("C1", 256, 1, 1, agp.SqExponentialKernel(), o.Kernel(o.SE), oracle_check=True)
    if "C2" in which:
        exact("C2", 16384, 3, 2, agp.SqExponentialKernel(), o.Kernel(o.SE))
    if "C3" in which:
        exact("C3 Matern32∘ScaleTransform(0.5)", 32768, 8, 3, agp.Matern32Kernel() @ agp.ScaleTransform(0.5),
              o.Kernel(o.MATERN32, 1.0, 0.5))
    if "C3ard" in which:
        v = np.linspace(0.25, 1, 8)
        exact("C3 Matern32∘ARDTransform", 32768, 8, 3, agp.Matern32Kernel() @ agp.ARDTransform(v), o.Kernel(o.MATERN32, 1.0, v))
    if "C5" in which:
        vfe()
