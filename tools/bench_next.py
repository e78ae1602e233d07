"""Timings of the 'what callers do next' entry points on one MI355X (one JSON line each): predictive mean_and_var / cov,
sequential conditioning, rand, logpdf gradient."""
import json
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tools"))
import abstractgps_jl_amd as agp  # noqa: E402
from _synth import synth_inputs  # noqa: E402


def t(fn, reps=2):
    best = None
    for _ in range(reps):
        t0 = time.perf_counter()
        r = fn()
        dt = time.perf_counter() - t0
        best = dt if best is None or dt < best else best
    return best, r


def main(n=16384, ns=4096):
    x, y = synth_inputs(n, 3, 2)
    f = agp.GP(agp.SqExponentialKernel() @ agp.ScaleTransform(0.9))
    fx = f(agp.RowVecs(x), 0.01)
    dt_fit, post = t(lambda: agp.posterior(fx, y), 2)
    xs = x[:ns] + 0.05
    dt_mv, (m, v) = t(lambda: post.mean_and_var(agp.RowVecs(xs)))
    dt_cov, cm = t(lambda: post.cov(agp.RowVecs(xs[:1024])), 1)
    n2 = n // 8
    x2, y2 = synth_inputs(n2, 3, 77)
    dt_seq, p2 = t(lambda: agp.posterior(post(agp.RowVecs(x2), 0.01), y2), 1)
    dt_rand, smp = t(lambda: agp.rand(fx, 4, rng=np.random.default_rng(0)), 1)
    dt_grad, (lp, g) = t(lambda: agp.logpdf_and_grad(fx, y), 1)
    print(json.dumps({"n": n, "fit_ms": dt_fit * 1e3, f"mean_and_var_{ns}_ms": dt_mv * 1e3, "cov_1024_ms": dt_cov * 1e3,
                      f"sequential_update_{n2}_ms": dt_seq * 1e3, "rand_4_ms": dt_rand * 1e3, "logpdf_and_grad_ms": dt_grad * 1e3,
                      "var_min": float(v.min()), "grad_scale": g["scale"], "grad_noise": float(g["noise"]),
                      "logpdf": float(lp)}), flush=True)


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 16384, int(sys.argv[2]) if len(sys.argv) > 2 else 4096)
