"""The reference's sparse-GP example on the device: examples/0-intro-1d/script.jl:359-420 maximises the ELBO of a VFE approximation over the kernel variance,
the inverse lengthscale (both through softplus) and the pseudo-point locations (through the logistic function) with LBFGS — by finite differences there; here
every LBFGS evaluation is one `elbo_and_grad` call (gp_vfe_fit + gp_vfe_grad).  Synthetic 1-D data of any size.
    python tools/train_sparse_example.py [n=200000] [m=32] [iters=40]"""
import json
import sys
import time
from pathlib import Path

import numpy as np
from scipy.optimize import minimize

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import abstractgps_jl_amd as agp  # noqa: E402


def softplus(v):
    return np.logaddexp(0.0, v)


def logistic(v):
    return 1.0 / (1.0 + np.exp(-v))


def make_objective(x, y, noise_var, jitter=1e-6, calls=None):
    """negative_elbo(params) of the reference example and its gradient: params = [variance', inverse lengthscale', z'...] before the positivity /
    unit-interval transforms (script.jl:385-394)."""

    def fun(params):
        var, sc, z = softplus(params[0]), softplus(params[1]), logistic(params[2:])
        f = agp.GP(var * agp.Matern52Kernel() @ agp.ScaleTransform(sc))
        val, g = agp.elbo_and_grad(agp.VFE(f(z, jitter)), f(x, noise_var), y)
        grad = np.empty_like(params)
        grad[0] = g["variance"] * logistic(params[0])          # d softplus = logistic
        grad[1] = g["scale"] * logistic(params[1])
        grad[2:] = g["z"] * z * (1.0 - z)                       # d logistic
        if calls is not None:
            calls.append(float(val))
        return -float(val), -grad

    return fun


def main():
    opt = dict(a.split("=") for a in sys.argv[1:] if "=" in a)
    n, m, iters = int(opt.get("n", 200000)), int(opt.get("m", 32)), int(opt.get("iters", 40))
    rng = np.random.default_rng(0)
    x = rng.random(n)
    y = np.sin(4 * np.pi * x) + np.cos(11 * x) * x + 0.3 * rng.standard_normal(n)
    calls = []
    fun = make_objective(x, y, 0.09, calls=calls)
    p0 = rng.random(2 + m)
    t0 = time.perf_counter()
    v0 = -fun(p0)[0]
    res = minimize(fun, p0, jac=True, method="L-BFGS-B", options={"maxiter": iters})
    dt = time.perf_counter() - t0
    print(json.dumps({"n": n, "m": m, "elbo_start": v0, "elbo_end": -float(res.fun), "evaluations": len(calls), "s_total": dt, "ms_per_evaluation": dt / len(calls) * 1e3,
                      "variance": float(softplus(res.x[0])), "inverse_lengthscale": float(softplus(res.x[1])), "status": res.message if isinstance(res.message, str) else res.message.decode()}))


if __name__ == "__main__":
    main()
