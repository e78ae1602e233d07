#!/usr/bin/env python
"""Value-level parity at the full BASELINE sizes: the HIP path against the CPU oracle run on this box's host cores.

  python tools/fullsize_parity.py [--configs C2,C3,C3ard,C4,C5] [--out gpurun_out/fullsize_parity.jsonl] [--threads T]

For C2 / C3 / C3ard / C4 the oracle's in-place fused pair (oracle.gp_oracle.logpdf_and_posterior_inplace: one Fortran-
ordered N×N array, dpotrf('U') in place — SURVEY.md §8(d)) gives logpdf, logdet and α; the GPU's gp_posterior_fit must
agree to logpdf rel <= 1e-10 and ‖α−α_ref‖/‖α_ref‖ <= 1e-8.  For C5 the fp64 oracle VFE fit
(vfe_posterior + objective_from_posterior) is compared with the engine's fp32 fit (ELBO rel <= 1e-4, predictive mean
abs <= 1e-3) and fp64 fit (ELBO rel <= 1e-8).  One JSON line per config, with the oracle's phase timings (Gram / potrf /
solves: the real CPU baseline of the same box) and the BLAS threadpool description.  This is a checker: it imports
oracle/, the product never does.  /root/reference is not touched.
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def threadpools():
    try:
        from threadpoolctl import threadpool_info

        return [{k: p.get(k) for k in ("user_api", "internal_api", "version", "num_threads", "threading_layer")}
                for p in threadpool_info()]
    except Exception as e:  # noqa: BLE001
        return [{"error": str(e)}]


def relnorm(a, b):
    return float(np.linalg.norm(a - b) / np.linalg.norm(b))


def exact_config(name, agp, o, threads):
    if name == "C2":
        n, d, seed, kind, scale = 16384, 3, 2, o.SE, None
    elif name == "C3":
        n, d, seed, kind, scale = 32768, 8, 3, o.MATERN32, 0.5
    elif name == "C3ard":
        n, d, seed, kind, scale = 32768, 8, 3, o.MATERN32, np.linspace(0.25, 1.0, 8)
    elif name == "C4":
        n, d, seed, kind, scale = 65536, 3, 4, o.SE, None
    else:
        raise ValueError(name)
    x, y = o.synth_inputs(n, d, seed)
    k = agp.Kernel(kind)
    if scale is not None:
        k = k @ (agp.ScaleTransform(float(scale)) if np.ndim(scale) == 0 else agp.ARDTransform(scale))
    f = agp.GP(k)
    fx = f(agp.RowVecs(x), 0.01)
    agp.posterior(fx, y).data.C.free()  # warm-up
    t0 = time.perf_counter()
    post = agp.posterior(fx, y)
    t_gpu = time.perf_counter() - t0
    lp_gpu, alpha_gpu = float(post.logpdf_value), np.array(post.data.alpha)
    post.data.C.free()
    agp.default_context().trim()
    tm = {}
    t0 = time.perf_counter()
    lp, alpha, logdet = o.logpdf_and_posterior_inplace(o.FiniteGP(o.GP(o.Kernel(kind, 1.0, scale)), x, 0.01), y, threads=threads,
                                                       timings=tm)
    t_cpu = time.perf_counter() - t0
    rec = {"config": name, "n": n, "d": d, "kernel": o.KERNEL_NAMES[kind], "scale": None if scale is None else np.asarray(scale).tolist(),
           "logpdf_gpu": lp_gpu, "logpdf_oracle": lp, "logdet_oracle": logdet, "logpdf_rel": abs(lp_gpu - lp) / abs(lp),
           "alpha_rel": relnorm(alpha_gpu, alpha), "gpu_pair_s": t_gpu, "oracle_pair_s": t_cpu, "oracle_phases_s": tm,
           "oracle_points_per_s_fused": n / t_cpu,
           # the reference's own logpdf + posterior build and factor the Gram matrix twice
           "oracle_points_per_s_two_factorisations": n / (2 * (tm["gram_s"] + tm["potrf_s"]) + tm["solves_s"]),
           "oracle_potrf_gflops": n**3 / 3 / tm["potrf_s"] / 1e9, "gram_threads": threads,
           "tol": {"logpdf_rel": 1e-10, "alpha_rel": 1e-8}}
    rec["pass"] = bool(rec["logpdf_rel"] <= 1e-10 and rec["alpha_rel"] <= 1e-8)
    return rec


def c5_config(agp, o, threads):
    n, m, d = 262144, 4096, 3
    rng = np.random.default_rng(5)  # same generator as tests/test_gpu_fullsize.py / tools/bench_configs.py
    X = rng.uniform(0, 1, (n, d)) * 4
    y = np.sin(X.sum(1)) + 0.3 * rng.standard_normal(n)
    z = X[rng.permutation(n)[:m]].copy()
    xs = rng.uniform(0, 1, (4096, d)) * 4
    s2, jitter = 0.1, 1e-4
    X32, z32, y32, xs32 = (a.astype(np.float32) for a in (X, z, y, xs))
    out = {"config": "C5", "n": n, "m": m, "d": d, "sigma2": s2, "jitter": jitter}
    f = agp.GP(agp.SqExponentialKernel())
    # the engine in fp32 (the BASELINE configuration) and in fp64
    for tag, (Xa, za, ya, xsa) in {"f32": (X32, z32, y32, xs32), "f64": (X, z, y, xs)}.items():
        vfe = agp.VFE(f(agp.RowVecs(za), jitter))
        fx = f(agp.RowVecs(Xa), s2)
        agp.posterior(vfe, fx, ya)
        t0 = time.perf_counter()
        ap = agp.posterior(vfe, fx, ya)
        out[f"gpu_{tag}_fit_s"] = time.perf_counter() - t0
        out[f"elbo_gpu_{tag}"] = float(ap.objective)
        mm, vv = ap.mean_and_var(agp.RowVecs(xsa))
        out[f"_mean_{tag}"], out[f"_var_{tag}"] = np.asarray(mm, dtype=np.float64), np.asarray(vv, dtype=np.float64)
        del ap
    agp.default_context().trim()
    # fp64 oracle on the fp32-representable inputs (the values both engines actually saw in fp32 mode)
    of = o.GP(o.Kernel(o.SE))
    Xo, zo, yo, xso = (a.astype(np.float64) for a in (X32, z32, y32, xs32))
    t0 = time.perf_counter()
    ofx = o.FiniteGP(of, Xo, s2)
    op = o.vfe_posterior(of, zo, jitter, ofx, yo)
    elbo32in = o.objective_from_posterior(op, ofx, yo, vfe=True)
    mo, vo = op.mean_and_var(xso)
    out["oracle_fit_s"] = time.perf_counter() - t0
    out["elbo_oracle_f32_inputs"] = elbo32in
    out["elbo_rel_f32"] = abs(out["elbo_gpu_f32"] - elbo32in) / abs(elbo32in)
    out["mean_abs_f32"] = float(np.max(np.abs(out["_mean_f32"] - mo)))
    out["var_abs_f32"] = float(np.max(np.abs(out["_var_f32"] - vo)))
    del op
    # fp64 oracle on the fp64 inputs vs the fp64 engine
    ofx = o.FiniteGP(of, X, s2)
    op = o.vfe_posterior(of, z, jitter, ofx, y)
    elbo64 = o.objective_from_posterior(op, ofx, y, vfe=True)
    mo, vo = op.mean_and_var(xs)
    out["elbo_oracle_f64"] = elbo64
    out["elbo_rel_f64"] = abs(out["elbo_gpu_f64"] - elbo64) / abs(elbo64)
    out["mean_abs_f64"] = float(np.max(np.abs(out["_mean_f64"] - mo)))
    out["var_abs_f64"] = float(np.max(np.abs(out["_var_f64"] - vo)))
    for kname in [k_ for k_ in out if k_.startswith("_")]:
        del out[kname]
    out["tol"] = {"elbo_rel_f32": 1e-4, "mean_abs_f32": 1e-3, "elbo_rel_f64": 1e-8, "mean_abs_f64": 1e-6}
    out["pass"] = bool(out["elbo_rel_f32"] <= 1e-4 and out["mean_abs_f32"] <= 1e-3 and out["elbo_rel_f64"] <= 1e-8
                       and out["mean_abs_f64"] <= 1e-6)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", default="C2,C3,C3ard,C4,C5")
    ap.add_argument("--out", default=str(ROOT / "gpurun_out" / "fullsize_parity.jsonl"))
    ap.add_argument("--threads", type=int, default=min(32, os.cpu_count() or 1), help="threads of the oracle's Gram assembly")
    args = ap.parse_args()
    import abstractgps_jl_amd as agp
    from oracle import gp_oracle as o

    Path(args.out).parent.mkdir(parents=True, exist_ok=True)
    host = {"cpu_count": os.cpu_count(), "threadpools": threadpools()}
    ok = True
    with open(args.out, "a") as fh:
        for name in args.configs.split(","):
            t0 = time.perf_counter()
            try:
                rec = c5_config(agp, o, args.threads) if name == "C5" else exact_config(name, agp, o, args.threads)
            except Exception as e:  # noqa: BLE001  (keep going: one config out of host memory must not lose the others)
                rec = {"config": name, "error": repr(e), "pass": False}
            rec["host"] = host
            rec["wall_s"] = time.perf_counter() - t0
            print(json.dumps(rec), flush=True)
            fh.write(json.dumps(rec) + "\n")
            fh.flush()
            ok = ok and rec["pass"]
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
