"""Round-3 sweep on one MI355X: CU-partitioned look-ahead (cu_split / cu_split_nb / cu_split_tail) for the mid-size exact configs,
with a logpdf check against the unpartitioned run of the same process (identical inputs).  One JSON line per measurement.
   python tools/sweep_r3.py [C2] [C3] [C4] [quick]"""
import json
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import abstractgps_jl_amd as agp  # noqa: E402

DEFAULTS = {"nb": 2048, "lookahead": 1, "cu_split": 0, "cu_split_nb": 512, "cu_split_tail": 8192, "cu_split_max_n": 40000}


def synth(n, d, seed):
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((n, d))
    return X, np.sin(X.sum(axis=1)) + 0.1 * rng.standard_normal(n)


def exact(tag, n, d, seed, kernel, params, reps=3, ref=None):
    x, y = synth(n, d, seed)
    ctx = agp.default_context(0)
    for k, v in {**DEFAULTS, **params}.items():
        ctx.set_param(k, v)
    fx = agp.GP(kernel)(agp.RowVecs(x), 0.01)
    best, lp = None, None
    for i in range(reps + 1):
        t0 = time.perf_counter()
        post = agp.posterior(fx, y)
        dt = time.perf_counter() - t0
        tm = ctx.timings()
        lp = float(post.logpdf_value)
        post.data.C.free()
        if i > 0 and (best is None or dt < best[0]):
            best = (dt, tm)
    dt, tm = best
    rec = {"config": tag, "n": n, "params": params, "pair_ms": round(dt * 1e3, 3), "pair_tflops": round((n**3 / 3 + 3 * n**2) / dt / 1e12, 2),
           "frac": round((n**3 / 3 + 3 * n**2) / dt / 78.6e12, 4), "phases_ms": {k: round(tm[k], 3) for k in ("assemble_ms", "potrf_ms", "solve_ms")},
           "logpdf": lp}
    if ref is not None:
        rec["logpdf_rel_vs_unsplit"] = abs(lp - ref) / abs(ref)
    print(json.dumps(rec), flush=True)
    return lp


if __name__ == "__main__":
    which = [a for a in sys.argv[1:] if a != "quick"] or ["C2", "C3"]
    quick = "quick" in sys.argv
    cfgs = {"C2": (16384, 3, 2, agp.SqExponentialKernel()), "C3": (32768, 8, 3, agp.Matern32Kernel() @ agp.ScaleTransform(0.5)),
            "C4": (65536, 3, 4, agp.SqExponentialKernel()), "M8": (8192, 3, 12, agp.SqExponentialKernel())}
    for name in which:
        n, d, seed, kern = cfgs[name]
        reps = 1 if name == "C4" else 3
        ref = exact(name, n, d, seed, kern, {}, reps=reps)
        exact(name, n, d, seed, kern, {"nb": 512}, reps=reps, ref=ref)           # narrow panels alone (no partition)
        grid = [(32, 512, 8192)] if quick else [(32, 512, 8192), (32, 256, 8192), (32, 1024, 8192), (16, 512, 8192), (64, 512, 8192),
                                                (32, 512, 4096), (32, 512, 12288), (64, 512, 4096), (48, 512, 6144)]
        for r, nb, tail in grid:
            if tail >= n:
                continue
            exact(name, n, d, seed, kern, {"cu_split": r, "cu_split_nb": nb, "cu_split_tail": tail, "cu_split_max_n": 1 << 30}, reps=reps, ref=ref)
    for k, v in DEFAULTS.items():
        agp.default_context(0).set_param(k, v)
