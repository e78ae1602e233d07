"""Timing sweeps on the GPU box (not a test): prints one JSON line per measurement."""
import argparse
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


def fit_time(ctx, n, d, nb, la, reps=2, timed=True):
    x, y = synth_inputs(n, d, 4)
    ctx.set_param("nb", nb)
    ctx.set_param("lookahead", la)
    ctx.set_param("time_kernels", 1 if timed else 0)
    f = agp.GP(agp.SqExponentialKernel(), ctx=ctx)
    fx = f(agp.RowVecs(x) if d > 1 else x, 0.01)
    best = None
    for _ in range(reps):
        t0 = time.perf_counter()
        post = agp.posterior(fx, y)
        dt = time.perf_counter() - t0
        tm = ctx.timings()
        post.data.C.free()
        if best is None or dt < best[0]:
            best = (dt, tm, float(post.logpdf_value))
    dt, tm, lp = best
    fl = n**3 / 3 + 3 * n**2
    out = {"n": n, "d": d, "nb": nb, "lookahead": la, "wall_s": dt, "pair_tflops": fl / dt / 1e12,
           "points_per_s": n / dt, "logpdf": lp, **{k: round(v, 3) if isinstance(v, float) else v for k, v in tm.items()}}
    if tm["gemm_ms"] > 0:
        out["gemm_tflops"] = tm["gemm_flops"] / tm["gemm_ms"] / 1e9
    print(json.dumps(out), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", default="4096,16384")
    ap.add_argument("--nbs", default="0,1024,2048,4096")
    ap.add_argument("--big", type=int, default=0)
    ap.add_argument("--params", default="", help="comma list name=value applied to the ctx (e.g. nb=1024,leaf_group=256)")
    args = ap.parse_args()
    ctx = agp.Context(0)
    for kv in [t for t in args.params.split(",") if t]:
        k, v = kv.split("=")
        ctx.set_param(k, int(v))
    c = agp._lib.C.c_double()
    agp._lib.check(ctx.lib.gp_bench_mfma_f64(ctx.handle, 20000, agp._lib.C.byref(c)))
    print(json.dumps({"mfma_f64_ceiling_tflops": c.value}), flush=True)
    for n in [int(s) for s in args.sizes.split(",")]:
        for nb in [int(s) for s in args.nbs.split(",")]:
            for la in ([0] if nb == 0 else [0, 1]):
                try:
                    fit_time(ctx, n, 3, nb, la)
                except Exception as e:  # keep sweeping
                    print(json.dumps({"n": n, "nb": nb, "error": repr(e)}), flush=True)
    if args.big:
        for nb in (2048, 4096):
            fit_time(ctx, args.big, 3, nb, 1, reps=1)


if __name__ == "__main__":
    main()
