"""Round 5: the SURVEY §8(f) rows (what an optimiser loop / a plotting caller runs) under ctx-parameter settings, on one MI355X —
predictive marginals at 4 096 points, the full 1 024² covariance, sequential conditioning on 8 192 new observations, value + gradient.
Same-size warm-up, then the median of 3; one JSON line per (row, setting).

  python tools/r5_next.py 65536 -- dib_nb=0 dib_nb=2048 dib_nb=1024
"""
import json
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import abstractgps_jl_amd as agp  # noqa: E402

PK = 78.6e12


def med3(fn, reps=3):
    fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts)), ts


def main():
    args = sys.argv[1:]
    cut = args.index("--") if "--" in args else len(args)
    n = int(args[0]) if cut > 0 else 65536
    rows = args[1:cut] or ["mv", "cov", "upd", "grad"]
    settings = args[cut + 1:] or ["base"]
    ctx = agp.default_context(0)
    rng = np.random.default_rng(4 if n == 65536 else 2)
    x = rng.standard_normal((n, 3))
    y = np.sin(x.sum(1)) + 0.1 * rng.standard_normal(n)
    fx = agp.GP(agp.SqExponentialKernel() @ agp.ScaleTransform(1.0), ctx=ctx)(agp.RowVecs(x), 0.01)
    post = agp.posterior(fx, y)
    r2 = np.random.default_rng(11)
    xs = r2.standard_normal((4096, 3))
    n2 = 8192
    x2 = r2.standard_normal((n2, 3))
    y2 = np.sin(x2.sum(1)) + 0.1 * r2.standard_normal(n2)
    for st in settings:
        saved = {}
        for kv in [kv for kv in st.split(",") if "=" in kv]:
            k, v = kv.split("=")
            saved[k] = ctx.get_param(k)
            ctx.set_param(k, int(v))

        def out(row, dt, ts, flops, **kw):
            print(json.dumps({"n": n, "row": row, "setting": st, "ms": dt * 1e3, "ms_all": [round(t * 1e3, 2) for t in ts], "frac": flops / dt / PK, **kw}), flush=True)

        if "mv" in rows:
            dt, ts = med3(lambda: post.mean_and_var(agp.RowVecs(xs)))
            m, v = post.mean_and_var(agp.RowVecs(xs[:256]))
            out("mean_and_var_4096", dt, ts, float(n) * n * 4096, var_min=float(v.min()), var_sum=float(v.sum()))
        if "cov" in rows:
            dt, ts = med3(lambda: post.cov(agp.RowVecs(xs[:1024])))
            out("cov_1024", dt, ts, float(n) * n * 1024 + float(n) * 1024 * 1024)
        if "upd" in rows:
            holder = {}

            def upd():
                p2 = agp.posterior(post(agp.RowVecs(x2), 0.01), y2)
                holder["lp"] = float(p2.logpdf_value)
                p2.data.C.free()

            dt, ts = med3(upd)
            out("sequential_update_8192", dt, ts, float(n) * n * n2 + float(n) * n2 * n2 + n2**3 / 3.0, logpdf=holder["lp"])
        if "grad" in rows:
            post.data.C.free()  # the gradient refits: give the three N×N workspaces room inside the cache cap
            res = {}

            def gr():
                res["lp"], res["g"] = agp.logpdf_and_grad(fx, y)

            if n >= 65536:
                gr()
            dt, ts = med3(gr)
            out("value_and_gradient", dt, ts, float(n) ** 3, logpdf=float(res["lp"]), g_scale=[float(v) for v in np.atleast_1d(res["g"]["scale"])], g_noise=float(res["g"]["noise"]))
            post = agp.posterior(fx, y)
        for k, v in saved.items():
            ctx.set_param(k, v)
    post.data.C.free()


if __name__ == "__main__":
    main()
