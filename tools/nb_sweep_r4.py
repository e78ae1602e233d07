"""Round 4 (panel width / CU partition after the leaf change): pair time of C2 / C3 / N = 4 096 / 8 192 with the round-3 leaf (leaf_v2=0) and the register-resident leaf (leaf_v2=1, XR auto / 64 / 128),
value-checked against each other (same inputs): one JSON line per setting.
(profiles/r4/nb_sweep.jsonl also holds three "split*" settings: the CU-partitioned schedule, measured one last time with this script before it was removed.)"""
import json
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import abstractgps_jl_amd as agp  # noqa: E402


def synth(n, d, seed):
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((n, d))
    return X, np.sin(X.sum(1)) + 0.1 * rng.standard_normal(n)


ctx = agp.default_context(0)
cases = [("N4096", 4096, 3, 9, agp.SqExponentialKernel()), ("N8192", 8192, 3, 8, agp.SqExponentialKernel()),
         ("C2", 16384, 3, 2, agp.SqExponentialKernel()), ("C3", 32768, 8, 3, agp.Matern32Kernel() @ agp.ScaleTransform(0.5))]
if len(sys.argv) > 1:
    cases = [c for c in cases if c[0] in sys.argv[1:]]
for name, n, d, seed, kern in cases:
    x, y = synth(n, d, seed)
    fx = agp.GP(kern, ctx=ctx)(agp.RowVecs(x), 0.01)
    ref = None
    for tag, params in (("default", {}), ("gemm_only", {"updk_max_k": 0, "upd128": 0}), ("k128_only", {"updk_max_k": 0}), ("k256", {"updk_max_k": 256}),
                        ("k512_all_m", {"updk_tall_m": 1 << 30}), ("rt4", {"updk_rt": 4}), ("rec", {"nb": 0})):
        ctx.set_param("updk_max_k", 512)
        ctx.set_param("updk_tall_m", 8192)
        ctx.set_param("updk_rt", 0)
        ctx.set_param("lookahead_min_n", 24576)
        ctx.set_param("upd128", 1)
        ctx.set_param("leaf_group", 128)
        ctx.set_param("nb", 2048)
        ctx.set_param("lookahead", 1)
        for k, v in params.items():
            ctx.set_param(k, v)
        ts = []
        for rep in range(6):
            t0 = time.perf_counter()
            post = agp.posterior(fx, y)
            ts.append(time.perf_counter() - t0)
            lp, al = float(post.logpdf_value), np.array(post.data.alpha)
            post.data.C.free()
        if ref is None:
            ref = (lp, al)
        tm = ctx.timings()
        print(json.dumps({"case": name, "n": n, "setting": tag, "ms_min": min(ts[1:]) * 1e3, "ms_med": float(np.median(ts[1:])) * 1e3,
                          "potrf_ms": tm["potrf_ms"], "solve_ms": tm["solve_ms"], "tflops": (n**3 / 3 + 3 * n**2) / min(ts[1:]) / 1e12,
                          "logpdf_rel_vs_default": abs(lp - ref[0]) / abs(ref[0]), "alpha_rel_vs_default": float(np.linalg.norm(al - ref[1]) / np.linalg.norm(ref[1]))}), flush=True)
ctx.set_param("leaf_group", 128)
ctx.set_param("nb", 2048)
ctx.set_param("leaf_v2", 1)
ctx.set_param("leaf_xr", 0)
ctx.set_param("leaf_cols", 128)
ctx.set_param("lookahead", 1)
ctx.set_param("lookahead_min_n", 24576)
ctx.set_param("upd128", 1)
ctx.set_param("updk_max_k", 512)
ctx.set_param("updk_tall_m", 8192)
ctx.set_param("updk_rt", 0)
