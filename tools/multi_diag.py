"""Diagnostic: repeat multi-device (virtual-rank) fits and report errors per grid / depth / call index."""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import abstractgps_jl_amd as agp  # noqa: E402
from oracle import gp_oracle as o  # noqa: E402

n, d, nb = 1500, 3, 128
x, y = o.synth_inputs(n, d, 41)
of = o.GP(o.Kernel(o.SE, 1.0, 0.8))
lp_ref, opost = o.logpdf_and_posterior(o.FiniteGP(of, x, 0.05), y)
Y = np.stack([y, np.cos(y), 0.3 * y], axis=1)
lpY = o.logpdf(o.FiniteGP(of, x, 0.05), Y)
import os
grids = [tuple(int(v) for v in g.split('x')) for g in os.environ.get('DIAG_GRIDS', '2x2,4x2,2x4,2x3,4x1,1x4,8x1,3x1,4x4').split(',')]
extra = {k: int(v) for k, v in (kv.split("=") for kv in os.environ.get("DIAG_PARAMS", "").split(",") if kv)}
NIT = int(os.environ.get("DIAG_ITERS", "6"))
print("params", extra, flush=True)
bad = 0
for depth in (2, 1, 3):
    for P, Q in grids:
        ctx = agp.Context(devices=[0] * (P * Q), P=P, Q=Q, nb=nb)
        ctx.set_param("lookahead_depth", depth)
        for kk, vv in extra.items():
            ctx.set_param(kk, vv)
        f = agp.GP(agp.SqExponentialKernel() @ agp.ScaleTransform(0.8), ctx=ctx)
        fx = f(agp.RowVecs(x), 0.05)
        res = []
        for it in range(NIT):
            try:
                if it % 3 == 0:
                    v = abs(agp.logpdf(fx, y) - lp_ref) / abs(lp_ref)
                elif it % 3 == 1:
                    v = float(np.max(np.abs(agp.logpdf(fx, Y) - lpY) / np.abs(lpY)))
                else:
                    post = agp.posterior(fx, y)
                    v = float(np.linalg.norm(post.data.alpha - opost.alpha) / np.linalg.norm(opost.alpha))
                res.append(f"{v:.1e}" if v > 1e-10 else ".")
            except Exception as e:  # noqa: BLE001
                res.append(type(e).__name__ + ":" + str(getattr(e, "info", "")))
        print(f"depth={depth} grid={P}x{Q}: {res}", flush=True)
        bad += sum(r != "." for r in res)
        ctx.close()
print('FAILURES', bad)
sys.exit(1 if bad else 0)
