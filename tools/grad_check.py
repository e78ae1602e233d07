"""value + gradient of logpdf (gp_logpdf_grad) against central differences of gp_logpdf, COMPONENT by component, at a ladder of sizes up to C4
(round 6: the bench's new directional check passed at C2 and failed at C4).  One JSON line per size."""
import json
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import abstractgps_jl_amd as agp  # noqa: E402


def synth(n, d, seed):
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((n, d))
    return X, np.sin(X.sum(1)) + 0.1 * rng.standard_normal(n)


def main():
    sizes = [int(v) for v in sys.argv[1:]] or [16384, 32768, 49152, 65536]
    ctx = agp.default_context()
    for n in sizes:
        x, y = synth(n, 3, 4)

        def lp(var, sc, nz):
            return float(agp.logpdf(agp.GP(var * agp.SqExponentialKernel() @ agp.ScaleTransform(sc), ctx=ctx)(agp.RowVecs(x), nz), y))

        for dib in (2048, 0):
            ctx.set_param("dib_nb", dib)
            lp0, g = agp.logpdf_and_grad(agp.GP(agp.SqExponentialKernel() @ agp.ScaleTransform(1.0), ctx=ctx)(agp.RowVecs(x), 0.01), y)
            out = {"n": n, "dib_nb": dib, "logpdf": float(lp0), "grad": {k: float(g[k]) for k in ("variance", "scale", "noise")}}
            if dib:
                h = 1e-4
                out["logpdf_direct"] = lp(1.0, 1.0, 0.01)
                out["fd"] = {"variance": (lp(1 + h, 1.0, 0.01) - lp(1 - h, 1.0, 0.01)) / (2 * h),
                             "scale": (lp(1.0, 1 + h, 0.01) - lp(1.0, 1 - h, 0.01)) / (2 * h),
                             "noise": (lp(1.0, 1.0, 0.01 * (1 + h)) - lp(1.0, 1.0, 0.01 * (1 - h))) / (2 * h * 0.01)}
                out["rel"] = {k: abs(out["grad"][k] - out["fd"][k]) / abs(out["fd"][k]) for k in out["fd"]}
            print(json.dumps(out), flush=True)
            ctx.trim()
        ctx.set_param("dib_nb", 2048)


if __name__ == "__main__":
    main()
