"""Round 5: alternating A/B of the two Gram-kernel forms INSIDE fits on one box (ctx parameter "kmat_rows": 1 = row by row, 0 = accumulate form): five
fits each, interleaved, the assemble phase of every fit from the library's phase events.  profiles/r5/kmat_ab.jsonl; the selection rule in
csrc/kernels.hpp launch_kmat comes from these numbers."""
import json, sys, time
import numpy as np
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import abstractgps_jl_amd as agp
ctx = agp.default_context(0)
for n, d, kern in ((65536, 8, agp.Matern32Kernel() @ agp.ScaleTransform(0.5)), (49152, 3, agp.SqExponentialKernel()), (32768, 3, agp.SqExponentialKernel()), (8192, 3, agp.SqExponentialKernel())):
    rng = np.random.default_rng(4)
    x = rng.standard_normal((n, d)); y = np.sin(x.sum(1)) + 0.1 * rng.standard_normal(n)
    fx = agp.GP(kern, ctx=ctx)(agp.RowVecs(x), 0.01)
    agp.posterior(fx, y).data.C.free()
    res = {0: [], 1: []}
    for rep in range(5):
        for rows in (1, 0):
            ctx.set_param("kmat_rows", rows)
            agp.posterior(fx, y).data.C.free()
            res[rows].append(round(ctx.timings()["assemble_ms"], 3))
    ctx.set_param("kmat_rows", 1)
    print(json.dumps({"n": n, "d": d, "assemble_ms_rows1": res[1], "assemble_ms_rows0": res[0], "median_rows1": float(np.median(res[1])), "median_rows0": float(np.median(res[0]))}), flush=True)
    ctx.trim()
