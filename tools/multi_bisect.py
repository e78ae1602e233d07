"""Re-run failing seeds of tools/random_sweep2.py (exact part) with labelled checks; the parent loops over engine-parameter
variants (GPMI_PARAMS / GPMI_COMM_PRIO / GPU_MAX_HW_QUEUES) in subprocesses."""
import os
import subprocess
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))


def child(seeds, reps):
    import abstractgps_jl_amd as agp
    from oracle import gp_oracle as o
    sys.path.insert(0, str(ROOT / "tools"))
    import random_sweep2 as R

    out = []
    for seed in seeds:
        for rep in range(reps):
            rng = np.random.default_rng(seed)
            n = int(rng.choice([130, 257, 500, 777, 1024, 1300, 2049]))
            d = int(rng.integers(1, 5))
            kind, kern, mean, of = R.random_gp(rng, d)
            X = rng.standard_normal((n, d))
            y = np.sin(X.sum(1)) + 0.1 * rng.standard_normal(n)
            sig = float(rng.uniform(0.03, 0.3)) if rng.random() < 0.5 else rng.uniform(0.03, 0.3, n)
            P, Q = R.GRIDS[int(rng.integers(0, len(R.GRIDS)))]
            nb = int(rng.choice([128, 256]))
            depth = int(rng.integers(1, 4))
            ofx = o.FiniteGP(of, X, sig)
            lp_ref, opost = o.logpdf_and_posterior(ofx, y)
            res = []
            for grid in ((P, Q), (1, 1), None):
                try:
                    if grid is None:
                        ctx = agp.Context(0)
                    else:
                        ctx = agp.Context(devices=[0] * (grid[0] * grid[1]), P=grid[0], Q=grid[1], nb=nb)
                        ctx.set_param("lookahead_depth", depth)
                    f = agp.GP(kern, ctx=ctx) if mean is None else agp.GP(mean, kern, ctx=ctx)
                    post = agp.posterior(f(agp.RowVecs(X), sig), y)
                    e1 = abs(float(post.logpdf_value) - lp_ref) / abs(lp_ref)
                    e2 = R.rel(post.data.alpha, opost.alpha)
                    xs = rng.standard_normal((7, d))
                    m, v = post.mean_and_var(agp.RowVecs(xs))
                    mo, vo = opost.mean_and_var(xs)
                    res.append(f"{grid}: lp {e1:.1e} al {e2:.1e} m {np.max(np.abs(m - mo)):.1e} v {np.max(np.abs(v - vo)):.1e}")
                    ctx.close()
                except Exception as e:  # noqa: BLE001
                    res.append(f"{grid}: EXC {repr(e)[:80]}")
            print(f"seed {seed} rep {rep} n={n} d={d} kind={kind} nb={nb} depth={depth} | " + " | ".join(res), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child([int(s) for s in sys.argv[2].split(",")], int(sys.argv[3]))
        sys.exit(0)
    seeds = sys.argv[1] if len(sys.argv) > 1 else "9055,9069,9092,9097"
    variants = [("default", {}), ("hwq16", {"GPU_MAX_HW_QUEUES": "16"}),
                ("hwq16 prio0", {"GPU_MAX_HW_QUEUES": "16", "GPMI_COMM_PRIO": "0"}),
                ("hwq16 leaf64", {"GPU_MAX_HW_QUEUES": "16", "GPMI_PARAMS": "leaf_group=64"}),
                ("hwq16 sk0", {"GPU_MAX_HW_QUEUES": "16", "GPMI_PARAMS": "gemm_streamk=0"}),
                ("hwq16 trsv1024", {"GPU_MAX_HW_QUEUES": "16", "GPMI_PARAMS": "trsv_nb=1024"}),
                ("hwq16 old", {"GPU_MAX_HW_QUEUES": "16", "GPMI_COMM_PRIO": "0", "GPMI_PARAMS": "leaf_group=64,gemm_streamk=0,trsv_nb=1024"})]
    for name, env in variants:
        print("=== " + name, flush=True)
        e = dict(os.environ)
        e.update(env)
        r = subprocess.run([sys.executable, __file__, "child", seeds, "2"], env=e, capture_output=True, text=True, timeout=900)
        print(r.stdout[-6000:], flush=True)
        if r.returncode != 0:
            print("rc", r.returncode, r.stderr[-500:], flush=True)
