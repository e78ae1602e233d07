"""First fit on a FRESH multi-device context (virtual ranks, GPU_MAX_HW_QUEUES=16): failure counts per configuration of the
rank contexts (stream-K on/off) and of the comm stream (priority), with and without a small warm-up fit."""
import os
import sys
from pathlib import Path

import numpy as np

os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tools"))
import abstractgps_jl_amd as agp  # noqa: E402
from oracle import gp_oracle as o  # noqa: E402
import random_sweep2 as R  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
cases = []
for seed in (9069, 9092):
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
    lp_ref, opost = o.logpdf_and_posterior(o.FiniteGP(of, X, sig), y)
    cases.append((seed, kern, mean, X, y, sig, P, Q, nb, depth, lp_ref, opost.alpha))
for name, sk, prio, warm in (("sk1 prio1", 1, "1", 0), ("sk0 prio0 (default)", 0, "0", 0), ("sk1 prio1 + warm-up fit", 1, "1", 1),
                             ("sk1 prio0", 1, "0", 0), ("sk0 prio1", 0, "1", 0)):
    os.environ["GPMI_COMM_PRIO"] = prio
    bad = tot = 0
    for seed, kern, mean, X, y, sig, P, Q, nb, depth, lp_ref, alpha in cases:
        for i in range(reps):
            ctx = agp.Context(devices=[0] * (P * Q), P=P, Q=Q, nb=nb)
            ctx.set_param("lookahead_depth", depth)
            ctx.set_param("gemm_streamk", sk)
            f = agp.GP(kern, ctx=ctx) if mean is None else agp.GP(mean, kern, ctx=ctx)
            try:
                if warm:
                    agp.posterior(f(agp.RowVecs(X[:300]), sig if np.ndim(sig) == 0 else sig[:300]), y[:300])
                post = agp.posterior(f(agp.RowVecs(X), sig), y)
                ok = abs(float(post.logpdf_value) - lp_ref) <= 1e-10 * abs(lp_ref) and R.rel(post.data.alpha, alpha) <= 1e-8
            except Exception:  # noqa: BLE001
                ok = False
            tot += 1
            bad += not ok
            ctx.close()
    print(f"{name:28s} wrong first fits: {bad}/{tot}", flush=True)
