"""First fit on a FRESH multi-device context (virtual ranks, GPU_MAX_HW_QUEUES=16): failure counts per configuration of the
rank contexts (stream-K on/off) and of the comm stream (priority), with and without a small warm-up fit.
  python tools/multi_first_fit.py [reps] [variant indices, comma separated]   — run each call under `timeout`: one variant run of the
  stress tool hung with 24 streams on 16 hardware queues."""
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
VARIANTS = [("sk1 prio1", 1, "1", 0, {}), ("sk0 prio0 (default)", 0, "0", 0, {}), ("sk1 prio1 + warm-up fit", 1, "1", 1, {}),
            ("sk1 prio0", 1, "0", 0, {}), ("sk0 prio1", 0, "1", 0, {}),
            # localisation (next round): own copy kernel instead of hipMemcpy2DAsync; host syncs after exchange / panel / bulk
            ("sk0 prio1 copy_kernel", 0, "1", 0, {"copy_kernel": 1}), ("sk0 prio1 sync-after-exchange", 0, "1", 0, {"multi_debug_sync": 1}),
            ("sk0 prio1 sync-after-panel", 0, "1", 0, {"multi_debug_sync": 8}), ("sk0 prio1 sync-after-bulk", 0, "1", 0, {"multi_debug_sync": 16}),
            ("sk0 prio1 host-side event waits", 0, "1", 0, {"multi_debug_sync": 4})]
pick = sys.argv[2].split(",") if len(sys.argv) > 2 else None
for vi, (name, sk, prio, warm, extra) in enumerate(VARIANTS):
    if pick is not None and str(vi) not in pick:
        continue
    os.environ["GPMI_COMM_PRIO"] = prio
    bad = tot = 0
    for seed, kern, mean, X, y, sig, P, Q, nb, depth, lp_ref, alpha in cases:
        for i in range(reps):
            ctx = agp.Context(devices=[0] * (P * Q), P=P, Q=Q, nb=nb)
            ctx.set_param("lookahead_depth", depth)
            ctx.set_param("gemm_streamk", sk)
            for k, v in extra.items():
                ctx.set_param(k, v)
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
    print(f"[{vi}] {name:34s} wrong first fits: {bad}/{tot}", flush=True)
