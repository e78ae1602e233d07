"""Round 4: per-launch rate of ONE trailing-update shape (8 192 x 8 192 lower, K = 2 048: 2 112 tiles, ≈ 2 ms) launched 14 times in a row, after
(a) 60 ms of the same GEMM (hot), (b) 100 ms idle, (c) 20 ms of pure register-only MFMA work (core clock hot, no memory traffic),
(d) ≈ 10 ms / (e) ≈ 2.6 ms of leaf chain (N = 2 048 fits).  Separates 'core clock' from 'memory side' in the slow first update of a fit."""
import ctypes as C
import json
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import abstractgps_jl_amd as agp  # noqa: E402
from abstractgps_jl_amd._lib import check, gp_grid  # noqa: E402

ctx = agp.default_context(0)
lib, h = ctx.lib, ctx.handle
m = n = 8192
k = 2048
lda = k + 32
A = torch.randn(m + 128, lda, dtype=torch.float64, device="cuda")
Cm = torch.zeros(m + 128, n + 32, dtype=torch.float64, device="cuda")
g = gp_grid(1, 0, 1, 0, 1, 1)
P = lambda t: C.c_void_p(t.data_ptr())
fl = 2.0 * m * n * k * 0.5 * (1 + 1.0 / 64)


def gemm():
    t0 = time.perf_counter()
    check(lib.gpd_gemm_nt(h, P(Cm), n + 32, P(A), lda, P(A), lda, m, n, k, C.byref(g), 0, 0))
    check(lib.gpd_sync(h))
    return round(fl / (time.perf_counter() - t0) / 1e12, 1)


def mfma(iters):
    out = C.c_double()
    check(lib.gp_bench_mfma_f64(h, iters, C.byref(out)))


rng = np.random.default_rng(1)
X = rng.standard_normal((2048, 3)); y = np.sin(X.sum(1))
fx = agp.GP(agp.SqExponentialKernel(), ctx=ctx)(agp.RowVecs(X), 0.01)
agp.posterior(fx, y).data.C.free()
torch.cuda.synchronize()
for rep in range(2):
    for mode in ("hot", "idle_100ms", "after_pure_mfma_20ms", "after_leaf_chain_10ms", "after_leaf_chain_2.6ms"):
        for _ in range(30):
            gemm()
        if mode == "idle_100ms":
            time.sleep(0.1)
        elif mode == "after_pure_mfma_20ms":
            mfma(12000)
        elif mode.startswith("after_leaf_chain"):
            for _ in range(8 if mode.endswith("10ms") else 2):
                agp.posterior(fx, y).data.C.free()
        print(json.dumps({"mode": mode, "tflops_per_launch": [gemm() for _ in range(14)]}), flush=True)
