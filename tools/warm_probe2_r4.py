"""Round 4: does a short LOW-ACTIVITY phase (a latency-bound leaf chain: one workgroup busy) put the GPU into a state from which the fp64 MFMA
rate needs ≈ 20 ms to recover?  Pure-MFMA bursts of ≈ 2 ms (gp_bench_mfma_f64) measured right after (a) 20 ms of MFMA work, (b) the same followed
by two N = 2 048 fits (≈ 2 ms of leaf chain, nothing else), (c) the same followed by 2 ms of host sleep, (d) 100 ms of idle."""
import ctypes as C
import json
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import abstractgps_jl_amd as agp  # noqa: E402

ctx = agp.default_context(0)


def mfma(iters):
    out = C.c_double()
    assert ctx.lib.gp_bench_mfma_f64(ctx.handle, iters, C.byref(out)) == 0
    return round(out.value, 1)


rng = np.random.default_rng(1)
n = 2048
X = rng.standard_normal((n, 3)); y = np.sin(X.sum(1)) + 0.1 * rng.standard_normal(n)
fx = agp.GP(agp.SqExponentialKernel(), ctx=ctx)(agp.RowVecs(X), 0.01)
agp.posterior(fx, y).data.C.free()
t0 = time.perf_counter(); agp.posterior(fx, y).data.C.free(); small_ms = (time.perf_counter() - t0) * 1e3
for rep in range(3):
    for mode in ("hot", "hot_then_leaf_chain_2ms", "hot_then_leaf_chain_8ms", "hot_then_sleep_2ms", "idle_100ms"):
        mfma(12000)
        if mode.startswith("hot_then_leaf_chain"):
            for _ in range(2 if mode.endswith("2ms") else 8):
                agp.posterior(fx, y).data.C.free()
        elif mode == "hot_then_sleep_2ms":
            time.sleep(0.002)
        elif mode == "idle_100ms":
            time.sleep(0.1)
        seq = [mfma(1200) for _ in range(8)]
        print(json.dumps({"mode": mode, "small_fit_ms": round(small_ms, 2), "tflops_of_2ms_bursts": seq}), flush=True)
