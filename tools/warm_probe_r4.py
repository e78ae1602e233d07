"""Round 4: is the slow first trailing update of a fit (profiles/r4/gemm_dump.txt: the same launch shape runs 14 % slower at the start of a C2
fit than in the middle of a C3 fit) a clock / power-state effect?  Fits timed (a) back to back, (b) right after 20 ms of pure fp64 MFMA work,
(c) after 100 ms of idle; and the pure-MFMA rate of a ≈1 ms burst measured cold (after idle) and hot (after a long burst)."""
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
lib = ctx.lib


def mfma(iters):
    out = C.c_double()
    rc = lib.gp_bench_mfma_f64(ctx.handle, iters, C.byref(out))
    assert rc == 0
    return out.value


# pure-MFMA rate: short burst cold / hot
for rep in range(3):
    time.sleep(0.1)
    cold = [mfma(600) for _ in range(1)][0]
    mfma(12000)
    hot = mfma(600)
    seq = []
    time.sleep(0.1)
    for _ in range(12):
        seq.append(round(mfma(1200), 1))
    print(json.dumps({"probe": "mfma_f64_burst", "cold_tflops": cold, "hot_tflops": hot, "after_idle_sequence_of_2ms_bursts": seq}), flush=True)

for name, n, d, seed in (("N8192", 8192, 3, 8), ("C2", 16384, 3, 2)):
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((n, d)); y = np.sin(X.sum(1)) + 0.1 * rng.standard_normal(n)
    fx = agp.GP(agp.SqExponentialKernel(), ctx=ctx)(agp.RowVecs(X), 0.01)
    agp.posterior(fx, y).data.C.free()
    for mode in ("back_to_back", "after_mfma_20ms", "after_idle_100ms", "back_to_back"):
        ts = []
        for rep in range(6):
            if mode == "after_mfma_20ms":
                mfma(12000)
            elif mode == "after_idle_100ms":
                time.sleep(0.1)
            t0 = time.perf_counter()
            post = agp.posterior(fx, y)
            ts.append(time.perf_counter() - t0)
            post.data.C.free()
        print(json.dumps({"case": name, "mode": mode, "ms_min": min(ts[1:]) * 1e3, "ms_med": float(np.median(ts[1:])) * 1e3, "ms_all": [round(t * 1e3, 2) for t in ts]}), flush=True)
