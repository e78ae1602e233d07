"""Round 5: is the Gram kernel's 4.6 TB/s inside a fit the kernel, or where it sits in the fit?  gpd_assemble (the same kmat_kernel launch, lower tiles of
K + σ²I, N = 65 536, D = 3, SE) eight times back to back on a hot device, each bracketed by events, and once after 50 ms of idle — against the 3.7–3.9 ms
(4.5–4.6 TB/s) the same launch takes as the first kernel of a fit, right behind the previous fit's latency-bound tail and the host gap."""
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
from abstractgps_jl_amd._lib import check, gp_grid, gp_kernel  # noqa: E402

n, d = int(sys.argv[1]) if len(sys.argv) > 1 else 65536, 3
ctx = agp.default_context(0)
lib, h = ctx.lib, ctx.handle
rng = np.random.default_rng(4)
x = torch.tensor(rng.standard_normal((d, n)), dtype=torch.float64, device="cuda")  # dimension-major [d][n]
noise = torch.full((n,), 0.01, dtype=torch.float64, device="cuda")
ld = n + 32
A = torch.empty((n + 128) * ld, dtype=torch.float64, device="cuda")
k = gp_kernel(0, 0, 1.0, 0, None)
g = gp_grid(1, 0, 1, 0, 1, 1)
P = lambda t: C.c_void_p(t.data_ptr())
bytes_alg = 8.0 * n * (n + 1) / 2


def one():
    check(lib.gpd_assemble(h, C.byref(k), P(x), n, n, d, P(noise), C.byref(g), P(A), ld, n, n))


def timed(reps):
    ts = []
    for _ in range(reps):
        check(lib.gpd_sync(h))
        t0 = time.perf_counter()
        one()
        check(lib.gpd_sync(h))
        ts.append(time.perf_counter() - t0)
    return ts


one()
check(lib.gpd_sync(h))
hot = timed(8)
time.sleep(0.05)
cold = timed(1)
# keep the device busy with MFMA work right before (the clock a GEMM phase leaves behind)
v = C.c_double()
check(lib.gp_bench_mfma_f64(h, 20000, C.byref(v)))
after_mfma = timed(1)
print(json.dumps({"n": n, "hot_back_to_back_ms": [round(t * 1e3, 3) for t in hot], "hot_TBps": [round(bytes_alg / t / 1e12, 2) for t in hot],
                  "after_50ms_idle_ms": round(cold[0] * 1e3, 3), "after_50ms_idle_TBps": round(bytes_alg / cold[0] / 1e12, 2),
                  "after_mfma_burst_ms": round(after_mfma[0] * 1e3, 3), "after_mfma_burst_TBps": round(bytes_alg / after_mfma[0] / 1e12, 2),
                  "note": "host-timed launch + sync (adds ~20 us); algorithmic bytes 8 N (N + 1) / 2"}), flush=True)
