"""One exact fit (after a warm-up fit) at N points with ctx params from the command line (k=v ...), for rocprofv3 kernel
traces:  python tools/trace_fit.py 16384 nb=1024 lookahead=0"""
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import abstractgps_jl_amd as agp  # noqa: E402

n = int(sys.argv[1])
params = dict(a.split("=") for a in sys.argv[2:])
mfma_ref = int(params.pop("mfma_ref", 0))
rng = np.random.default_rng(2)
X = rng.standard_normal((n, 3))
y = np.sin(X.sum(axis=1)) + 0.1 * rng.standard_normal(n)
ctx = agp.default_context(0)
for k, v in params.items():
    ctx.set_param(k, float(v))
fx = agp.GP(agp.SqExponentialKernel())(agp.RowVecs(X), 0.01)
for rep in range(3):
    t0 = time.perf_counter()
    post = agp.posterior(fx, y)
    dt = time.perf_counter() - t0
    post.data.C.free()
    print(f"fit {rep}: {dt * 1e3:.3f} ms", flush=True)
if mfma_ref:  # the pure-MFMA reference kernel inside the same profiled process (normalises SQ_VALU_MFMA_BUSY_CYCLES / GRBM_GUI_ACTIVE)
    import ctypes as C

    v = C.c_double()
    ctx.lib.gp_bench_mfma_f64(ctx.handle, 20000, C.byref(v))
