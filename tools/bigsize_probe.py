"""One-off beyond the BASELINE sizes: N = 131 072 (a 137 GB factor, 1.7e10 matrix elements) — fit, the normal equations through the Gram-row kernel and on the host,
predictive variances in range, then a second fit (the first one's block comes back from the cache: round 6 lets ONE block above pool_cap_mb stay cached).  profiles/r6/n131072_properties.json."""
import sys, time, json
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import abstractgps_jl_amd as agp
from oracle import gp_oracle as o
n, d, s2 = 131072, 3, 0.01
x, y = o.synth_inputs(n, d, 8)
t0 = time.perf_counter()
post = agp.posterior(agp.GP(agp.SqExponentialKernel())(agp.RowVecs(x), s2), y)
t1 = time.perf_counter() - t0
alpha = np.array(post.data.alpha)
idx = np.linspace(0, n - 1, 512).astype(int)
m_tr, v_tr = post.mean_and_var(agp.RowVecs(x[idx]))
res_dev = float(np.max(np.abs(m_tr - (y[idx] - s2 * alpha[idx]))))
K = o.kernelmatrix(o.Kernel(o.SE), x[idx[-8:]], x)
res_host = float(np.max(np.abs(K @ alpha + s2 * alpha[idx[-8:]] - y[idx[-8:]])))
_, v_far = post.mean_and_var(agp.RowVecs(x[:64] + 3.0))
out = {"n": n, "fit_s": t1, "tflops": (n**3 / 3 + 3 * n * n) / t1 / 1e12, "logpdf": float(post.logpdf_value), "alpha_finite": bool(np.all(np.isfinite(alpha))),
       "normal_equations_residual_512_rows_device": res_dev, "normal_equations_residual_8_rows_host": res_host,
       "var_at_data_min_max": [float(v_tr.min()), float(v_tr.max())], "var_far_min_max": [float(v_far.min()), float(v_far.max())]}
post.data.C.free()
t0 = time.perf_counter()
post = agp.posterior(agp.GP(agp.SqExponentialKernel())(agp.RowVecs(x), s2), y)
out["fit_s_warm"] = time.perf_counter() - t0
out["tflops_warm"] = (n**3 / 3 + 3 * n * n) / out["fit_s_warm"] / 1e12
out["frac_warm"] = out["tflops_warm"] / 78.6
print(json.dumps(out))
