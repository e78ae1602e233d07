"""Per-launch GEMM records of one C2 / N fit (time_kernels = 1, GPMI_DUMP_GEMM): grouped by K and by size class."""
import os, sys, re, subprocess, json
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, str(ROOT))
    import numpy as np
    import abstractgps_jl_amd as agp
    n = int(sys.argv[2])
    rng = np.random.default_rng(2); X = rng.standard_normal((n, 3)); y = np.sin(X.sum(1)) + 0.1 * rng.standard_normal(n)
    ctx = agp.default_context(0)
    fx = agp.GP(agp.SqExponentialKernel(), ctx=ctx)(agp.RowVecs(X), 0.01)
    agp.posterior(fx, y).data.C.free()
    ctx.set_param("time_kernels", 1)
    print("BEGIN", file=sys.stderr, flush=True)
    agp.posterior(fx, y).data.C.free()
    sys.exit(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
r = subprocess.run([sys.executable, __file__, "child", str(n)], env=dict(os.environ, GPMI_DUMP_GEMM="1"), capture_output=True, text=True)
lines = r.stderr.split("BEGIN")[-1].splitlines()
recs = [tuple(float(v) for v in re.findall(r"M=(\d+) N=(\d+) K=(\d+) ms=([\d.]+) tflops=([\d.]+)", l)[0]) for l in lines if l.startswith("GEMM")]
tot_ms = sum(r[3] for r in recs); tot_fl = sum(r[3] * r[4] for r in recs)
print(f"N={n}: {len(recs)} launches, {tot_ms:.2f} ms, {tot_fl / tot_ms:.1f} TF/s average")
byk = {}
for M, N, K, ms, tf in recs:
    d = byk.setdefault(int(K), [0, 0.0, 0.0]); d[0] += 1; d[1] += ms; d[2] += ms * tf
for K in sorted(byk):
    c, ms, fl = byk[K]
    print(f"  K={K:5d}: {c:4d} launches {ms:7.3f} ms  {fl / ms:6.1f} TF/s   ideal at 64 TF/s {fl / 64:7.3f} ms  loss {ms - fl / 64:6.3f} ms")
print("  K = 2048 launches one by one (M, N, tiles on / below the diagonal, ms, TF/s):")
for M, N, K, ms, tf in recs:
    if int(K) >= 2048:
        tiles = ms * tf * 1e9 / (2 * 128 * 128 * K)
        print(f"    M={int(M):6d} N={int(N):6d} K={int(K):5d} tiles={tiles:7.0f} rounds={tiles / 512:5.2f} ms={ms:7.3f} TF/s={tf:5.1f}")
