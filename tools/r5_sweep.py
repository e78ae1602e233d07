"""Round 5 A/B harness on one MI355X: exact pairs (N, kernel of the BASELINE config), C5 (VFE fp32) and isolated fp64 GEMM launches under
lists of ctx-parameter settings.  One JSON line per (case, setting); settings are applied to ONE context and reset to the documented
defaults (include/gpmi355.h GPMI355_PARAM_DEFAULTS) between settings.

  python tools/r5_sweep.py pair:4096,16384,32768,65536 c5 gemm -- "base" "gemm_pipe=0" "gemm_pipe=0,gemm_pad_f32=0"
"""
import ctypes as C
import json
import re
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import abstractgps_jl_amd as agp  # noqa: E402
from abstractgps_jl_amd._lib import check  # noqa: E402


def defaults() -> dict:
    txt = (ROOT / "include" / "gpmi355.h").read_text()
    m = re.search(r"#define GPMI355_PARAM_DEFAULTS(.*?)\nint32_t gp_ctx_set_param", txt, flags=re.S)
    body = "".join(re.findall(r'"([^"]*)"', m.group(1)))
    return {kv.split("=")[0]: int(kv.split("=")[1]) for kv in body.split(",") if kv}


DEF = defaults()
ctx = agp.default_context(0)


def apply(setting: str):
    for k, v in DEF.items():
        if k not in ("gemm_pad_lds", "pool_cap_mb"):
            ctx.set_param(k, v)
    for kv in [kv for kv in setting.split(",") if "=" in kv]:
        ctx.set_param(kv.split("=")[0], int(kv.split("=")[1]))


def synth(n, d, seed):
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((n, d))
    return X, np.sin(X.sum(1)) + 0.1 * rng.standard_normal(n)


def pair(n, setting, reps):
    d, seed, kern = (8, 3, agp.Matern32Kernel() @ agp.ScaleTransform(0.5)) if n == 32768 else (3, {16384: 2, 65536: 4}.get(n, 9), agp.SqExponentialKernel())
    x, y = synth(n, d, seed)
    fx = agp.GP(kern, ctx=ctx)(agp.RowVecs(x), 0.01)
    ts, lp = [], None
    for _ in range(reps + 1):
        t0 = time.perf_counter()
        p = agp.posterior(fx, y)
        ts.append(time.perf_counter() - t0)
        lp = float(p.logpdf_value)
        tm = ctx.timings()
        p.data.C.free()
    med = float(np.median(ts[1:]))
    print(json.dumps({"case": f"pair N={n}", "setting": setting, "ms_med": med * 1e3, "ms_min": min(ts[1:]) * 1e3, "frac": (n**3 / 3 + 3 * n**2) / med / 78.6e12,
                      "assemble_ms": tm["assemble_ms"], "potrf_ms": tm["potrf_ms"], "solve_ms": tm["solve_ms"], "logpdf": lp}), flush=True)


_c5 = {}


def c5(setting, reps):
    if not _c5:
        rng = np.random.default_rng(5)
        n, m, d = 262144, 4096, 3
        X = (rng.uniform(0, 1, (n, d)) * 4).astype(np.float32)
        y = (np.sin(X.sum(1)) + 0.3 * rng.standard_normal(n)).astype(np.float32)
        z = X[rng.permutation(n)[:m]].copy()
        f = agp.GP(agp.SqExponentialKernel(), ctx=ctx)
        _c5.update(n=n, m=m, fx=f(agp.RowVecs(X), np.float32(0.1)), approx=agp.VFE(f(agp.RowVecs(z), 1e-4)), y=y)
    n, m = _c5["n"], _c5["m"]
    ts, obj = [], None
    for _ in range(reps + 1):
        t0 = time.perf_counter()
        p = agp.posterior(_c5["approx"], _c5["fx"], _c5["y"])
        obj = float(p.objective)
        ts.append(time.perf_counter() - t0)
        del p
    med = float(np.median(ts[1:]))
    flops = 2.0 * n * m * m + 2.0 * m**3 / 3
    print(json.dumps({"case": "C5", "setting": setting, "ms_med": med * 1e3, "ms_min": min(ts[1:]) * 1e3, "frac_fp32": flops / med / 157.3e12, "elbo": obj}), flush=True)


def gemm(setting, reps):
    import torch

    lib, h = ctx.lib, ctx.handle
    from abstractgps_jl_amd._lib import gp_grid

    for (m, n, k, lower) in ((8192, 8192, 2048, 1), (16384, 16384, 2048, 1), (49152, 49152, 2048, 1), (32768, 2048, 2048, 0), (30720, 1024, 1024, 0), (16384, 512, 512, 0)):
        lda = k + 32
        A = torch.randn(max(m, n) + 128, lda, dtype=torch.float64, device="cuda")
        Cm = torch.zeros(m + 128, n + 32, dtype=torch.float64, device="cuda")
        g = gp_grid(1, 0, 1, 0, 1, lower)
        P = lambda t: C.c_void_p(t.data_ptr())
        for _ in range(2):
            check(lib.gpd_gemm_nt(h, P(Cm), n + 32, P(A), lda, P(A), lda, m, n, k, C.byref(g), 0, 0))
        check(lib.gpd_sync(h))
        t0 = time.perf_counter()
        for _ in range(reps):
            check(lib.gpd_gemm_nt(h, P(Cm), n + 32, P(A), lda, P(A), lda, m, n, k, C.byref(g), 0, 0))
        check(lib.gpd_sync(h))
        dt = (time.perf_counter() - t0) / reps
        fl = 2.0 * m * n * k * (0.5 if lower else 1.0)
        print(json.dumps({"case": f"gemm {m}x{n}x{k} lower={lower}", "setting": setting, "ms": dt * 1e3, "tflops": fl / dt / 1e12}), flush=True)
        del A, Cm
    torch.cuda.empty_cache()


if __name__ == "__main__":
    args = sys.argv[1:]
    cut = args.index("--") if "--" in args else len(args)
    cases, settings = args[:cut], (args[cut + 1:] or ["base"])
    c = C.c_double()
    check(ctx.lib.gp_bench_mfma_f64(ctx.handle, 20000, C.byref(c)))
    c32 = C.c_double()
    check(ctx.lib.gp_bench_mfma_f32(ctx.handle, 0, 20000, C.byref(c32)))
    print(json.dumps({"mfma_f64_ceiling_tflops": c.value, "mfma_f32_ceiling_tflops": c32.value}), flush=True)
    for case in cases:
        for st in settings:
            apply(st)
            if case.startswith("pair:"):
                for n in [int(v) for v in case[5:].split(",")]:
                    pair(n, st, 3 if n >= 65536 else 5)
                    if n >= 32768:
                        ctx.trim()
            elif case == "c5":
                c5(st, 5)
            elif case == "gemm":
                gemm(st, 5)
    apply("base")
