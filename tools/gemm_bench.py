"""Isolated timing of the MFMA trailing-update kernel (gpd_gemm_nt) — used with rocprofv3 --pmc."""
import argparse
import ctypes as C
import json
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import abstractgps_jl_amd as agp  # noqa: E402
from abstractgps_jl_amd._lib import check, gp_grid  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", default="8192x8192x2048,16384x16384x2048,32768x2048x2048,8192x8192x64")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--lower", type=int, default=0)
    ap.add_argument("--lda", type=int, default=0, help="operand leading dimension (0: k + 32)")
    ap.add_argument("--params", default="", help="ctx parameters name=value,...")
    args = ap.parse_args()
    ctx = agp.Context(0)
    for kv in [kv for kv in args.params.split(",") if kv]:
        ctx.set_param(kv.split("=")[0], int(kv.split("=")[1]))
    lib, h = ctx.lib, ctx.handle
    c = C.c_double()
    check(lib.gp_bench_mfma_f64(h, 20000, C.byref(c)))
    print(json.dumps({"mfma_f64_ceiling_tflops": c.value}), flush=True)
    for shp in args.shapes.split(","):
        m, n, k = [int(v) for v in shp.split("x")]
        lda = args.lda or (k + 32)
        A = torch.randn(max(m, n) + 128, lda, dtype=torch.float64, device="cuda")
        Cm = torch.zeros(m + 128, n + 32, dtype=torch.float64, device="cuda")
        g = gp_grid(1, 0, 1, 0, 1, args.lower)
        P = lambda t: C.c_void_p(t.data_ptr())
        torch.cuda.synchronize()
        check(lib.gpd_gemm_nt(h, P(Cm), n + 32, P(A), lda, P(A), lda, m, n, k, C.byref(g), 0, 0))
        check(lib.gpd_sync(h))
        t0 = time.perf_counter()
        for _ in range(args.reps):
            check(lib.gpd_gemm_nt(h, P(Cm), n + 32, P(A), lda, P(A), lda, m, n, k, C.byref(g), 0, 0))
        check(lib.gpd_sync(h))
        dt = (time.perf_counter() - t0) / args.reps
        fl = 2.0 * m * n * k * (0.5 if args.lower else 1.0)
        print(json.dumps({"m": m, "n": n, "k": k, "lower": args.lower, "params": args.params, "ms": round(dt * 1e3, 4), "tflops": round(fl / dt / 1e12, 2)}), flush=True)


if __name__ == "__main__":
    main()
