"""Round 6: the pieces of the multi-device panel step, standalone on one idle device through the library's device-level entry points (the durations the cost model
tools/grid_model.py prices the critical cycle with): Cholesky of the nb×nb diagonal block (gpd_potrf, rank-context settings: 64-column leaves, no stream-K tails),
its inverse −inv(L_kk) level by level / by the recursion (gpd_inv_lower), and the rows-below solve of a rank's share (m rows) by one triangular-k GEMM with the
inverse (gpd_trsm_inv) against the substitution recursion (gpd_trsm).  HIP events on the ctx stream, median of 7 after a warm-up.  One JSON line per nb."""
import ctypes as C
import json
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import abstractgps_jl_amd as agp  # noqa: E402
from abstractgps_jl_amd._lib import check  # noqa: E402


def P(t):
    return C.c_void_p(t.data_ptr())


def main():
    st = torch.cuda.Stream()
    ctx = agp.Context(0, stream=st.cuda_stream)
    ctx.set_param("gemm_streamk", 0)
    ctx.set_param("leaf_cols", 64)
    lib, h = ctx.lib, ctx.handle
    info = torch.zeros(16, dtype=torch.int32, device="cuda")
    logdet = torch.zeros(128, dtype=torch.float64, device="cuda")

    def timed(fn, reset=None, reps=7):
        ts = []
        for i in range(reps + 1):
            if reset:
                reset()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            with torch.cuda.stream(st):
                e0.record(st)
                fn()
                e1.record(st)
            e1.synchronize()
            if i:
                ts.append(e0.elapsed_time(e1))
        return float(np.median(ts))

    for nb in (512, 1024, 2048):
        ld = nb + 32
        g = torch.Generator(device="cuda").manual_seed(nb)
        G = torch.randn(nb, 64, dtype=torch.float64, device="cuda", generator=g)
        A0 = torch.zeros(nb + 128, ld, dtype=torch.float64, device="cuda")
        A0[:nb, :nb] = G @ G.T / 64 + 2.0 * torch.eye(nb, dtype=torch.float64, device="cuda")
        A = A0.clone()
        W = torch.zeros(nb + 128, ld, dtype=torch.float64, device="cuda")
        S1 = torch.zeros(nb + 128, ld, dtype=torch.float64, device="cuda")
        S2 = torch.zeros(nb + 128, ld, dtype=torch.float64, device="cuda")

        def reset_a():
            with torch.cuda.stream(st):
                A.copy_(A0)

        out = {"nb": nb}
        out["potrf_diag_ms"] = timed(lambda: check(lib.gpd_potrf(h, P(A), ld, nb, nb, P(info), 0, nb, P(logdet))), reset_a)
        out["inv_levels_ms"] = timed(lambda: check(lib.gpd_inv_lower(h, P(A), ld, nb, P(W), ld, P(S1), P(S2))))
        out["inv_recursion_ms"] = timed(lambda: check(lib.gpd_inv_lower(h, P(A), ld, nb, P(W), ld, P(S1), None)))
        for m in (8192, 32768):
            X0 = torch.randn(m + 128, ld, dtype=torch.float64, device="cuda", generator=g)
            X = X0.clone()
            S = torch.zeros(m + 128, ld, dtype=torch.float64, device="cuda")

            def reset_x():
                with torch.cuda.stream(st):
                    X.copy_(X0)

            out[f"trsm_inv_ms_m{m}"] = timed(lambda: check(lib.gpd_trsm_inv(h, P(X), ld, m, P(W), ld, nb, P(S), ld)), reset_x)
            out[f"trsm_subst_ms_m{m}"] = timed(lambda: check(lib.gpd_trsm(h, P(X), ld, m, P(A), ld, nb)), reset_x)
            del X0, X, S
        print(json.dumps({k: (round(v, 4) if isinstance(v, float) else v) for k, v in out.items()}), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
