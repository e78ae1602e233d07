"""Stage-by-stage errors of the passes over the distributed factor (C \\ B, sequential update ×2, predictions, gather) against the
oracle for a few virtual-rank grids — prints every number instead of stopping at the first assertion (debug aid for
tests/test_gpu_multi.py::test_sequential_update_and_solve_on_the_pieces)."""
import sys
import traceback
from pathlib import Path

import numpy as np
import scipy.linalg as sla

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import abstractgps_jl_amd as agp  # noqa: E402
from oracle import gp_oracle as o  # noqa: E402


def rel(a, b):
    return float(np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(b), 1e-300))


def run(P, Q, n1=1100, n2=300, n3=41, d=3, nb=128):
    n = n1 + n2 + n3
    x, y = o.synth_inputs(n, d, 90 + P * 10 + Q)
    rng = np.random.default_rng(P * 11 + Q)
    s2 = 0.04 + 0.05 * rng.random(n)
    of = o.GP(o.Kernel(o.MATERN52, 1.3, 0.8), 0.2)
    ob1 = o.posterior(o.FiniteGP(of, x[:n1], s2[:n1]), y[:n1])
    ob2 = o.posterior(o.FiniteGP(of, x[: n1 + n2], s2[: n1 + n2]), y[: n1 + n2])
    ob3 = o.posterior(o.FiniteGP(of, x, s2), y)
    xs = rng.standard_normal((200, d)) * 1.2
    ctx = agp.Context(devices=[0] * (P * Q), P=P, Q=Q, nb=nb)
    out = {"grid": f"{P}x{Q}"}

    def stage(name, fn):
        try:
            out[name] = fn()
        except Exception as e:  # noqa: BLE001
            out[name] = "ERR " + str(e)[:160]
            traceback.print_exc(limit=1)

    try:
        f = agp.GP(0.2, 1.3 * agp.Matern52Kernel() @ agp.ScaleTransform(0.8), ctx=ctx)
        p1 = agp.posterior(f(agp.RowVecs(x[:n1]), s2[:n1]), y[:n1])
        out["fit_alpha"] = rel(p1.data.alpha, ob1.alpha)
        B = rng.standard_normal((n1, 3))
        stage("solve3", lambda: rel(p1.data.C.solve(B), sla.cho_solve((ob1.U, False), B)))
        stage("solve1", lambda: rel(p1.data.C.solve(B[:, 0]), sla.cho_solve((ob1.U, False), B[:, 0])))
        st = {}

        def upd2():
            st["p2"] = agp.posterior(p1(agp.RowVecs(x[n1 : n1 + n2]), s2[n1 : n1 + n2]), y[n1 : n1 + n2])
            return rel(st["p2"].data.alpha, ob2.alpha)

        stage("upd2_alpha", upd2)
        if "p2" in st:
            p2 = st["p2"]
            stage("upd2_alpha_new_rows", lambda: rel(p2.data.alpha[n1:], ob2.alpha[n1:]))
            stage("upd2_logpdf", lambda: abs(float(p2.logpdf_value) - float(o.logpdf(o.FiniteGP(of, x[: n1 + n2], s2[: n1 + n2]), y[: n1 + n2]))))
            stage("upd2_var", lambda: float(np.max(np.abs(p2.var(agp.RowVecs(xs)) - ob2.mean_and_var(xs)[1]))))
            stage("upd2_mean", lambda: float(np.max(np.abs(p2.mean(agp.RowVecs(xs)) - ob2.mean_and_var(xs)[0]))))

            def upd3():
                st["p3"] = agp.posterior(p2(agp.RowVecs(x[n1 + n2 :]), s2[n1 + n2 :]), y[n1 + n2 :])
                return rel(st["p3"].data.alpha, ob3.alpha)

            stage("upd3_alpha", upd3)
            if "p3" in st:
                p3 = st["p3"]
                stage("upd3_var", lambda: float(np.max(np.abs(p3.var(agp.RowVecs(xs)) - ob3.mean_and_var(xs)[1]))))
                B3 = rng.standard_normal(n)
                stage("upd3_solve", lambda: rel(p3.data.C.solve(B3), sla.cho_solve((ob3.U, False), B3)))
                stage("stats", lambda: ctx.multi_stats())
                stage("upd3_U", lambda: float(np.max(np.abs(p3.data.C.U - ob3.U))))
            stage("upd2_U", lambda: float(np.max(np.abs(p2.data.C.U - ob2.U))))
        stage("p1_var_after", lambda: float(np.max(np.abs(p1.var(agp.RowVecs(xs[:50])) - ob1.mean_and_var(xs[:50])[1]))))
    finally:
        ctx.close()
    print(out, flush=True)


if __name__ == "__main__":
    grids = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]] or [(1, 1), (2, 2), (2, 3)]
    for P, Q in grids:
        run(P, Q)
