"""GPU: the IN-LIBRARY multi-device driver (gp_ctx_create_multi, csrc/multi.hip) with virtual ranks — P×Q ranks sharing the one
GPU of the test box, same schedule / kernels / block-cyclic predicate as on a node, same-device copies instead of xGMI
transfers (SURVEY.md §8(e) "virtual-rank mode").  Every grid is compared with the CPU oracle at the single-GPU tolerances
(logpdf rel <= 1e-10, α rel <= 1e-8, predictive mean abs <= 1e-8, var abs <= 1e-9), and everything downstream of a fit runs
on the gathered factor."""
import numpy as np
import pytest

from oracle import gp_oracle as o
from tests.conftest import rank_devices

pytestmark = pytest.mark.gpu


def _relnorm(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


GRIDS = [(1, 1), (2, 1), (1, 2), (2, 2), (4, 1), (1, 4), (4, 2), (2, 4), (8, 1), (3, 1), (2, 3)]


@pytest.mark.parametrize("P,Q", GRIDS, ids=lambda v: str(v))
def test_virtual_ranks_vs_oracle(agp, P, Q):
    n, d, nb = 1500, 3, 128
    x, y = o.synth_inputs(n, d, 40 + P * 10 + Q)
    rng = np.random.default_rng(P * 100 + Q)
    s2 = 0.02 + 0.05 * rng.random(n)
    of = o.GP(o.Kernel(o.MATERN52, 1.4, 0.8), 0.25)
    ofx = o.FiniteGP(of, x, s2)
    lp_ref, opost = o.logpdf_and_posterior(ofx, y)
    ctx = agp.Context(devices=rank_devices(P * Q), P=P, Q=Q, nb=nb)
    try:
        info = ctx.multi_info()
        assert (info["P"], info["Q"], info["nb"]) == (P, Q, nb)
        assert info["comm"] == "copies" or len(set(rank_devices(P * Q))) > 1   # (virtual ranks copy; real devices may use RCCL)
        f = agp.GP(0.25, 1.4 * agp.Matern52Kernel() @ agp.ScaleTransform(0.8), ctx=ctx)
        fx = f(agp.RowVecs(x), s2)
        assert agp.logpdf(fx, y) == pytest.approx(lp_ref, rel=1e-10)
        Y = np.stack([y, np.cos(y), 0.3 * y], axis=1)                       # matrix Y: three RHS rows ride along
        np.testing.assert_allclose(agp.logpdf(fx, Y), o.logpdf(ofx, Y), rtol=1e-10)
        post = agp.posterior(fx, y)
        assert float(post.logpdf_value) == pytest.approx(lp_ref, rel=1e-10)
        assert _relnorm(post.data.alpha, opost.alpha) <= 1e-8
        xs = x[:70] + 0.04
        np.testing.assert_allclose(post.mean(agp.RowVecs(xs)), opost.mean(xs), atol=1e-8)      # α only: no gather needed
        m, v = post.mean_and_var(agp.RowVecs(xs))                                              # gathers the factor onto device 0
        mo, vo = opost.mean_and_var(xs)
        np.testing.assert_allclose(m, mo, atol=1e-8)
        np.testing.assert_allclose(v, vo, atol=1e-9)
        assert np.max(np.abs(post.data.C.U - opost.U)) <= 1e-10                                # C.U of the reference
        np.testing.assert_allclose(post.cov(agp.RowVecs(xs)), opost.cov(xs), atol=1e-9)
        ys = rng.standard_normal(70)
        assert agp.logpdf(post(agp.RowVecs(xs), 0.1), ys) == pytest.approx(float(o.logpdf(o.FiniteGP(opost, xs, 0.1), ys)), rel=1e-9)
    finally:
        ctx.close()


@pytest.mark.parametrize("depth", [1, 2, 3])
@pytest.mark.parametrize("P,Q,n,nb", [(2, 2, 2300, 256), (4, 1, 1111, 128), (1, 3, 1000, 128)])
def test_lookahead_depths_and_ragged_sizes(agp, depth, P, Q, n, nb):
    x, y = o.synth_inputs(n, 2, 7 + depth)
    of = o.GP(o.Kernel(o.SE, 1.0, 1.3))
    lp_ref, opost = o.logpdf_and_posterior(o.FiniteGP(of, x, 0.05), y)
    ctx = agp.Context(devices=rank_devices(P * Q), P=P, Q=Q, nb=nb)
    try:
        ctx.set_param("lookahead_depth", depth)
        assert ctx.multi_info()["lookahead_depth"] == depth
        f = agp.GP(agp.SqExponentialKernel() @ agp.ScaleTransform(1.3), ctx=ctx)
        for _ in range(2):                                     # twice: the cross-rank event generations advance
            post = agp.posterior(f(agp.RowVecs(x), 0.05), y)
            assert float(post.logpdf_value) == pytest.approx(lp_ref, rel=1e-10)
            assert _relnorm(post.data.alpha, opost.alpha) <= 1e-8
    finally:
        ctx.close()


@pytest.mark.parametrize("inv,cus", [(0, 0), (2, 0), (1, 16), (0, 16), (1, 64)], ids=["subst", "inv_by_recursion", "inv_chain16", "subst_chain16", "inv_chain64"])
@pytest.mark.parametrize("P,Q,n,nb", [(2, 2, 2300, 256), (4, 1, 1800, 128), (2, 4, 2100, 128)])
def test_rows_below_solve_variants_and_the_masked_chain_stream(agp, P, Q, n, nb, inv, cus):
    """The non-default settings of the panel step: "multi_trsm_inv" = 0 (L_kk travels, substitution recursion below it; the default 1 — −inv(L_kk), formed level
    by level in batched launches, travels, one triangular-k GEMM per owner — is what every other test of this file runs; 2 forms the inverse by the recursion) and "multi_chain_cus" = r (the diagonal block's Cholesky + inverse on a
    stream masked to r CUs, panel / main work on the complement), against the oracle at the single-GPU tolerances, two fits each; the masked streams are
    dropped again with 0."""
    x, y = o.synth_inputs(n, 3, 90 + P + Q)
    rng = np.random.default_rng(n)
    s2 = 0.03 + 0.04 * rng.random(n)
    of = o.GP(o.Kernel(o.MATERN32, 1.2, 0.9))
    lp_ref, opost = o.logpdf_and_posterior(o.FiniteGP(of, x, s2), y)
    ctx = agp.Context(devices=rank_devices(P * Q), P=P, Q=Q, nb=nb)
    try:
        assert ctx.get_param("multi_trsm_inv") == 1 and ctx.get_param("multi_chain_cus") == 0      # the defaults
        ctx.set_param("multi_trsm_inv", inv)
        ctx.set_param("multi_chain_cus", cus)
        assert ctx.get_param("multi_trsm_inv") == inv and ctx.get_param("multi_chain_cus") == cus
        f = agp.GP(1.2 * agp.Matern32Kernel() @ agp.ScaleTransform(0.9), ctx=ctx)
        for _ in range(2):
            post = agp.posterior(f(agp.RowVecs(x), s2), y)
            assert float(post.logpdf_value) == pytest.approx(lp_ref, rel=1e-10)
            assert _relnorm(post.data.alpha, opost.alpha) <= 1e-8
        xs = x[:50] + 0.03
        m, v = post.mean_and_var(agp.RowVecs(xs))
        mo, vo = opost.mean_and_var(xs)
        np.testing.assert_allclose(m, mo, atol=1e-8)
        np.testing.assert_allclose(v, vo, atol=1e-9)
        assert np.max(np.abs(post.data.C.U - opost.U)) <= 1e-10
        ctx.set_param("multi_chain_cus", 0)
        assert ctx.get_param("multi_chain_cus") == 0
        post = agp.posterior(f(agp.RowVecs(x), s2), y)
        assert float(post.logpdf_value) == pytest.approx(lp_ref, rel=1e-10)
        assert ctx.multi_stats()["retries"] == 0
    finally:
        ctx.close()


def test_ill_conditioned_inputs_keep_the_substitution_solve(agp):
    """The explicit inverse of a diagonal block is used only while sqrt((variance + max Σy) / min Σy) <= 1e5 (every pivot of K + Σy lies between min Σy and
    variance + max Σy): one observation with noise variance 1e-12 puts the bound at 1e6, and the fit is then BITWISE the fit of a context with
    "multi_trsm_inv" = 0 — while with ordinary noise the two settings agree to rounding only."""
    n = 1200
    x, y = o.synth_inputs(n, 2, 17)
    out = {}
    for tag, tiny in (("guarded", 1e-12), ("plain", 0.05)):
        s2 = np.full(n, 0.05)
        s2[7] = tiny
        for inv in (1, 0):
            ctx = agp.Context(devices=rank_devices(4), P=2, Q=2, nb=128)
            try:
                ctx.set_param("multi_trsm_inv", inv)
                post = agp.posterior(agp.GP(agp.Matern52Kernel(), ctx=ctx)(agp.RowVecs(x), s2), y)
                out[(tag, inv)] = np.array(post.data.C.U)
            finally:
                ctx.close()
    assert np.array_equal(out[("guarded", 1)], out[("guarded", 0)])
    assert not np.array_equal(out[("plain", 1)], out[("plain", 0)])
    np.testing.assert_allclose(out[("plain", 1)], out[("plain", 0)], atol=1e-11)


def test_multi_sequential_update_and_rand_after_gather(agp):
    """posterior(f_post(x2, σ²), y2) on a posterior that was fitted block-cyclically (extended on the pieces, then gathered for
    C.U) and prior sampling through the gathered factor."""
    n1, n2 = 900, 300
    x, y = o.synth_inputs(n1 + n2, 3, 5)
    of = o.GP(o.Kernel(o.MATERN32))
    ob = o.posterior(o.FiniteGP(of, x, 0.05), y)
    ctx = agp.Context(devices=rank_devices(4), P=2, Q=2, nb=128)
    try:
        f = agp.GP(agp.Matern32Kernel(), ctx=ctx)
        p1 = agp.posterior(f(agp.RowVecs(x[:n1]), 0.05), y[:n1])
        p2 = agp.posterior(p1(agp.RowVecs(x[n1:]), 0.05), y[n1:])
        assert _relnorm(p2.data.alpha, ob.alpha) <= 1e-8
        np.testing.assert_allclose(p2.data.C.U, ob.U, atol=1e-9)
        xi = np.random.default_rng(0).standard_normal((n1, 2))
        fx = f(agp.RowVecs(x[:n1]), 0.05)
        np.testing.assert_allclose(agp.rand(fx, 2, xi=xi), o.rand_from(o.FiniteGP(of, x[:n1], 0.05), xi), atol=1e-10)
    finally:
        ctx.close()


def test_multi_not_positive_definite_reports_first_minor(agp):
    """Σy with one large negative entry makes the leading minor of order 401 (and later ones) indefinite while everything
    before it is well conditioned: PosDefException(info) must carry the FIRST failing minor, exactly as LAPACK reports it
    through the reference's cholesky (src/finite_gp_projection.jl:308) — also when several ranks flag later columns."""
    n = 700
    x, y = o.synth_inputs(n, 2, 3)
    s2 = np.full(n, 0.1)
    s2[400] = -5.0
    s2[650] = -7.0
    ctx = agp.Context(devices=rank_devices(4), P=2, Q=2, nb=128)
    one = agp.Context(0)
    try:
        for c in (ctx, one):
            f = agp.GP(agp.SqExponentialKernel(), ctx=c)
            with pytest.raises(agp.PosDefException) as ei:
                agp.posterior(f(agp.RowVecs(x), s2), y)
            assert ei.value.info == 401
    finally:
        ctx.close()
        one.close()


def test_multi_fp32_and_wide_Y_and_other_entry_points_run_on_device0(agp):
    """what the block-cyclic driver does not take (fp32, more than 128 right-hand-side columns) runs on the single-device engine of
    devices[0] — `every other entry point works unchanged` of include/gpmi355.h"""
    x, y = o.synth_inputs(600, 3, 9)
    ctx = agp.Context(devices=rank_devices(2), nb=128)
    try:
        assert ctx.multi_info()["P"] == 2 and ctx.multi_info()["Q"] == 1       # default grid: P = ndev, Q = 1
        f = agp.GP(agp.SqExponentialKernel(), ctx=ctx)
        of = o.GP(o.Kernel(o.SE))
        lp32 = agp.logpdf(f(agp.RowVecs(x.astype(np.float32)), np.float32(0.1)), y.astype(np.float32))
        assert isinstance(lp32, np.float32)
        assert float(lp32) == pytest.approx(float(o.logpdf(o.FiniteGP(of, x.astype(np.float32).astype(np.float64), 0.1), y.astype(np.float32).astype(np.float64))), rel=1e-3)
        Y = np.random.default_rng(0).standard_normal((600, 130))                # 130 columns > 128 RHS rows of the multi-device layout
        np.testing.assert_allclose(agp.logpdf(f(agp.RowVecs(x), 0.1), Y), o.logpdf(o.FiniteGP(of, x, 0.1), Y), rtol=1e-10)
        ctx.trim()                                                              # also empties the rank contexts' caches
        # VFE / kernelmatrix / gradients on a multi ctx run on its first device
        z = x[:50]
        e = agp.elbo(agp.VFE(f(agp.RowVecs(z), 1e-6)), f(agp.RowVecs(x), 0.1), y)
        assert e == pytest.approx(o.elbo(of, z, 1e-6, o.FiniteGP(of, x, 0.1), y), rel=1e-8)
        lp, g = agp.logpdf_and_grad(f(agp.RowVecs(x), 0.1), y)
        assert lp == pytest.approx(float(o.logpdf(o.FiniteGP(of, x, 0.1), y)), rel=1e-10)
    finally:
        ctx.close()


@pytest.mark.parametrize("P,Q", [(1, 1), (2, 1), (1, 2), (2, 2), (4, 2), (2, 3), (8, 1), (3, 1)], ids=lambda v: str(v))
def test_predictive_variance_on_the_distributed_factor(agp, P, Q, tmp_path):
    """var / mean_and_var / marginals of a multi-device posterior WITHOUT gathering the factor (src/exact_gpr_posterior.jl:68-70,
    85-90): a block forward solve with L left on its ranks, N*×nb blocks of the solution travelling (csrc/multi.hip: solve_rank).
    Against the oracle at the single-GPU tolerances, incl. a chunked N* and a ragged one; the schedule of the device run goes
    through the happens-before checker."""
    import os
    import sys
    from pathlib import Path

    sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "tools"))
    import multi_schedule_check as M

    n, d, nb = 1400, 3, 128
    x, y = o.synth_inputs(n, d, 60 + P * 10 + Q)
    rng = np.random.default_rng(P * 7 + Q)
    s2 = 0.03 + 0.05 * rng.random(n)
    of = o.GP(o.Kernel(o.MATERN32, 1.7, 0.7), -0.3)
    opost = o.posterior(o.FiniteGP(of, x, s2), y)
    ctx = agp.Context(devices=rank_devices(P * Q), P=P, Q=Q, nb=nb)
    try:
        f = agp.GP(-0.3, 1.7 * agp.Matern32Kernel() @ agp.ScaleTransform(0.7), ctx=ctx)
        post = agp.posterior(f(agp.RowVecs(x), s2), y)
        xs = rng.standard_normal((333, d)) * 1.1
        trace = tmp_path / "solve.jsonl"
        os.environ["GPMI_TRACE_SCHEDULE"] = str(trace)
        try:
            v = post.var(agp.RowVecs(xs))
        finally:
            os.environ.pop("GPMI_TRACE_SCHEDULE", None)
        mo, vo = opost.mean_and_var(xs)
        np.testing.assert_allclose(v, vo, atol=1e-9)
        assert ctx.multi_stats()["solves"] == 1
        hdr, problems, rs = M.check_trace(trace)
        assert hdr.get("mode") == "solve" and hdr["dry"] == 0 and not problems and not rs, (problems[:3], rs[:3])
        xs2 = rng.standard_normal((4500, d))                                  # two chunks of test points
        m2, v2 = post.mean_and_var(agp.RowVecs(xs2))
        mo2, vo2 = opost.mean_and_var(xs2)
        np.testing.assert_allclose(m2, mo2, atol=1e-8)
        np.testing.assert_allclose(v2, vo2, atol=1e-9)
        assert ctx.multi_stats()["solves"] == 3
        mm, sd = agp.marginals(post(agp.RowVecs(xs), 0.02))
        np.testing.assert_allclose(sd, np.sqrt(vo + 0.02), atol=1e-9)
        # full covariance and joint mean_and_cov on the pieces as well (one SYRK per rank over the solution blocks it owns)
        np.testing.assert_allclose(post.cov(agp.RowVecs(xs[:257])), opost.cov(xs[:257]), atol=1e-9)
        mc, cc = post.mean_and_cov(agp.RowVecs(xs[:40]))
        np.testing.assert_allclose(cc, opost.cov(xs[:40]), atol=1e-9)
        assert ctx.multi_stats()["solves"] == 6
        # held-out logpdf and posterior sampling (src/finite_gp_projection.jl:306-311, 233-237 over a posterior) on the pieces too
        ys = rng.standard_normal((60, 2))
        np.testing.assert_allclose(agp.logpdf(post(agp.RowVecs(xs[:60]), 0.1), ys), o.logpdf(o.FiniteGP(opost, xs[:60], 0.1), ys), rtol=1e-9)
        xi = rng.standard_normal((60, 3))
        np.testing.assert_allclose(agp.rand(post(agp.RowVecs(xs[:60]), 0.05), 3, xi=xi), o.rand_from(o.FiniteGP(opost, xs[:60], 0.05), xi), atol=1e-8)
        assert ctx.multi_stats()["solves"] == 8
        # what needs the whole factor gathers it now (C.U); the variances of the gathered, single-device path agree
        assert np.max(np.abs(post.data.C.U - opost.U)) <= 1e-10
        np.testing.assert_allclose(post.var(agp.RowVecs(xs)), vo, atol=1e-9)
        assert ctx.multi_stats()["solves"] == 8
    finally:
        ctx.close()


@pytest.mark.parametrize("P,Q", [(1, 1), (2, 1), (2, 2), (4, 2), (2, 3), (3, 1), (8, 1), (1, 4)], ids=lambda v: str(v))
def test_sequential_update_and_solve_on_the_pieces(agp, P, Q, tmp_path):
    """posterior(f_post(x2, Σ2), y2) on a multi-device posterior WITHOUT gathering the factor (src/exact_gpr_posterior.jl:46-56,
    update_chol src/util/common_covmat_ops.jl:38-42): the block-cyclic factor is extended where it lives (csrc/multi.hip:
    multi_update) — twice in a row, so that the pieces hold two batches with their own padded blocks — and `C \\ B`
    (gp_posterior_solve) by a forward pass + backward sweeps on the pieces.  α, logpdf, predictions and C.U against the oracle's batch
    posterior at the single-GPU tolerances; the schedule of the device run goes through the happens-before checker."""
    import os
    import sys
    from pathlib import Path

    import scipy.linalg as sla

    sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "tools"))
    import multi_schedule_check as M

    n1, n2, n3, d, nb = 1100, 300, 41, 3, 128
    n = n1 + n2 + n3
    x, y = o.synth_inputs(n, d, 90 + P * 10 + Q)
    rng = np.random.default_rng(P * 11 + Q)
    s2 = 0.04 + 0.05 * rng.random(n)
    of = o.GP(o.Kernel(o.MATERN52, 1.3, 0.8), 0.2)
    ob2 = o.posterior(o.FiniteGP(of, x[: n1 + n2], s2[: n1 + n2]), y[: n1 + n2])
    ob3 = o.posterior(o.FiniteGP(of, x, s2), y)
    ctx = agp.Context(devices=rank_devices(P * Q), P=P, Q=Q, nb=nb)
    try:
        f = agp.GP(0.2, 1.3 * agp.Matern52Kernel() @ agp.ScaleTransform(0.8), ctx=ctx)
        p1 = agp.posterior(f(agp.RowVecs(x[:n1]), s2[:n1]), y[:n1])
        # C \ B on the pieces (3 columns: forward pass + 3 backward sweeps)
        B = rng.standard_normal((n1, 3))
        ob1 = o.posterior(o.FiniteGP(of, x[:n1], s2[:n1]), y[:n1])
        ref = sla.cho_solve((ob1.U, False), B)
        assert _relnorm(p1.data.C.solve(B), ref) <= 1e-8
        assert _relnorm(p1.data.C.solve(B[:, 0]), ref[:, 0]) <= 1e-8
        s0 = ctx.multi_stats()["solves"]
        assert s0 == 2
        trace = tmp_path / "update.jsonl"
        os.environ["GPMI_TRACE_SCHEDULE"] = str(trace)
        try:
            p2 = agp.posterior(p1(agp.RowVecs(x[n1 : n1 + n2]), s2[n1 : n1 + n2]), y[n1 : n1 + n2])
        finally:
            os.environ.pop("GPMI_TRACE_SCHEDULE", None)
        assert ctx.multi_stats()["solves"] == s0 + 2       # K(x2, x1) L11⁻ᵀ on the old pieces, then α on the extended ones: no gather
        hdr, problems, rs = M.check_trace(trace)
        assert hdr.get("mode") == "solve" and hdr["flags"] == 5 and hdr["dry"] == 0 and not problems and not rs, (problems[:3], rs[:3])
        assert _relnorm(p2.data.alpha, ob2.alpha) <= 1e-8
        np.testing.assert_allclose(p2.logpdf_value, o.logpdf(o.FiniteGP(of, x[: n1 + n2], s2[: n1 + n2]), y[: n1 + n2]), rtol=1e-10)
        xs = rng.standard_normal((200, d)) * 1.2
        m2, v2 = p2.mean_and_var(agp.RowVecs(xs))              # predictions of the updated posterior: still on the pieces
        mo2, vo2 = ob2.mean_and_var(xs)
        np.testing.assert_allclose(m2, mo2, atol=1e-8)
        np.testing.assert_allclose(v2, vo2, atol=1e-9)
        np.testing.assert_allclose(p2.cov(agp.RowVecs(xs[:70])), ob2.cov(xs[:70]), atol=1e-9)
        assert ctx.multi_stats()["solves"] == s0 + 4
        # a second update on top (41 points: one more padded batch), then everything again
        p3 = agp.posterior(p2(agp.RowVecs(x[n1 + n2 :]), s2[n1 + n2 :]), y[n1 + n2 :])
        assert ctx.multi_stats()["solves"] == s0 + 6
        assert _relnorm(p3.data.alpha, ob3.alpha) <= 1e-8
        np.testing.assert_allclose(p3.logpdf_value, o.logpdf(o.FiniteGP(of, x, s2), y), rtol=1e-10)
        m3, v3 = p3.mean_and_var(agp.RowVecs(xs))
        mo3, vo3 = ob3.mean_and_var(xs)
        np.testing.assert_allclose(m3, mo3, atol=1e-8)
        np.testing.assert_allclose(v3, vo3, atol=1e-9)
        B3 = rng.standard_normal(n)
        assert _relnorm(p3.data.C.solve(B3), sla.cho_solve((ob3.U, False), B3)) <= 1e-8
        # C.U' ξ (the sampling transform) on the extended pieces: every rank multiplies its blocks, no exchange
        xi3 = rng.standard_normal((n, 2))
        np.testing.assert_allclose(p3.data.C.Ut_mul(xi3), ob3.U.T @ xi3, atol=1e-10)
        assert ctx.multi_stats()["solves"] == s0 + 9
        # the old posteriors are untouched (the reference returns new objects): p1 still predicts from its own pieces
        np.testing.assert_allclose(p1.var(agp.RowVecs(xs[:50])), ob1.mean_and_var(xs[:50])[1], atol=1e-9)
        # C.U gathers the extended pieces — only the real points travel, the padding between the batches is dropped
        assert np.max(np.abs(p3.data.C.U - ob3.U)) <= 1e-9
        np.testing.assert_allclose(p3.var(agp.RowVecs(xs)), vo3, atol=1e-9)   # (now the gathered, single-device path)
        assert np.max(np.abs(p2.data.C.U - ob2.U)) <= 1e-9
        # rand(fx) of the PRIOR at the training inputs: factor of K + Σy fitted block-cyclically, C.U' ξ on its pieces
        sb = ctx.multi_stats()["solves"]
        xi1 = rng.standard_normal((n1, 2))
        np.testing.assert_allclose(agp.rand(f(agp.RowVecs(x[:n1]), s2[:n1]), 2, xi=xi1), o.rand_from(o.FiniteGP(of, x[:n1], s2[:n1]), xi1), atol=1e-10)
        assert ctx.multi_stats()["solves"] == sb + 1
    finally:
        ctx.close()


def test_self_check_repeats_a_spoiled_fit_once(agp):
    """multi_verify (default on): a fit whose alpha does not satisfy delta'alpha = ||L^-1 delta||^2 / (K + Sigma_y) alpha = delta is repeated
    once from the inputs — exercised by spoiling alpha on the host ("multi_inject_fault"); with the check off the spoiled value
    comes through, which is what the check exists to prevent (stale kernel arguments on this stack: DESIGN.md §5)."""
    x, y = o.synth_inputs(900, 2, 21)
    of = o.GP(o.Kernel(o.MATERN32))
    lp_ref, opost = o.logpdf_and_posterior(o.FiniteGP(of, x, 0.05), y)
    ctx = agp.Context(devices=rank_devices(4), P=2, Q=2, nb=128)
    try:
        f = agp.GP(agp.Matern32Kernel(), ctx=ctx)
        post = agp.posterior(f(agp.RowVecs(x), 0.05), y)
        st = ctx.multi_stats()
        assert (st["fits"], st["retries"]) == (1, 0) and _relnorm(post.data.alpha, opost.alpha) <= 1e-8
        ctx.set_param("multi_inject_fault", 1)
        post = agp.posterior(f(agp.RowVecs(x), 0.05), y)
        st = ctx.multi_stats()
        assert (st["fits"], st["retries"]) == (3, 1)
        assert float(post.logpdf_value) == pytest.approx(lp_ref, rel=1e-10) and _relnorm(post.data.alpha, opost.alpha) <= 1e-8
        np.testing.assert_allclose(post.data.C.U, opost.U, atol=1e-10)          # the kept factor is the repeated fit's
        assert agp.logpdf(f(agp.RowVecs(x), 0.05), y) == pytest.approx(lp_ref, rel=1e-10)   # logpdf alone is checked too
        ctx.set_param("multi_verify", 0)
        ctx.set_param("multi_inject_fault", 1)
        bad = agp.posterior(f(agp.RowVecs(x), 0.05), y)
        assert _relnorm(bad.data.alpha, opost.alpha) > 1e-3 and ctx.multi_stats()["retries"] == 1
    finally:
        ctx.close()


def test_more_devices_than_visible_is_an_error(agp):
    import torch

    ndev = torch.cuda.device_count()
    with pytest.raises(ValueError):
        agp.Context(devices=list(range(ndev + 1)))


def test_schedule_under_truly_concurrent_streams():
    """The schedule's event dependencies under real concurrency: with the default 4 hardware queues HIP mostly serialises the
    3·P·Q streams of the virtual ranks, which hid a missing dependency in development (a rank's own panel was not covered by
    its `arrived` event when gcd(P, Q) > 1).  GPU_MAX_HW_QUEUES=32 has to be set before HIP initialises, hence a subprocess:
    tools/multi_diag.py repeats logpdf / matrix-logpdf / posterior on four grids with gcd(P, Q) > 1 or coprime P, Q (up to 16 ranks = 48 streams) and three look-ahead
    depths vs the oracle (round 6: one call of each kind per context, and the P×1 / 1×Q / 2×4 grids that the parity tests above cover left out — 176 s -> ≈ 75 s of a suite that had grown
    past ten minutes; the schedule of every grid is checked exhaustively on its trace without a device, tests/test_multi_schedule.py)."""
    import os
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parent.parent
    env = dict(os.environ, GPU_MAX_HW_QUEUES="32", DIAG_ITERS="3", DIAG_GRIDS="2x2,4x2,2x3,4x4")
    r = subprocess.run([sys.executable, str(root / "tools" / "multi_diag.py")], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "FAILURES 0" in r.stdout


def test_fresh_context_first_fits_with_every_diagnostic_on():
    """The regime of round 2's wrong first fits, in the suite instead of a tool: fresh 8-rank contexts (8x1 and 4x2, 16 hardware queues,
    stream-K and the high-priority comm stream on in the rank contexts), FIRST fit only, "multi_check" 7 (event markers, operand
    verification against the owners' final blocks, NaN-poisoned buffers) and the shipped self-check ("multi_verify" 1, one repetition).
    tools/multi_fresh_stress.py prints one line per wrong result / error and exits 1 on any: a library-side race that the repetition
    would otherwise paper over shows up as a device-side finding of the diagnostics (an ERROR line), not as a silent retry.  The count
    of repetitions is printed into the log (a non-zero count is the arguments-below-the-library effect of DESIGN.md §5.3)."""
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parent.parent
    for comm in ("p2p", "rccl"):
        r = subprocess.run([sys.executable, str(root / "tools" / "multi_fresh_stress.py"), "4", "check=7", "verify=1", f"comm={comm}", "grids=8x1,4x2", "n=1537"],
                           capture_output=True, text=True, timeout=600)
        tail = (r.stdout.strip().splitlines() or [""])[-1]
        print(f"[fresh-context stress, {comm}] {tail}")
        assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-1500:])
        assert "0 bad of 8" in tail, tail
