"""GPU parity tests proper: the HIP path (through the C ABI / the API mirror) against the CPU oracle
and the committed golden fixtures on the same seeded inputs.  Tolerances (fp64):
K entries abs <= 1e-14·σ_k²; logpdf rel <= 1e-10; α rel(2-norm) <= 1e-8; predictive mean abs <= 1e-8,
var abs <= 1e-9.  fp32: logpdf/ELBO rel <= 1e-4 against the fp64 oracle.

What the two comparisons are: the committed fixtures tests/golden/*.npz were WRITTEN BY THE ORACLE (tests/golden/make_golden.py imports
oracle.gp_oracle), so "golden and oracle" is one source seen twice — a regression pin of the oracle's past output plus the oracle's present
output, not two independent references.  The independent pins live in tests/test_oracle.py (MvNormal, 60-digit mpmath, scikit-learn's
GaussianProcessRegressor) and, once a maintainer has run tests/golden/make_golden.jl, in tests/test_julia_golden.py (the real AbstractGPs.jl).
(The fixture `c1_se_1d_256` is BASELINE config C1 by SURVEY.md §8(d)'s literal recipe — N = 256, D = 1, x ~ N(0, 1) from PCG64 seed 1,
y = sin(3x) + 0.1 ε, SE, σ² = 0.01: `oracle.gp_oracle.synth_c1`; `__graft_entry__.smoke()` runs the same inputs.)"""
import glob
from pathlib import Path

import numpy as np
import pytest

from oracle import gp_oracle as o

pytestmark = pytest.mark.gpu
GOLDEN = sorted(glob.glob(str(Path(__file__).parent / "golden" / "*.npz")))


def _golden(agp, path):
    g = np.load(path)
    kind, var = int(g["kind"]), float(g["variance"])
    k = var * agp.Kernel(kind)
    ok = None
    if not np.isnan(g["scale"]).all():
        if g["scale"].ndim == 0:
            k = k @ agp.ScaleTransform(float(g["scale"]))
            ok = float(g["scale"])
        else:
            k = k @ agp.ARDTransform(g["scale"])
            ok = g["scale"]
    mean = None if np.isnan(g["mean"]) else float(g["mean"])
    s2 = float(g["sigma2"]) if g["sigma2"].ndim == 0 else g["sigma2"]
    f = agp.GP(k) if mean is None else agp.GP(mean, k)
    of = o.GP(o.Kernel(kind, var, ok), mean)
    return g, f, f(g["x"], s2), of, o.FiniteGP(of, g["x"], s2)


def _relnorm(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


@pytest.mark.parametrize("path", GOLDEN, ids=lambda p: Path(p).stem)
def test_kernelmatrix_vs_oracle(agp, path):
    g, f, fx, of, ofx = _golden(agp, path)
    K = agp.kernelmatrix(f.kernel, g["x"])
    Ko = o.kernelmatrix(of.kernel, g["x"])
    assert np.max(np.abs(K - Ko)) <= 1e-14 * of.kernel.variance
    assert np.array_equal(K, K.T)  # exactly symmetric, like kernelmatrix(k, x)
    Kc = agp.kernelmatrix(f.kernel, g["x"], g["xs"])
    assert np.max(np.abs(Kc - o.kernelmatrix(of.kernel, g["x"], g["xs"]))) <= 1e-14 * of.kernel.variance


@pytest.mark.parametrize("path", GOLDEN, ids=lambda p: Path(p).stem)
def test_logpdf_posterior_vs_golden_and_oracle(agp, path):
    """device against the oracle's committed output (fixture) and the oracle's present output — one source twice, see the module docstring"""
    g, f, fx, of, ofx = _golden(agp, path)
    lp = agp.logpdf(fx, g["y"])
    assert isinstance(lp, np.float64)
    assert lp == pytest.approx(float(g["logpdf"]), rel=1e-10)
    np.testing.assert_allclose(agp.logpdf(fx, g["Y"]), g["logpdf_Y"], rtol=1e-10)  # matrix Y: per column
    post = agp.posterior(fx, g["y"])
    assert post.logpdf_value == pytest.approx(float(g["logpdf"]), rel=1e-10)
    assert _relnorm(post.data.alpha, g["alpha"]) <= 1e-8
    np.testing.assert_allclose(post.data.delta, o.posterior(ofx, g["y"]).delta, atol=0)
    # factor: C.U of the reference (column-major upper)
    U = post.data.C.U
    Uo = o.posterior(ofx, g["y"]).U
    assert np.max(np.abs(U - Uo)) <= 1e-10
    m, v = agp.mean_and_var(post(g["xs"], 0.0))
    np.testing.assert_allclose(m, g["post_mean"], atol=1e-8)
    np.testing.assert_allclose(v, g["post_var"], atol=1e-9)
    np.testing.assert_allclose(post.cov(g["xs"]), g["post_cov"], atol=1e-9)
    mm, cc = post.mean_and_cov(g["xs"])
    np.testing.assert_allclose(mm, g["post_mean"], atol=1e-8)
    np.testing.assert_allclose(np.diag(cc), g["post_var"], atol=1e-9)


@pytest.mark.parametrize("path", GOLDEN, ids=lambda p: Path(p).stem)
def test_vfe_vs_golden(agp, path):
    g, f, fx, of, ofx = _golden(agp, path)
    vfe = agp.VFE(f(g["z"], float(g["jitter"])))
    assert agp.elbo(vfe, fx, g["y"]) == pytest.approx(float(g["elbo"]), rel=1e-8)
    assert agp.approx_log_evidence(agp.DTC(vfe.fz), fx, g["y"]) == pytest.approx(float(g["dtc"]), rel=1e-8)
    ap = agp.posterior(vfe, fx, g["y"])
    assert _relnorm(ap.data["alpha"], g["vfe_alpha"]) <= 1e-5  # α carries cond(K_zz) at jitter 1e-6
    m, v = ap.mean_and_var(g["xs"])
    np.testing.assert_allclose(m, g["vfe_mean"], atol=1e-7)
    np.testing.assert_allclose(v, g["vfe_var"], atol=1e-7)


@pytest.mark.parametrize("path", GOLDEN, ids=lambda p: Path(p).stem)
def test_vfe_grad_vs_golden(agp, path):
    """gp_vfe_grad against the committed fixtures' `elbo_grad_*` / `dtc_grad_*` fields (written by oracle.elbo_grad; tests/golden/make_golden.py): the
    pseudo-points of the fixtures are a subset of the inputs at jitter 1e-6 — cond(K_zz) up to 1e8, so every block is held to 1e-5 of its largest
    component (the well-conditioned configurations of tests/test_gpu_vfe_grad.py are held to 1e-7)."""
    g, f, fx, of, ofx = _golden(agp, path)
    for tag, A in (("elbo", agp.VFE), ("dtc", agp.DTC)):
        val, gr = agp.elbo_and_grad(A(f(g["z"], float(g["jitter"]))), fx, g["y"], wrt_x=True)
        assert val == pytest.approx(float(g[tag]), rel=1e-8)
        pairs = [("variance", gr["variance"]), ("noise", gr["noise_diag"] if g["sigma2"].ndim else gr["noise"]), ("y", gr["y"]), ("z", gr["z"]), ("x", gr["x"])]
        if gr["scale"] is not None:
            pairs.append(("scale", gr["scale"]))
        for key, got in pairs:
            ref = np.asarray(g[f"{tag}_grad_{key}"], dtype=np.float64)
            got = np.asarray(got, dtype=np.float64).reshape(ref.shape)
            err = float(np.max(np.abs(got - ref))) / max(1.0, float(np.max(np.abs(ref))))
            assert err <= 1e-5, (tag, key, err)


@pytest.mark.parametrize("n,d,kind,layout", [(1000, 3, 0, "row"), (777, 8, 2, "col"), (2048, 1, 3, "vec"),
                                             (4099, 3, 0, "row")])
def test_mid_size_parity(agp, n, d, kind, layout):
    """ragged N (padding), all input layouts, outer-panel loop + look-ahead (N > nb)."""
    x, y = o.synth_inputs(n, d, 100 + n)
    scale = 0.7
    of = o.GP(o.Kernel(kind, 1.0, scale))
    ref_lp, ref_post = o.logpdf_and_posterior(o.FiniteGP(of, x, 0.01), y)
    xin = x if d == 1 else (agp.RowVecs(x) if layout == "row" else agp.ColVecs(x.T.copy()))
    ctx = agp.default_context()
    ctx.set_param("nb", 1024)
    try:
        f = agp.GP(agp.Kernel(kind) @ agp.ScaleTransform(scale))
        post = agp.posterior(f(xin, 0.01), y)
    finally:
        ctx.set_param("nb", -1)
    assert post.logpdf_value == pytest.approx(ref_lp, rel=1e-10)
    assert _relnorm(post.data.alpha, ref_post.alpha) <= 1e-8
    xs = x[:130] + 0.05
    m, v = post.mean_and_var(xs if d == 1 else agp.RowVecs(xs))
    mo, vo = ref_post.mean_and_var(xs)
    np.testing.assert_allclose(m, mo, atol=1e-8)
    np.testing.assert_allclose(v, vo, atol=1e-9)


def test_variants_agree(agp):
    """Panel widths, look-ahead on/off, recursive-only, XCD-aware workgroup order, stream-K tails on / off, the round-2 GEMM k loop: same answer."""
    x, y = o.synth_inputs(3000, 3, 9)
    f = agp.GP(agp.SqExponentialKernel())
    ctx = agp.default_context()
    vals = []

    def reset():  # back to the documented defaults (include/gpmi355.h); tests/conftest.py asserts it before and after every GPU test
        ctx.set_param("nb", -1), ctx.set_param("lookahead", 1), ctx.set_param("xcd_swizzle", 0)
        ctx.set_param("xcd_min_tiles", 256), ctx.set_param("gemm_streamk", 1), ctx.set_param("leaf_group", 128)
        ctx.set_param("gemm_pipe", 1)

    try:
        for nb, la, extra in [(-1, 1, {}), (2048, 1, {}), (1024, 0, {"gemm_streamk": 0}), (0, 0, {"gemm_streamk": 0}), (1024, 1, {}), (512, 0, {"gemm_streamk": 0}),
                              (512, 1, {"xcd_swizzle": 1, "xcd_min_tiles": 4, "gemm_streamk": 0}),
                              (512, 0, {"xcd_swizzle": 1, "xcd_min_tiles": 4}), (0, 0, {}),
                              (1024, 1, {"leaf_group": 64, "gemm_streamk": 0}), (1024, 1, {"leaf_group": 256}),
                              (1024, 0, {"gemm_pipe": 0}), (512, 1, {"gemm_pipe": 0, "gemm_streamk": 0})]:
            ctx.set_param("nb", nb), ctx.set_param("lookahead", la)
            for kname, kval in extra.items():
                ctx.set_param(kname, kval)
            vals.append(float(agp.logpdf(f(agp.RowVecs(x), 0.01), y)))
            reset()
    finally:
        reset()
    ref = float(o.logpdf(o.FiniteGP(o.GP(o.Kernel(o.SE)), x, 0.01), y))
    for v in vals:
        assert v == pytest.approx(ref, rel=1e-10)


def test_posdef_exception(agp):
    """Not-PD surfaces as PosDefException(info) like cholesky at src/finite_gp_projection.jl:308."""
    x = np.zeros(300)
    x[150:] = np.linspace(0, 1, 150)
    f = agp.GP(agp.SqExponentialKernel())
    with pytest.raises(agp.PosDefException) as e:
        agp.logpdf(f(x, -0.5), np.zeros(300))  # K + (-0.5) I is indefinite
    assert 1 <= e.value.info <= 300


def test_float32_type_stability_and_accuracy(agp):
    """Float32 in -> Float32 out (test/finite_gp_projection.jl:180-191); value within fp32 tolerance."""
    x, y = o.synth_inputs(1500, 2, 11)
    f = agp.GP(agp.Matern52Kernel())
    lp32 = agp.logpdf(f(agp.RowVecs(x.astype(np.float32)), np.float32(0.1)), y.astype(np.float32))
    assert isinstance(lp32, np.float32)
    ref = o.logpdf(o.FiniteGP(o.GP(o.Kernel(o.MATERN52)), x, 0.1), y)
    assert float(lp32) == pytest.approx(ref, rel=1e-4)


def test_interpolation_and_noise_vector(agp):
    """test/exact_gpr_posterior.jl:14-22: posterior interpolates with tiny noise; vector noise accepted."""
    rng = np.random.default_rng(3)
    x = np.sort(rng.uniform(-3, 3, 40))
    y = np.sin(x)
    post = agp.posterior(agp.GP(agp.Matern32Kernel())(x, 1e-12), y)
    m, v = post.mean_and_var(x)
    np.testing.assert_allclose(m, y, atol=1e-7)
    np.testing.assert_allclose(v, 0, atol=1e-7)
    s2 = 0.01 + 0.1 * rng.random(40)
    lp = agp.logpdf(agp.GP(agp.Matern32Kernel())(x, s2), y)
    assert lp == pytest.approx(o.logpdf(o.FiniteGP(o.GP(o.Kernel(o.MATERN32)), x, s2), y), rel=1e-10)


def test_elbo_z_equals_x_matches_logpdf(agp):
    """src/util/TestUtils.jl:213-217: elbo(VFE(f(x, jitter)), fx, y) ≈ logpdf(fx, y), rtol = atol = 1e-5."""
    x, y = o.synth_inputs(400, 1, 21)
    f = agp.GP(agp.SqExponentialKernel())
    fx = f(x, 0.1)
    lp = agp.logpdf(fx, y)
    assert agp.elbo(agp.VFE(f(x, 1e-7)), fx, y) == pytest.approx(lp, rel=1e-5, abs=1e-5)
    z = np.linspace(-3, 3, 17)
    assert agp.elbo(agp.VFE(f(z, 1e-9)), fx, y) < lp  # test/sparse_approximations.jl:99


def test_vfe_fp32_streaming(agp):
    """config-5 shape at reduced size: fp32 streamed SYRK + fp64 M×M side vs the fp64 oracle."""
    rng = np.random.default_rng(5)
    n, m = 20000, 256
    X = rng.uniform(0, 4, (n, 3))
    y = np.sin(X.sum(1)) + 0.3 * rng.standard_normal(n)
    z = X[rng.permutation(n)[:m]]
    of = o.GP(o.Kernel(o.SE))
    ref = o.elbo(of, z, 1e-4, o.FiniteGP(of, X, 0.1), y)
    f = agp.GP(agp.SqExponentialKernel())
    X32, y32, z32 = X.astype(np.float32), y.astype(np.float32), z.astype(np.float32)
    got = agp.elbo(agp.VFE(f(agp.RowVecs(z32), 1e-4)), f(agp.RowVecs(X32), np.float32(0.1)), y32)
    assert isinstance(got, np.float32)
    assert float(got) == pytest.approx(ref, rel=1e-4)
    got64 = agp.elbo(agp.VFE(f(agp.RowVecs(z), 1e-4)), f(agp.RowVecs(X), 0.1), y)
    assert got64 == pytest.approx(ref, rel=1e-7)


@pytest.mark.parametrize("dual,inv_nb", [(1, 128), (0, 256), (1, 0), (0, 0)], ids=["dual_inv128", "single_inv256", "dual_leaves", "single_leaves"])
def test_vfe_schedule_variants(agp, dual, inv_nb):
    """The two round-6 switches of the sparse fit on a private context: "vfe_dual" (the chunk's SYRK on a third stream beside the next chunk's triangular
    product; 0 = back to back) and "vfe_inv_nb" (inv(L_z) of the prelude through batched inverse diagonal blocks of that width — small here so that
    M = 700 pseudo-points (padded 768: six 128-blocks, ragged three 256-blocks) takes the path the C5 default takes at M = 4 096; 0 = 64-wide leaves).
    Three chunks of 16 384 observations (double buffers reused), fp32 and fp64, fit + ELBO + predictions + update_posterior against the fp64 oracle
    (src/sparse_approximations.jl:58-75, 248-254, 183-217, 87-121)."""
    rng = np.random.default_rng(55)
    n, n2, m, d = 40000, 3000, 700, 3
    X = rng.uniform(0, 4, (n + n2, d)).astype(np.float32).astype(np.float64)
    y = (np.sin(X.sum(1)) + 0.3 * rng.standard_normal(n + n2)).astype(np.float32).astype(np.float64)
    z = X[rng.permutation(n)[:m]].copy()
    xs = rng.uniform(0, 4, (200, d)).astype(np.float32).astype(np.float64)
    of = o.GP(o.Kernel(o.SE))
    ofx = o.FiniteGP(of, X[:n], 0.1)
    op = o.vfe_posterior(of, z, 1e-4, ofx, y[:n])
    elbo = o.objective_from_posterior(op, ofx, y[:n], vfe=True)
    mo, vo = op.mean_and_var(xs)
    op2 = o.vfe_posterior(of, z, 1e-4, o.FiniteGP(of, X, 0.1), y)
    ctx = agp.Context(0)
    ctx.set_param("vfe_dual", dual)
    ctx.set_param("vfe_inv_nb", inv_nb)
    ctx.set_param("vfe_chunk", 16384)   # three chunks (the automatic choice for M = 700 would stream this batch as one)
    try:
        f = agp.GP(agp.SqExponentialKernel(), ctx=ctx)
        for dt, rtol, atol in ((np.float32, 1e-4, 1e-3), (np.float64, 1e-8, 1e-6)):
            ap = agp.posterior(agp.VFE(f(agp.RowVecs(z.astype(dt)), 1e-4)), f(agp.RowVecs(X[:n].astype(dt)), dt(0.1)), y[:n].astype(dt))
            assert float(ap.objective) == pytest.approx(elbo, rel=rtol)
            mm, vv = ap.mean_and_var(agp.RowVecs(xs.astype(dt)))
            np.testing.assert_allclose(mm, mo, atol=atol)
            np.testing.assert_allclose(vv, vo, atol=atol)
            ap2 = agp.update_posterior(ap, f(agp.RowVecs(X[n:].astype(dt)), dt(0.1)), y[n:].astype(dt))
            np.testing.assert_allclose(ap2.mean(agp.RowVecs(xs.astype(dt))), op2.mean(xs), atol=atol)
            del ap, ap2
    finally:
        ctx.close()


def test_sequential_conditioning_matches_batch(agp):
    """posterior(p_fx1(X2, σ²), y2) ≡ posterior(f(X, σ²), y) — reference test/exact_gpr_posterior.jl:29-43 (atol 1e-5 there),
    here also against the oracle's update_chol path and with n1, n2 that are not tile multiples."""
    rng = np.random.default_rng(11)
    for n1, n2, d in [(3, 2, 1), (300, 77, 3), (1000, 500, 2)]:
        X = rng.standard_normal((n1 + n2, d))
        y = np.sin(X.sum(1)) + 0.1 * rng.standard_normal(n1 + n2)
        xin = (lambda a: a[:, 0]) if d == 1 else agp.RowVecs
        f = agp.GP(0.3, 1.3 * agp.Matern52Kernel() @ agp.ScaleTransform(0.7))
        p1 = agp.posterior(f(xin(X[:n1]), 0.1), y[:n1])
        p12 = agp.posterior(p1(xin(X[n1:]), 0.1), y[n1:])
        pb = agp.posterior(f(xin(X), 0.1), y)
        np.testing.assert_allclose(p12.data.alpha, pb.data.alpha, rtol=0, atol=1e-8 * np.abs(pb.data.alpha).max())
        np.testing.assert_allclose(p12.data.delta, pb.data.delta, rtol=0, atol=1e-14)
        np.testing.assert_allclose(np.triu(p12.data.C.U), np.triu(pb.data.C.U), rtol=0, atol=1e-10)
        assert float(p12.logpdf_value) == pytest.approx(float(pb.logpdf_value), rel=1e-10)
        of = o.GP(o.Kernel(o.MATERN52, 1.3, 0.7), 0.3)
        xo = X[:, 0] if d == 1 else X
        op1 = o.posterior(o.FiniteGP(of, xo[:n1], 0.1), y[:n1])
        op12 = o.posterior(o.FiniteGP(op1, xo[n1:], 0.1), y[n1:])
        np.testing.assert_allclose(p12.data.alpha, op12.alpha, rtol=0, atol=1e-8 * np.abs(op12.alpha).max())
        xs = xin(X[:7] + 0.05)
        m_g, v_g = p12.mean_and_var(xs)
        m_o, v_o = op12.mean_and_var(xo[:7] + 0.05)
        np.testing.assert_allclose(m_g, m_o, atol=1e-8)
        np.testing.assert_allclose(v_g, v_o, atol=1e-9)


def test_rand_matches_oracle_with_given_normals(agp):
    """rand(rng, fx, N) = m .+ C.U' * randn — src/finite_gp_projection.jl:233-237, same standard normals on both sides."""
    rng = np.random.default_rng(3)
    n, d = 333, 2
    X = rng.standard_normal((n, d))
    xi = rng.standard_normal((n, 3))
    f = agp.GP(-0.2, agp.SqExponentialKernel() @ agp.ScaleTransform(1.7))
    got = agp.rand(f(agp.RowVecs(X), 0.05), 3, xi=xi)
    ref = o.rand_from(o.FiniteGP(o.GP(o.Kernel(o.SE, 1.0, 1.7), -0.2), X, 0.05), xi)
    assert got.shape == (n, 3)
    np.testing.assert_allclose(got, ref, rtol=0, atol=1e-11)
    one = agp.rand(f(agp.RowVecs(X), 0.05), xi=xi[:, 0])
    np.testing.assert_allclose(one, ref[:, 0], rtol=0, atol=1e-11)
    # posterior samples at a few test points (sampleplot-style usage)
    y = np.sin(X.sum(1))
    post = agp.posterior(f(agp.RowVecs(X), 0.05), y)
    opost = o.posterior(o.FiniteGP(o.GP(o.Kernel(o.SE, 1.0, 1.7), -0.2), X, 0.05), y)
    xs = rng.standard_normal((9, d))
    xi2 = rng.standard_normal((9, 2))
    mo, Co = opost.mean_and_cov(xs)
    ref2 = mo[:, None] + np.linalg.cholesky(Co + 1e-6 * np.eye(9)) @ xi2
    got2 = agp.rand(post(agp.RowVecs(xs), 1e-6), 2, xi=xi2)
    np.testing.assert_allclose(got2, ref2, rtol=0, atol=1e-7)


@pytest.mark.parametrize("kind,okind", [(0, o.SE), (1, o.MATERN12), (2, o.MATERN32), (3, o.MATERN52)])
def test_logpdf_grad_vs_oracle(agp, kind, okind):
    """gp_logpdf_grad against the oracle's dense-calculus gradient (itself checked against finite differences in
    tests/test_oracle.py): kernel variance, ScaleTransform / ARDTransform parameters, scalar and diagonal noise, y."""
    rng = np.random.default_rng(40 + kind)
    n, d = 300, 3
    X = rng.standard_normal((n, d))
    y = np.sin(X.sum(1)) + 0.1 * rng.standard_normal(n)
    for scale, sig in [(None, 0.05), (0.8, 0.05), (np.array([0.5, 1.1, 0.9]), rng.uniform(0.03, 0.1, n))]:
        kern = 1.4 * agp.Kernel(kind)
        if scale is not None:
            kern = kern @ (agp.ScaleTransform(scale) if np.ndim(scale) == 0 else agp.ARDTransform(scale))
        f = agp.GP(0.2, kern)
        lp, g = agp.logpdf_and_grad(f(agp.RowVecs(X), sig), y)
        ofx = o.FiniteGP(o.GP(o.Kernel(okind, 1.4, scale), 0.2), X, sig)
        go = o.logpdf_grad(ofx, y)
        assert float(lp) == pytest.approx(float(o.logpdf(ofx, y)), rel=1e-10)
        sc = max(1.0, abs(go["variance"]))
        assert g["variance"] == pytest.approx(go["variance"], rel=1e-8, abs=1e-8 * sc)
        if scale is None:
            assert g["scale"] is None
        else:
            np.testing.assert_allclose(g["scale"], go["scale"], rtol=1e-8, atol=1e-8 * max(1.0, np.abs(go["scale"]).max()))
        np.testing.assert_allclose(g["noise"], go["noise"], rtol=1e-7, atol=1e-7 * max(1.0, np.abs(go["noise"]).max()))
        np.testing.assert_allclose(g["y"], go["y"], rtol=0, atol=1e-8 * np.abs(go["y"]).max())


@pytest.mark.parametrize("exact_mode", ["default", "no_atomics"], indirect=True)
@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 127, 128, 129, 255, 257, 1025])
def test_tile_boundary_sizes(agp, n, exact_mode):
    """ragged sizes around every tile boundary (64-wide leaves, 128-wide tiles, identity padding), custom mean function,
    matrix-valued Y, cross-covariance of the posterior — in the production configuration (stream-K GEMM tails with fp64 atomics: what
    every launch of these sizes takes) and, on purpose, without any floating-point atomics (gemm_streamk = 0, deterministic = 1)."""
    rng = np.random.default_rng(n)
    X = rng.standard_normal((n, 2))
    Y = rng.standard_normal((n, 3))
    mfun = lambda v: 0.3 * v[0] - 0.1  # CustomMean  src/mean_function.jl:52-55
    f = agp.GP(mfun, 0.7 * agp.Matern32Kernel() @ agp.ScaleTransform(1.3))
    of = o.GP(o.Kernel(o.MATERN32, 0.7, 1.3), mfun)
    fx, ofx = f(agp.RowVecs(X), 0.2), o.FiniteGP(of, X, 0.2)
    np.testing.assert_allclose(agp.logpdf(fx, Y), o.logpdf(ofx, Y), rtol=1e-10)
    post, opost = agp.posterior(fx, Y[:, 0]), o.posterior(ofx, Y[:, 0])
    assert _relnorm(post.data.alpha, opost.alpha) <= 1e-8
    xs, zs = rng.standard_normal((5, 2)), rng.standard_normal((4, 2))
    np.testing.assert_allclose(post.cov(agp.RowVecs(xs), agp.RowVecs(zs)), opost.cov(xs, zs), atol=1e-9)
    m, v = post.mean_and_var(agp.RowVecs(xs))
    mo, vo = opost.mean_and_var(xs)
    np.testing.assert_allclose(m, mo, atol=1e-8)
    np.testing.assert_allclose(v, vo, atol=1e-9)


def test_float32_gradient_update_and_rand(agp):
    """Float32 in -> Float32 out for the newer entry points too (reference type-stability tests,
    test/finite_gp_projection.jl:180-191), values within fp32 accuracy of the fp64 oracle."""
    rng = np.random.default_rng(8)
    n, d = 500, 2
    X = rng.standard_normal((n, d))
    y = np.sin(X.sum(1)) + 0.1 * rng.standard_normal(n)
    X32, y32 = X.astype(np.float32), y.astype(np.float32)
    f = agp.GP(1.2 * agp.SqExponentialKernel() @ agp.ScaleTransform(0.9))
    ofx = o.FiniteGP(o.GP(o.Kernel(o.SE, 1.2, 0.9)), X, 0.1)
    lp, g = agp.logpdf_and_grad(f(agp.RowVecs(X32), np.float32(0.1)), y32)
    go = o.logpdf_grad(ofx, y)
    assert isinstance(lp, np.float32) and g["y"].dtype == np.float32
    assert float(lp) == pytest.approx(float(o.logpdf(ofx, y)), rel=2e-4)
    assert g["variance"] == pytest.approx(go["variance"], rel=5e-3, abs=5e-3 * abs(go["noise"]))
    assert g["scale"] == pytest.approx(go["scale"], rel=5e-3, abs=5e-3 * abs(go["noise"]))
    assert float(g["noise"]) == pytest.approx(go["noise"], rel=5e-3)
    p1 = agp.posterior(f(agp.RowVecs(X32[:300]), np.float32(0.1)), y32[:300])
    p12 = agp.posterior(p1(agp.RowVecs(X32[300:]), np.float32(0.1)), y32[300:])
    assert p12.data.alpha.dtype == np.float32
    oa = o.posterior(ofx, y).alpha
    assert _relnorm(p12.data.alpha.astype(np.float64), oa) <= 5e-3
    xi = rng.standard_normal((n, 2)).astype(np.float32)
    smp = agp.rand(f(agp.RowVecs(X32), np.float32(0.1)), 2, xi=xi)
    assert smp.dtype == np.float32
    np.testing.assert_allclose(smp, o.rand_from(ofx, xi.astype(np.float64)), atol=5e-3)


@pytest.mark.parametrize("dtc", [False, True])
def test_vfe_update_posterior_matches_batch(agp, dtc):
    """update_posterior(f_post_approx, fx2, y2) ≡ posterior(vfe, f([x; x2]), [y; y2]) and pseudo-point appends ≡ a fit with
    vcat(z_old, z_new) — reference test/sparse_approximations.jl:32-84 — also against the oracle's batch fit."""
    rng = np.random.default_rng(17)
    n1, n2, m, d = 700, 333, 40, 2
    X = rng.uniform(-2, 2, (n1 + n2, d))
    y = np.sin(X.sum(1)) + 0.1 * rng.standard_normal(n1 + n2)
    z = rng.uniform(-2, 2, (m, d))
    z2 = rng.uniform(-2, 2, (7, d))
    f = agp.GP(0.1, 0.9 * agp.Matern52Kernel() @ agp.ScaleTransform(1.1))
    A = agp.DTC if dtc else agp.VFE
    appr = A(f(agp.RowVecs(z), 1e-9))
    p1 = agp.posterior(appr, f(agp.RowVecs(X[:n1]), 0.07), y[:n1])
    p12 = agp.update_posterior(p1, f(agp.RowVecs(X[n1:]), 0.07), y[n1:])
    pb = agp.posterior(appr, f(agp.RowVecs(X), 0.07), y)
    assert float(p12.objective) == pytest.approx(float(pb.objective), rel=1e-9)
    np.testing.assert_allclose(p12.data["alpha"], pb.data["alpha"], rtol=0, atol=1e-7 * np.abs(pb.data["alpha"]).max())
    xs = rng.uniform(-2, 2, (11, d))
    m1, v1 = p12.mean_and_var(agp.RowVecs(xs))
    m2, v2 = pb.mean_and_var(agp.RowVecs(xs))
    np.testing.assert_allclose(m1, m2, atol=1e-8)
    np.testing.assert_allclose(v1, v2, atol=1e-9)
    of = o.GP(o.Kernel(o.MATERN52, 0.9, 1.1), 0.1)
    op = o.vfe_posterior(of, z, 1e-9, o.FiniteGP(of, X, 0.07), y)
    mo, vo = op.mean_and_var(xs)
    np.testing.assert_allclose(m1, mo, atol=1e-7)
    np.testing.assert_allclose(v1, vo, atol=1e-8)
    # pseudo-point append
    pz = agp.update_posterior(p12, f(agp.RowVecs(z2), 1e-9))
    opz = o.vfe_posterior(of, np.concatenate([z, z2]), 1e-9, o.FiniteGP(of, X, 0.07), y)
    mz, vz = pz.mean_and_var(agp.RowVecs(xs))
    moz, voz = opz.mean_and_var(xs)
    np.testing.assert_allclose(mz, moz, atol=1e-7)
    np.testing.assert_allclose(vz, voz, atol=1e-8)


def test_multi_panel_parity_8192(agp):
    """N = 8 192 (four 2 048-column panels, look-ahead, recursion depth 5, RHS block row) against the oracle's LAPACK path."""
    x, y = o.synth_inputs(8192, 3, 12)
    f = agp.GP(agp.SqExponentialKernel() @ agp.ScaleTransform(1.2))
    post = agp.posterior(f(agp.RowVecs(x), 0.01), y)
    lp, opost = o.logpdf_and_posterior(o.FiniteGP(o.GP(o.Kernel(o.SE, 1.0, 1.2)), x, 0.01), y)
    assert float(post.logpdf_value) == pytest.approx(lp, rel=1e-10)
    assert _relnorm(post.data.alpha, opost.alpha) <= 1e-8
    xs = x[:64] + 0.03
    m, v = post.mean_and_var(agp.RowVecs(xs))
    mo, vo = opost.mean_and_var(xs)
    np.testing.assert_allclose(m, mo, atol=1e-8)
    np.testing.assert_allclose(v, vo, atol=1e-9)


def test_two_stream_lookahead_forced_at_a_small_size(agp):
    """The look-ahead schedule (panel stream beside the trailing update) is off below N = 24 576 by default ("lookahead_min_n"); forced
    on with narrow panels it gives the same numbers as the oracle.  (The CU-partitioned variant this test used to drive was removed in
    round 4: measured slower at every size, profiles/r4/nb_sweep.jsonl.)"""
    n = 10240
    x, y = o.synth_inputs(n, 3, 77)
    lp, opost = o.logpdf_and_posterior(o.FiniteGP(o.GP(o.Kernel(o.MATERN52, 1.2, 0.9)), x, 0.02), y)
    ctx = agp.Context(0)
    try:
        for k, v in {"lookahead": 1, "lookahead_min_n": 0, "nb": 512}.items():
            ctx.set_param(k, v)
        f = agp.GP(1.2 * agp.Matern52Kernel() @ agp.ScaleTransform(0.9), ctx=ctx)
        for _ in range(2):
            post = agp.posterior(f(agp.RowVecs(x), 0.02), y)
            assert float(post.logpdf_value) == pytest.approx(lp, rel=1e-10)
            assert _relnorm(post.data.alpha, opost.alpha) <= 1e-8
    finally:
        ctx.close()
