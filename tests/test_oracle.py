"""CPU: pins the oracle (oracle/gp_oracle.py) against the identities the reference's own tests assert,
using independent implementations, against a 60-digit mpmath evaluation, and against tests/golden/.
Reference test file:line is cited per test (paths relative to the upstream repo)."""
import glob
from pathlib import Path

import numpy as np
import pytest
import scipy.stats

from oracle import gp_oracle as o
from oracle import mp_check

GOLDEN = sorted(glob.glob(str(Path(__file__).parent / "golden" / "*.npz")))


def _case(path):
    g = np.load(path)
    scale = None if np.isnan(g["scale"]).all() else (float(g["scale"]) if g["scale"].ndim == 0 else g["scale"])
    mean = None if np.isnan(g["mean"]) else float(g["mean"])
    f = o.GP(o.Kernel(int(g["kind"]), float(g["variance"]), scale), mean)
    s2 = float(g["sigma2"]) if g["sigma2"].ndim == 0 else g["sigma2"]
    return g, f, o.FiniteGP(f, g["x"], s2)


@pytest.mark.parametrize("path", GOLDEN, ids=lambda p: Path(p).stem)
def test_golden_reproduces(path):
    g, f, fx = _case(path)
    assert o.logpdf(fx, g["y"]) == pytest.approx(float(g["logpdf"]), rel=1e-12)
    np.testing.assert_allclose(o.logpdf(fx, g["Y"]), g["logpdf_Y"], rtol=1e-12)
    post = o.posterior(fx, g["y"])
    np.testing.assert_allclose(post.alpha, g["alpha"], rtol=1e-9, atol=1e-9)
    m, v = post.mean_and_var(g["xs"])
    np.testing.assert_allclose(m, g["post_mean"], atol=1e-10)
    np.testing.assert_allclose(v, g["post_var"], atol=1e-10)
    assert o.elbo(f, g["z"], float(g["jitter"]), fx, g["y"]) == pytest.approx(float(g["elbo"]), rel=1e-10)
    for tag, vfe in (("elbo", True), ("dtc", False)):   # the sparse objectives' gradients (round 6)
        gr = o.elbo_grad(f, g["z"], float(g["jitter"]), fx, g["y"], vfe=vfe)
        assert gr["variance"] == pytest.approx(float(g[f"{tag}_grad_variance"]), rel=1e-7)
        for key in ("noise", "y", "z", "x"):
            ref = g[f"{tag}_grad_{key}"]
            np.testing.assert_allclose(gr[key], ref, rtol=0, atol=1e-7 * max(1.0, float(np.max(np.abs(ref)))))


@pytest.mark.parametrize("path", GOLDEN, ids=lambda p: Path(p).stem)
def test_logpdf_vs_mvnormal(path):
    """logpdf(fx, y) ≈ logpdf(MvNormal(mean, cov), y) — test/finite_gp_projection.jl:143; columns :147-150."""
    g, f, fx = _case(path)
    m, C = o.mean_and_cov(fx)
    ref = scipy.stats.multivariate_normal(m, C, allow_singular=False).logpdf(g["y"])
    assert o.logpdf(fx, g["y"]) == pytest.approx(ref, rel=1.5e-8)
    lpY = o.logpdf(fx, g["Y"])
    for s in range(g["Y"].shape[1]):
        assert lpY[s] == pytest.approx(o.logpdf(fx, g["Y"][:, s]), rel=1e-12)


def test_logpdf_vs_mpmath():
    x, y = o.synth_inputs(20, 2, 7)
    k = o.Kernel(o.MATERN52, 1.4, np.array([0.8, 1.3]))
    s2 = 0.03
    lp = o.logpdf(o.FiniteGP(o.GP(k, 0.2), x, s2), y)
    lp_mp, a_mp = mp_check.logpdf_alpha(k.kind, k.variance, k.scale_vec(2).tolist(), x.tolist(), [s2] * 20, [0.2] * 20,
                                        y.tolist())
    assert lp == pytest.approx(lp_mp, rel=1e-12)
    post = o.posterior(o.FiniteGP(o.GP(k, 0.2), x, s2), y)
    np.testing.assert_allclose(post.alpha, a_mp, rtol=1e-9, atol=1e-10)


def test_covmat_ops_identities():
    """test/util/common_covmat_ops.jl:52-97 on a 5×5 SPD matrix."""
    rng = np.random.default_rng(0)
    B = rng.standard_normal((5, 5))
    A = B @ B.T + np.eye(5)
    U = o.cholesky_upper(A.copy())
    np.testing.assert_allclose(U.T @ U, A, atol=1e-12)
    X = rng.standard_normal((5, 3))
    np.testing.assert_allclose(o.Xt_invA_X(U, X), X.T @ np.linalg.solve(A, X), atol=1e-12)  # :73
    np.testing.assert_allclose(o.diag_Xt_invA_X(U, X), np.diag(X.T @ np.linalg.solve(A, X)), atol=1e-12)  # :92
    assert o.tr_Xt_invA_X(U, X) == pytest.approx(np.trace(X.T @ np.linalg.solve(A, X)))
    assert o.logdet_chol(U) == pytest.approx(np.linalg.slogdet(A)[1])


def test_update_chol_matches_full():
    """test/util/common_covmat_ops.jl:21-37 (atol 1e-5)."""
    rng = np.random.default_rng(1)
    B = rng.standard_normal((9, 9))
    A = B @ B.T + 9 * np.eye(9)
    U1 = o.cholesky_upper(A[:5, :5].copy())
    U = o.update_chol(U1, A[:5, 5:], A[5:, 5:])
    np.testing.assert_allclose(U, o.cholesky_upper(A.copy()), atol=1e-5)


def test_posterior_interpolates():
    """test/exact_gpr_posterior.jl:14-22: mean(f_post, x) ≈ y, var ≈ 0 with tiny noise."""
    rng = np.random.default_rng(2)
    x = np.sort(rng.uniform(-3, 3, 12))
    y = np.sin(x)
    post = o.posterior(o.FiniteGP(o.GP(o.Kernel(o.MATERN32)), x, 1e-12), y)
    m, v = post.mean_and_var(x)
    np.testing.assert_allclose(m, y, atol=1e-8)
    np.testing.assert_allclose(v, 0, atol=1e-8)


def test_sequential_conditioning_equals_batch():
    """test/exact_gpr_posterior.jl:29-43 (atol 1e-5 on C.U, α, δ)."""
    x, y = o.synth_inputs(40, 1, 3)
    f = o.GP(o.Kernel(o.SE))
    batch = o.posterior(o.FiniteGP(f, x, 0.1), y)
    p1 = o.posterior(o.FiniteGP(f, x[:25], 0.1), y[:25])
    p2 = o.posterior(o.FiniteGP(p1, x[25:], 0.1), y[25:])
    np.testing.assert_allclose(p2.U, batch.U, atol=1e-5)
    np.testing.assert_allclose(p2.alpha, batch.alpha, atol=1e-5)
    np.testing.assert_allclose(p2.delta, batch.delta, atol=1e-5)


def test_vfe_z_equals_x_matches_exact():
    """test/sparse_approximations.jl:20-25, :94; src/util/TestUtils.jl:213-217 (rtol=atol=1e-5)."""
    x, y = o.synth_inputs(60, 1, 4)
    f = o.GP(o.Kernel(o.SE))
    fx = o.FiniteGP(f, x, 0.1)
    exact = o.posterior(fx, y)
    ap = o.vfe_posterior(f, x, 1e-9, fx, y)
    xs = np.linspace(-2, 2, 100)
    np.testing.assert_allclose(ap.mean(xs), exact.mean(xs), atol=1e-5, rtol=1e-5)
    np.testing.assert_allclose(ap.cov(xs), exact.cov(xs), atol=1e-5, rtol=1e-5)
    lp = o.logpdf(fx, y)
    assert o.elbo(f, x, 1e-9, fx, y) == pytest.approx(lp, rel=1e-5, abs=1e-5)
    assert o.dtc_log_evidence(f, x, 1e-9, fx, y) == pytest.approx(lp, rel=1e-5, abs=1e-5)


def test_elbo_below_logpdf_and_dtc_doctest():
    """elbo(z≠x) < logpdf — test/sparse_approximations.jl:99; DTC doctest src/sparse_approximations.jl:263-276."""
    rng = np.random.default_rng(5)
    x = rng.standard_normal(300)
    f = o.GP(o.Kernel(o.MATERN52))
    K = o.kernelmatrix(f.kernel, x) + 0.1 * np.eye(300)
    y = np.linalg.cholesky(K) @ rng.standard_normal(300)
    fx = o.FiniteGP(f, x, 0.1)
    lp = o.logpdf(fx, y)
    assert o.elbo(f, np.linspace(-5, 5, 13), 1e-12, fx, y) < lp
    assert o.dtc_log_evidence(f, np.linspace(-5, 5, 256), 1e-10, fx, y) == pytest.approx(lp, rel=1e-6, abs=1e-6)


def test_posdef_exception_info():
    """cholesky throws PosDefException(info) — src/finite_gp_projection.jl:308."""
    A = np.eye(4)
    A[2, 2] = -1.0
    with pytest.raises(o.PosDefException) as e:
        o.cholesky_upper(A)
    assert e.value.info == 3


def test_float32_type_stability():
    """logpdf(fx, y) isa T — test/finite_gp_projection.jl:180-191."""
    x, y = o.synth_inputs(30, 1, 6, dtype=np.float32)
    lp = o.logpdf(o.FiniteGP(o.GP(o.Kernel(o.SE)), x, np.float32(0.1)), y, dtype=np.float32)
    assert lp.dtype == np.float32


def test_logpdf_grad_matches_finite_differences():
    """oracle.logpdf_grad (dense matrix calculus) against central differences of oracle.logpdf — the check the
    reference runs with FiniteDifferences vs AD (test/finite_gp_projection.jl:152-178)."""
    rng = np.random.default_rng(21)
    n, d = 40, 3
    X = rng.standard_normal((n, d))
    y = np.sin(X.sum(1)) + 0.1 * rng.standard_normal(n)
    for kind in (o.SE, o.MATERN32, o.MATERN52):
        for scale in (0.8, np.array([0.5, 1.1, 0.9])):
            base = dict(kind=kind, variance=1.4, scale=scale)
            sig = 0.07
            g = o.logpdf_grad(o.FiniteGP(o.GP(o.Kernel(**base), 0.2), X, sig), y)

            def lp(variance=1.4, scale=scale, sig=sig, yy=y):
                return float(o.logpdf(o.FiniteGP(o.GP(o.Kernel(kind, variance, scale), 0.2), X, sig), yy))

            h = 1e-6
            assert g["variance"] == pytest.approx((lp(variance=1.4 + h) - lp(variance=1.4 - h)) / (2 * h), rel=1e-6, abs=1e-6)
            assert g["noise"] == pytest.approx((lp(sig=sig + h) - lp(sig=sig - h)) / (2 * h), rel=1e-5, abs=1e-5)
            if np.ndim(scale) == 0:
                assert g["scale"] == pytest.approx((lp(scale=scale + h) - lp(scale=scale - h)) / (2 * h), rel=1e-6, abs=1e-6)
            else:
                for p in range(d):
                    e = np.zeros(d)
                    e[p] = h
                    fd = (lp(scale=scale + e) - lp(scale=scale - e)) / (2 * h)
                    assert g["scale"][p] == pytest.approx(fd, rel=1e-6, abs=1e-6)
            e0 = np.zeros(n)
            e0[5] = h
            assert g["y"][5] == pytest.approx((lp(yy=y + e0) - lp(yy=y - e0)) / (2 * h), rel=1e-6, abs=1e-6)
            for i, p in [(0, 0), (7, 2), (n - 1, 1)]:  # ∂/∂x (deep-kernel style input gradients)
                E = np.zeros_like(X)
                E[i, p] = h
                fd = (float(o.logpdf(o.FiniteGP(o.GP(o.Kernel(**base), 0.2), X + E, sig), y))
                      - float(o.logpdf(o.FiniteGP(o.GP(o.Kernel(**base), 0.2), X - E, sig), y))) / (2 * h)
                assert g["x"][i, p] == pytest.approx(fd, rel=1e-6, abs=1e-6)


@pytest.mark.parametrize("kind,nu", [(o.SE, None), (o.MATERN12, 0.5), (o.MATERN32, 1.5), (o.MATERN52, 2.5)])
@pytest.mark.parametrize("ard", [False, True], ids=["scale", "ard"])
def test_oracle_vs_scikit_learn(kind, nu, ard):
    """An external pin of the restatement: scikit-learn's GaussianProcessRegressor is an independent implementation of exactly this
    path (Gram matrix of RBF / Matern kernels with (anisotropic) length scales, Cholesky, log marginal likelihood, predictive mean /
    covariance — Rasmussen & Williams alg. 2.1, which is what src/finite_gp_projection.jl:306-311 and src/exact_gpr_posterior.jl:29-90
    compute).  Kernel map: k ∘ ScaleTransform(s) / ARDTransform(v) <-> length_scale 1/s, 1/v; α·k <-> ConstantKernel(α); Σy (scalar or
    per-point) <-> alpha.  Not the reference itself (Julia is not available), but not this repository's code either."""
    from sklearn.gaussian_process import GaussianProcessRegressor
    from sklearn.gaussian_process.kernels import RBF, ConstantKernel, Matern

    n, d, var = 150, 3, 1.7
    x, y = o.synth_inputs(n, d, 40 + kind)
    rng = np.random.default_rng(kind * 2 + ard)
    scale = np.array([0.6, 1.3, 0.9]) if ard else 0.8
    s2 = 0.02 + 0.05 * rng.random(n) if ard else 0.05            # per-point noise in one half of the cases
    ls = 1.0 / np.asarray(scale, dtype=float)
    base = RBF(length_scale=ls) if nu is None else Matern(length_scale=ls, nu=nu)
    gpr = GaussianProcessRegressor(kernel=ConstantKernel(var) * base, alpha=s2, optimizer=None, normalize_y=False).fit(x, y)
    of = o.GP(o.Kernel(kind, var, scale))
    fx = o.FiniteGP(of, x, s2)
    assert float(o.logpdf(fx, y)) == pytest.approx(gpr.log_marginal_likelihood_value_, rel=1e-10)
    post = o.posterior(fx, y)
    np.testing.assert_allclose(post.alpha, gpr.alpha_.ravel(), rtol=1e-8, atol=1e-10)
    xs = rng.standard_normal((40, d))
    m_sk, c_sk = gpr.predict(xs, return_cov=True)
    m, v = post.mean_and_var(xs)
    np.testing.assert_allclose(m, m_sk, atol=1e-9)
    np.testing.assert_allclose(post.cov(xs), c_sk, atol=1e-9)
    np.testing.assert_allclose(v, np.diag(c_sk), atol=1e-9)
    # and the Gram matrix itself (kernelmatrix parity is otherwise pinned by closed forms only)
    np.testing.assert_allclose(o.kernelmatrix(of.kernel, x), gpr.kernel_(x), rtol=1e-13, atol=1e-14)


@pytest.mark.parametrize("kind", [o.SE, o.MATERN32])
def test_vfe_against_the_dense_textbook_formulas(kind):
    """The sparse restatement at GENERAL pseudo-points z ≠ x against formulas that share no code with it: Q_ff = K_fu K_uu⁻¹ K_uf by
    explicit inverses, DTC evidence = log N(y | m, Q_ff + Σy) by scipy's MvNormal, ELBO = DTC − ½ tr(Σy⁻¹ (K_ff − Q_ff)) (Titsias 2009,
    eq. 9 — what src/sparse_approximations.jl:248-313 evaluates through Cholesky factors), predictive mean / covariance
    K_*u Λ⁻¹ K_uf Σy⁻¹ (y − m) and K_** − K_*u K_uu⁻¹ K_u* + K_*u Λ⁻¹ K_u*, Λ = K_uu + K_uf Σy⁻¹ K_fu (:183-217).  Per-point noise,
    constant mean, D = 2."""
    n, mz, d, jitter = 120, 20, 2, 1e-6
    x, y = o.synth_inputs(n, d, 70 + kind)
    rng = np.random.default_rng(7 + kind)
    z = rng.standard_normal((mz, d)) * 1.2
    s2 = 0.05 + 0.1 * rng.random(n)
    f = o.GP(o.Kernel(kind, 1.4, 0.9), 0.25)
    fx = o.FiniteGP(f, x, s2)
    Kff = o.kernelmatrix(f.kernel, x)
    Kuu = o.kernelmatrix(f.kernel, z) + jitter * np.eye(mz)
    Kuf = o.kernelmatrix(f.kernel, z, x)
    Qff = Kuf.T @ np.linalg.inv(Kuu) @ Kuf
    dtc = scipy.stats.multivariate_normal(mean=np.full(n, 0.25), cov=Qff + np.diag(s2), allow_singular=False).logpdf(y)
    elbo = dtc - 0.5 * np.sum((np.diag(Kff) - np.diag(Qff)) / s2)
    assert o.dtc_log_evidence(f, z, jitter, fx, y) == pytest.approx(dtc, rel=1e-9)
    assert o.elbo(f, z, jitter, fx, y) == pytest.approx(elbo, rel=1e-9)
    ap = o.vfe_posterior(f, z, jitter, fx, y)
    xs = rng.standard_normal((30, d))
    Ksu = o.kernelmatrix(f.kernel, xs, z)
    Lam = Kuu + (Kuf / s2) @ Kuf.T
    mean = 0.25 + Ksu @ np.linalg.solve(Lam, (Kuf / s2) @ (y - 0.25))
    cov = o.kernelmatrix(f.kernel, xs) - Ksu @ np.linalg.solve(Kuu, Ksu.T) + Ksu @ np.linalg.solve(Lam, Ksu.T)
    np.testing.assert_allclose(ap.mean(xs), mean, atol=1e-8)
    np.testing.assert_allclose(ap.cov(xs), cov, atol=1e-8)
    np.testing.assert_allclose(ap.mean_and_var(xs)[1], np.diag(cov), atol=1e-8)


@pytest.mark.parametrize("vfe", [True, False])
@pytest.mark.parametrize("kind", [o.SE, o.MATERN12, o.MATERN32, o.MATERN52])
def test_elbo_grad_matches_finite_differences(kind, vfe):
    """oracle.elbo_grad (dense N×N calculus on C = Q_ff + Σy) against central differences of oracle.elbo / dtc_log_evidence, for every
    parameter block: variance, Scale / ARD parameters, scalar / vector noise, y, pseudo-inputs, inputs — the pin of the oracle the device's
    gp_vfe_grad is compared with (tests/test_gpu_vfe_grad.py)."""
    rng = np.random.default_rng(40 + kind)
    n, m, d, h = 60, 9, 2, 1e-6
    X, Z = rng.normal(size=(n, d)), rng.normal(size=(m, d))
    y = np.sin(X.sum(1)) + 0.1 * rng.normal(size=n)
    for scale in (None, 0.7, np.array([0.5, 1.2])):
        for s2 in (0.08, 0.05 + 0.1 * rng.random(n)):
            par = {"var": 1.3, "scale": scale, "s2": s2, "X": X, "Z": Z, "y": y}

            def obj(**kw):
                q = dict(par, **kw)
                f = o.GP(o.Kernel(kind, q["var"], q["scale"]))
                fx = o.FiniteGP(f, q["X"], q["s2"])
                return o.elbo(f, q["Z"], 1e-3, fx, q["y"]) if vfe else o.dtc_log_evidence(f, q["Z"], 1e-3, fx, q["y"])

            f = o.GP(o.Kernel(kind, 1.3, scale))
            g = o.elbo_grad(f, Z, 1e-3, o.FiniteGP(f, X, s2), y, vfe)

            def check(name, an, **plus_minus):
                lo = {k: par[k] - v for k, v in plus_minus.items()}
                hi = {k: par[k] + v for k, v in plus_minus.items()}
                fd = (obj(**hi) - obj(**lo)) / (2 * h)
                assert abs(fd - an) <= 1e-6 * max(1.0, abs(fd)), (name, fd, an)

            check("variance", g["variance"], var=h)
            if scale is None:
                assert g["scale"] is None
            elif np.ndim(scale) == 0:
                check("scale", g["scale"], scale=h)
            else:
                for p in range(d):
                    e = np.zeros(d)
                    e[p] = h
                    check("ard", g["scale"][p], scale=e)
            if np.ndim(s2) == 0:
                check("noise", g["noise"], s2=h)
            else:
                for i in (0, 31):
                    e = np.zeros(n)
                    e[i] = h
                    check("noise_i", g["noise"][i], s2=e)
            for i, p in ((0, 0), (5, 1)):
                E = np.zeros_like(Z)
                E[i, p] = h
                check("z", g["z"][i, p], Z=E)
                E = np.zeros_like(X)
                E[i, p] = h
                check("x", g["x"][i, p], X=E)
                e = np.zeros(n)
                e[i] = h
                check("y", g["y"][i], y=e)


def test_elbo_grad_survives_an_ill_conditioned_kzz():
    """128 pseudo-points on a line under the SE kernel with jitter 1.8e-6 (cond(K_zz) ≈ 1e8 — a configuration of the round-6 random sweep): the
    pseudo-input gradient along a direction against central differences of `elbo` (itself Cholesky-based).  The first form of `elbo_grad`, with
    explicit inverses, was 0.8 % off here; the device's gradient agreed with the differences."""
    rng = np.random.default_rng(1770111)
    n, m = 640, 128
    X, Z = rng.standard_normal((n, 1)), rng.standard_normal((m, 1))
    y = np.sin(X[:, 0]) + 0.1 * rng.standard_normal(n)
    f = o.GP(o.Kernel(o.SE, 1.7), 0.4)
    fx = o.FiniteGP(f, X, 0.11)
    g = o.elbo_grad(f, Z, 1.8e-6, fx, y)
    dZ = rng.standard_normal(Z.shape)
    fds = [(o.elbo(f, Z + h * dZ, 1.8e-6, fx, y) - o.elbo(f, Z - h * dZ, 1.8e-6, fx, y)) / (2 * h) for h in (1e-4, 1e-5)]
    an = float(np.sum(g["z"] * dZ))
    assert abs(fds[0] - fds[1]) <= 1e-3 * abs(fds[1])          # the difference quotient has converged
    assert an == pytest.approx(fds[1], rel=1e-3), (an, fds)
