"""GPU: seeded random sweep over what callers do with a fit — value + gradient of logpdf w.r.t. every parameter and the inputs
(AD of src/finite_gp_projection.jl:306-311 in the reference), sequential conditioning (src/exact_gpr_posterior.jl:46-56), the predictive
covariance family and held-out logpdf (:60-90, src/finite_gp_projection.jl:306-311), sampling with given ξ (:233-237) — against the oracle."""
import os

import numpy as np
import pytest

from oracle import gp_oracle as o

pytestmark = pytest.mark.gpu
# GPMI_TEST_RANDOM_CASES / GPMI_TEST_RANDOM_SEED: as in tests/test_gpu_random.py
NCASES = max(1, int(os.environ.get("GPMI_TEST_RANDOM_CASES", "24")) // 3)
SEED0 = int(os.environ.get("GPMI_TEST_RANDOM_SEED", "1000")) + 900000


def _case(agp, rng):
    n = int(rng.choice([2, 63, 64, 65, 127, 128, 129, 200, 384, 511, 513, 700, 1100]))
    n2 = int(rng.choice([1, 9, 64, 130, 300]))
    d = int(rng.integers(1, 6))
    kind = int(rng.integers(0, 4))
    variance = float(rng.uniform(0.3, 2.5))
    tr = rng.integers(0, 3)
    scale = None if tr == 0 else (float(rng.uniform(0.4, 1.6)) if tr == 1 else rng.uniform(0.4, 1.6, d))
    X = rng.standard_normal((n + n2, d))
    y = np.sin(X.sum(1)) + 0.1 * rng.standard_normal(n + n2)
    vec = rng.random() < 0.4
    sig = rng.uniform(0.03, 0.3, n + n2) if vec else float(rng.uniform(0.03, 0.3))
    s1, s2 = (sig[:n], sig[n:]) if vec else (sig, sig)
    mean = None if rng.random() < 0.5 else float(rng.normal())
    kern = variance * agp.Kernel(kind)
    if scale is not None:
        kern = kern @ (agp.ScaleTransform(scale) if np.ndim(scale) == 0 else agp.ARDTransform(scale))
    f = agp.GP(kern) if mean is None else agp.GP(mean, kern)
    of = o.GP(o.Kernel(kind, variance, scale), mean)
    col = rng.random() < 0.5
    wrap = (lambda a: agp.ColVecs(np.ascontiguousarray(a.T))) if col else agp.RowVecs
    desc = f"n={n} n2={n2} d={d} kind={kind} tr={tr} mean={mean} noise={'vec' if vec else 'scalar'} col={col}"
    fx, ofx = f(wrap(X[:n]), s1), o.FiniteGP(of, X[:n], s1)
    # value + gradient
    lp, g = agp.logpdf_and_grad(fx, y[:n], wrt_x=True)
    go = o.logpdf_grad(ofx, y[:n])
    assert float(lp) == pytest.approx(float(o.logpdf(ofx, y[:n])), rel=1e-10, abs=1e-9), desc
    assert g["variance"] == pytest.approx(go["variance"], rel=1e-7, abs=1e-7 * max(1.0, abs(go["variance"]))), desc
    if scale is None:
        assert g["scale"] is None, desc
    else:
        np.testing.assert_allclose(g["scale"], go["scale"], rtol=1e-7, atol=1e-7 * max(1.0, np.abs(go["scale"]).max()), err_msg=desc)
    np.testing.assert_allclose(g["noise"], go["noise"], rtol=1e-7, atol=1e-7 * max(1.0, np.abs(go["noise"]).max()), err_msg=desc)
    np.testing.assert_allclose(g["y"], go["y"], rtol=0, atol=1e-8 * max(np.abs(go["y"]).max(), 1e-30), err_msg=desc)
    gx = g["x"].T if col else g["x"]
    np.testing.assert_allclose(gx, go["x"], rtol=1e-7, atol=1e-7 * max(1.0, np.abs(go["x"]).max()), err_msg=desc)
    # the predictive family and held-out logpdf / sampling from the posterior
    post, opost = agp.posterior(fx, y[:n]), o.posterior(ofx, y[:n])
    xs, zs = rng.standard_normal((11, d)), rng.standard_normal((6, d))
    np.testing.assert_allclose(post.cov(wrap(xs)), opost.cov(xs), atol=1e-9, err_msg=desc)
    np.testing.assert_allclose(post.cov(wrap(xs), wrap(zs)), opost.cov(xs, zs), atol=1e-9, err_msg=desc)
    ys = rng.standard_normal(11)
    assert agp.logpdf(post(wrap(xs), 0.1), ys) == pytest.approx(float(o.logpdf(o.FiniteGP(opost, xs, 0.1), ys)), rel=1e-9, abs=1e-9), desc
    xi = rng.standard_normal((11, 2))
    np.testing.assert_allclose(agp.rand(post(wrap(xs), 0.05), 2, xi=xi), o.rand_from(o.FiniteGP(opost, xs, 0.05), xi), atol=1e-7, err_msg=desc)
    # sequential conditioning on n2 more points ≡ the oracle's
    p2 = agp.posterior(post(wrap(X[n:]), s2), y[n:])
    op2 = o.posterior(o.FiniteGP(opost, X[n:], s2), y[n:])
    np.testing.assert_allclose(p2.data.alpha, op2.alpha, rtol=0, atol=1e-8 * max(np.abs(op2.alpha).max(), 1e-30), err_msg=desc)
    m, v = p2.mean_and_var(wrap(xs))
    mo, vo = op2.mean_and_var(xs)
    np.testing.assert_allclose(m, mo, atol=1e-8, err_msg=desc)
    np.testing.assert_allclose(v, vo, atol=1e-9, err_msg=desc)


@pytest.mark.parametrize("seed", range(NCASES))
def test_random_next_rows(agp, seed):
    _case(agp, np.random.default_rng(SEED0 + seed))
