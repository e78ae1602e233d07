"""GPU: seeded random sweep over the API surface (sizes around tile boundaries, all kernel kinds and transforms, scalar /
diagonal noise, zero / constant mean, the three input layouts) against the oracle."""
import os

import numpy as np
import pytest

from oracle import gp_oracle as o

pytestmark = pytest.mark.gpu
# GPMI_TEST_RANDOM_CASES / GPMI_TEST_RANDOM_SEED: a longer or different sweep than the 24 cases of the suite (profiles/r5/random_sweep_extra.log)
NCASES = int(os.environ.get("GPMI_TEST_RANDOM_CASES", "24"))
SEED0 = int(os.environ.get("GPMI_TEST_RANDOM_SEED", "1000"))


def _case(agp, rng):
    n = int(rng.choice([1, 7, 63, 64, 65, 127, 128, 129, 200, 511, 512, 777, 1024, 1500, 2049, 2500]))
    d = int(rng.integers(1, 7))
    kind = int(rng.integers(0, 4))
    variance = float(rng.uniform(0.3, 2.5))
    tr = rng.integers(0, 3)
    scale = None if tr == 0 else (float(rng.uniform(0.4, 1.6)) if tr == 1 else rng.uniform(0.4, 1.6, d))
    X = rng.standard_normal((n, d))
    y = np.sin(X.sum(1)) + 0.1 * rng.standard_normal(n)
    sig = float(rng.uniform(0.02, 0.3)) if rng.random() < 0.6 else rng.uniform(0.02, 0.3, n)
    mean = None if rng.random() < 0.5 else float(rng.normal())
    kern = variance * agp.Kernel(kind)
    if scale is not None:
        kern = kern @ (agp.ScaleTransform(scale) if np.ndim(scale) == 0 else agp.ARDTransform(scale))
    f = agp.GP(kern) if mean is None else agp.GP(mean, kern)
    lay = rng.integers(0, 3)
    if d == 1 and lay == 0:
        xin = X[:, 0].copy()
        wrap = lambda a: a[:, 0].copy()
    elif lay == 1:
        xin = agp.ColVecs(np.ascontiguousarray(X.T))
        wrap = lambda a: agp.ColVecs(np.ascontiguousarray(a.T))
    else:
        xin = agp.RowVecs(X)
        wrap = agp.RowVecs
    ofx = o.FiniteGP(o.GP(o.Kernel(kind, variance, scale), mean), X if d > 1 else X[:, 0], sig)
    fx = f(xin, sig)
    lp_ref, opost = o.logpdf_and_posterior(ofx, y)
    desc = f"n={n} d={d} kind={kind} tr={tr} lay={lay} mean={mean} noise={'vec' if np.ndim(sig) else 'scalar'}"
    assert float(agp.logpdf(fx, y)) == pytest.approx(lp_ref, rel=1e-10, abs=1e-9), desc
    post = agp.posterior(fx, y)
    assert float(post.logpdf_value) == pytest.approx(lp_ref, rel=1e-10, abs=1e-9), desc
    assert np.linalg.norm(post.data.alpha - opost.alpha) <= 1e-8 * max(np.linalg.norm(opost.alpha), 1e-30), desc
    xs = rng.standard_normal((5, d))
    m, v = post.mean_and_var(wrap(xs))
    mo, vo = opost.mean_and_var(xs if d > 1 else xs[:, 0])
    np.testing.assert_allclose(m, mo, atol=1e-8, err_msg=desc)
    np.testing.assert_allclose(v, vo, atol=1e-9, err_msg=desc)


@pytest.mark.parametrize("seed", range(NCASES))
def test_random_configuration(agp, seed):
    """production configuration of the default context (tests/conftest.py asserts that it IS the documented default): stream-K GEMM tails
    with fp64 atomics, atomics in the backward sweep"""
    _case(agp, np.random.default_rng(SEED0 + seed))


@pytest.mark.parametrize("exact_mode", ["no_atomics"], indirect=True)
@pytest.mark.parametrize("seed", range(0, NCASES, 3))
def test_random_configuration_without_atomics(agp, seed, exact_mode):
    """the same cases with gemm_streamk = 0 and deterministic = 1: hardware-dispatched GEMMs, no floating-point atomics in the exact path"""
    _case(agp, np.random.default_rng(SEED0 + seed))
