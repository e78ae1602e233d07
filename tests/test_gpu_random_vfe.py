"""GPU: seeded random sweep over the sparse (VFE / DTC) path — sizes around tile and chunk boundaries, all kernel kinds and transforms, scalar /
diagonal noise, zero / constant mean, fp64 — against the oracle's restatement of src/sparse_approximations.jl:58-75 (posterior), :183-203
(predictions), :248-313 (elbo / dtc log evidence), then update_posterior with more observations (:87-129) against the oracle's batch fit."""
import os

import numpy as np
import pytest

from oracle import gp_oracle as o

pytestmark = pytest.mark.gpu
# GPMI_TEST_RANDOM_CASES / GPMI_TEST_RANDOM_SEED: as in tests/test_gpu_random.py
NCASES = max(1, int(os.environ.get("GPMI_TEST_RANDOM_CASES", "24")) // 2)
SEED0 = int(os.environ.get("GPMI_TEST_RANDOM_SEED", "1000")) + 500000


def _case(agp, rng):
    n = int(rng.choice([40, 127, 128, 129, 300, 640, 1000, 2047, 2048, 2049, 3000]))
    m = int(rng.choice([1, 5, 31, 63, 64, 65, 127, 128, 129, 200]))
    m = min(m, n // 2)
    n2 = int(rng.choice([0, 1, 77, 300]))
    d = int(rng.integers(1, 6))
    kind = int(rng.integers(0, 4))
    variance = float(rng.uniform(0.3, 2.5))
    tr = rng.integers(0, 3)
    scale = None if tr == 0 else (float(rng.uniform(0.4, 1.6)) if tr == 1 else rng.uniform(0.4, 1.6, d))
    X = rng.uniform(-2, 2, (n + n2, d))
    y = np.sin(X.sum(1)) + 0.1 * rng.standard_normal(n + n2)
    z = X[rng.permutation(n)[:m]] + 0.01 * rng.standard_normal((m, d))
    vec = rng.random() < 0.4
    sig = rng.uniform(0.02, 0.3, n + n2) if vec else float(rng.uniform(0.02, 0.3))
    s1, s2, sall = (sig[:n], sig[n:], sig) if vec else (sig, sig, sig)
    mean = None if rng.random() < 0.5 else float(rng.normal())
    dtc = rng.random() < 0.3
    jitter = 1e-4
    kern = variance * agp.Kernel(kind)
    if scale is not None:
        kern = kern @ (agp.ScaleTransform(scale) if np.ndim(scale) == 0 else agp.ARDTransform(scale))
    f = agp.GP(kern) if mean is None else agp.GP(mean, kern)
    of = o.GP(o.Kernel(kind, variance, scale), mean)
    desc = f"n={n} m={m} n2={n2} d={d} kind={kind} tr={tr} mean={mean} noise={'vec' if vec else 'scalar'} dtc={dtc}"
    A = agp.DTC if dtc else agp.VFE
    ap = agp.posterior(A(f(agp.RowVecs(z), jitter)), f(agp.RowVecs(X[:n]), s1), y[:n])
    ofx = o.FiniteGP(of, X[:n], s1)
    obj = o.dtc_log_evidence(of, z, jitter, ofx, y[:n]) if dtc else o.elbo(of, z, jitter, ofx, y[:n])
    assert float(ap.objective) == pytest.approx(obj, rel=1e-8, abs=1e-7), desc
    oap = o.vfe_posterior(of, z, jitter, ofx, y[:n])
    xs = rng.uniform(-2, 2, (9, d))
    mu, v = ap.mean_and_var(agp.RowVecs(xs))
    np.testing.assert_allclose(mu, oap.mean(xs), atol=1e-7, err_msg=desc)
    np.testing.assert_allclose(v, oap.var(xs), atol=1e-8, err_msg=desc)
    np.testing.assert_allclose(ap.cov(agp.RowVecs(xs)), oap.cov(xs), atol=1e-8, err_msg=desc)
    if n2:
        ap2 = agp.update_posterior(ap, f(agp.RowVecs(X[n:]), s2), y[n:])
        oap2 = o.vfe_posterior(of, z, jitter, o.FiniteGP(of, X, sall), y)
        mu, v = ap2.mean_and_var(agp.RowVecs(xs))
        np.testing.assert_allclose(mu, oap2.mean(xs), atol=1e-7, err_msg=desc)
        np.testing.assert_allclose(v, oap2.var(xs), atol=1e-8, err_msg=desc)


@pytest.mark.parametrize("seed", range(NCASES))
def test_random_sparse_configuration(agp, seed):
    _case(agp, np.random.default_rng(SEED0 + seed))
