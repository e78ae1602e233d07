"""60-digit mpmath restatement of logpdf / α for tiny N: bounds the fp64 oracle's own error.

TEST INFRASTRUCTURE ONLY.  Follows src/finite_gp_projection.jl:306-311 and
src/exact_gpr_posterior.jl:29-35 of the reference with exact-ish arithmetic."""
import mpmath as mp


def _kappa(kind, d2):
    if kind == 0:
        return mp.e ** (-d2 / 2)
    d = mp.sqrt(d2)
    if kind == 1:
        return mp.e ** (-d)
    if kind == 2:
        a = mp.sqrt(3) * d
        return (1 + a) * mp.e ** (-a)
    a = mp.sqrt(5) * d
    return (1 + a + mp.mpf(5) / 3 * d2) * mp.e ** (-a)


def logpdf_alpha(kind, variance, scale, X, sigma2, mean, y, dps=60):
    """X: list of points (each a list of floats), scale: list per dim.  Returns (logpdf, alpha) as floats."""
    mp.mp.dps = dps
    n = len(X)
    Xs = [[mp.mpf(v) * mp.mpf(s) for v, s in zip(p, scale)] for p in X]
    K = mp.matrix(n, n)
    for i in range(n):
        for j in range(n):
            d2 = sum((a - b) ** 2 for a, b in zip(Xs[i], Xs[j]))
            K[i, j] = mp.mpf(variance) * _kappa(kind, d2)
        K[i, i] += mp.mpf(sigma2[i])
    L = mp.cholesky(K)
    delta = mp.matrix([mp.mpf(a) - mp.mpf(b) for a, b in zip(y, mean)])
    z = mp.lu_solve(L, delta)  # L is triangular; exact enough at 60 digits
    logdet = 2 * sum(mp.log(L[i, i]) for i in range(n))
    lp = -(n * mp.log(2 * mp.pi) + logdet + sum(v * v for v in z)) / 2
    alpha = mp.lu_solve(L.T, z)
    return float(lp), [float(a) for a in alpha]
