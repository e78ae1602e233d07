"""Round-2 random sweep against the oracle (one-off wider runs; failures are printed, the exit code is the count):
  exact : the in-library multi-device driver on random virtual grids / block sizes / look-ahead depths, plus the device
          held-out logpdf, posterior sampling and the input gradient on the same random GP
  vfe   : random VFE / DTC fits (both eltypes of noise), predictive covariance, observation update, pseudo-point append."""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import abstractgps_jl_amd as agp  # noqa: E402
from oracle import gp_oracle as o  # noqa: E402

GRIDS = [(1, 1), (2, 1), (1, 2), (2, 2), (3, 1), (4, 1), (2, 3), (4, 2), (2, 4), (8, 1), (1, 3)]


def rel(a, b):
    return np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(b), 1e-300)


def random_gp(rng, d):
    kind = int(rng.integers(0, 4))
    variance = float(rng.uniform(0.3, 2.5))
    tr = rng.integers(0, 3)
    scale = None if tr == 0 else (float(rng.uniform(0.4, 1.6)) if tr == 1 else rng.uniform(0.4, 1.6, d))
    mean = None if rng.random() < 0.5 else float(rng.normal())
    kern = variance * agp.Kernel(kind)
    if scale is not None:
        kern = kern @ (agp.ScaleTransform(scale) if np.ndim(scale) == 0 else agp.ARDTransform(scale))
    return kind, kern, mean, o.GP(o.Kernel(kind, variance, scale), mean)


def exact_case(rng):
    n = int(rng.choice([130, 257, 500, 777, 1024, 1300, 2049]))
    d = int(rng.integers(1, 5))
    kind, kern, mean, of = random_gp(rng, d)
    X = rng.standard_normal((n, d))
    y = np.sin(X.sum(1)) + 0.1 * rng.standard_normal(n)
    sig = float(rng.uniform(0.03, 0.3)) if rng.random() < 0.5 else rng.uniform(0.03, 0.3, n)
    P, Q = GRIDS[int(rng.integers(0, len(GRIDS)))]
    nb = int(rng.choice([128, 256, 384, 512]))   # (384: not 64·2^m — the owner's inverse falls back to the recursion)
    depth = int(rng.integers(1, 4))
    inv = int(rng.choice([1, 1, 2, 0]))          # "multi_trsm_inv": level-wise inverse (default) / by the recursion / substitution solve
    desc = f"n={n} d={d} kind={kind} grid={P}x{Q} nb={nb} depth={depth} trsm_inv={inv}"
    ofx = o.FiniteGP(of, X, sig)
    lp_ref, opost = o.logpdf_and_posterior(ofx, y)
    ctx = agp.Context(devices=[0] * (P * Q), P=P, Q=Q, nb=nb)
    try:
        ctx.set_param("lookahead_depth", depth)
        ctx.set_param("multi_trsm_inv", inv)
        f = agp.GP(kern, ctx=ctx) if mean is None else agp.GP(mean, kern, ctx=ctx)
        fx = f(agp.RowVecs(X), sig)
        post = agp.posterior(fx, y)
        assert abs(float(post.logpdf_value) - lp_ref) <= 1e-10 * abs(lp_ref) + 1e-9, desc
        assert rel(post.data.alpha, opost.alpha) <= 1e-8, desc
        ns = int(rng.integers(3, 40))
        xs = rng.standard_normal((ns, d))
        m, v = post.mean_and_var(agp.RowVecs(xs))
        mo, vo = opost.mean_and_var(xs)
        assert np.max(np.abs(m - mo)) <= 1e-8 and np.max(np.abs(v - vo)) <= 1e-9, desc
        s2s = float(rng.uniform(0.05, 0.3))
        ys = rng.standard_normal(ns)
        lp_ho = float(agp.logpdf(post(agp.RowVecs(xs), s2s), ys))
        assert abs(lp_ho - float(o.logpdf(o.FiniteGP(opost, xs, s2s), ys))) <= 1e-8 * abs(lp_ho) + 1e-8, desc
        xi = rng.standard_normal((ns, 2))
        assert np.max(np.abs(agp.rand(post(agp.RowVecs(xs), s2s), 2, xi=xi) - o.rand_from(o.FiniteGP(opost, xs, s2s), xi))) <= 1e-7, desc
    finally:
        ctx.close()
    if kind != 1 and n <= 800:  # input gradient on a single-device ctx (Matern12 is not differentiable at coincident points)
        f1 = agp.GP(kern) if mean is None else agp.GP(mean, kern)
        _, g = agp.logpdf_and_grad(f1(agp.RowVecs(X), sig), y, wrt_x=True)
        go = o.logpdf_grad(ofx, y)
        assert np.max(np.abs(g["x"] - go["x"])) <= 1e-7 * max(1.0, np.abs(go["x"]).max()), desc + " dx"


def vfe_case(rng):
    n = int(rng.choice([300, 700, 1100]))
    d = int(rng.integers(1, 4))
    kind, kern, mean, of = random_gp(rng, d)
    X = rng.standard_normal((n, d))
    y = np.sin(X.sum(1)) + 0.1 * rng.standard_normal(n)
    sig = float(rng.uniform(0.05, 0.3)) if rng.random() < 0.5 else rng.uniform(0.05, 0.3, n)
    m1, m2 = int(rng.integers(8, 150)), int(rng.integers(3, 140))
    perm = rng.permutation(n)
    z1, z2 = X[perm[:m1]], X[perm[m1:m1 + m2]]
    jitter = 1e-6
    dtc = rng.random() < 0.3
    desc = f"n={n} d={d} kind={kind} m1={m1} m2={m2} dtc={dtc}"
    f = agp.GP(kern) if mean is None else agp.GP(mean, kern)
    A = agp.DTC if dtc else agp.VFE
    n1 = n // 2
    s1 = sig if np.ndim(sig) == 0 else sig[:n1]
    s2 = sig if np.ndim(sig) == 0 else sig[n1:]
    # the oracle takes the reference's own route (src/sparse_approximations.jl:87-176): fit on the first half, update_posterior with the second half, then
    # update_posterior with the new pseudo-points — whose block C22 carries NO jitter in the reference (:138), so a batch fit on vcat(z1, z2) is a different model
    # (and the appended block may be numerically singular: then BOTH sides must report it, as cholesky does in the reference)
    o1 = o.vfe_posterior(of, z1, jitter, o.FiniteGP(of, X[:n1], s1), y[:n1])
    o2 = o.vfe_update_obs(o1, o.FiniteGP(of, X[n1:], s2), y[n1:])
    p1 = agp.posterior(A(f(agp.RowVecs(z1), jitter)), f(agp.RowVecs(X[:n1]), s1), y[:n1])
    p2 = agp.update_posterior(p1, f(agp.RowVecs(X[n1:]), s2), y[n1:])
    try:
        ob = o.vfe_update_z(o2, z2)
    except o.PosDefException:
        try:
            agp.update_posterior(p2, f(agp.RowVecs(z2), jitter))
        except agp.PosDefException:
            return
        return  # (a pivot within rounding of zero: the device's summation order kept it positive — the same edge as the skip below)
    piv = float(np.min(np.diag(ob.U)[m1:]))
    if piv < 1e-4:  # the un-jittered appended block is numerically singular (smallest new pivot < 1e-4: cond > 1e8): both sides amplify rounding differently,
        return      # nothing to compare at 1e-7 — the reference itself would be at the mercy of its BLAS here
    p3 = agp.update_posterior(p2, f(agp.RowVecs(z2), jitter))
    obj = o.objective_from_posterior(ob, o.FiniteGP(of, X, sig), y, vfe=not dtc)
    assert abs(float(p3.objective) - obj) <= 1e-7 * abs(obj) + 1e-7, desc
    xs = rng.standard_normal((17, d))
    mm, cc = p3.mean_and_cov(agp.RowVecs(xs))
    mo, co = ob.mean_and_cov(xs)
    assert np.max(np.abs(mm - mo)) <= 1e-6 and np.max(np.abs(cc - co)) <= 1e-6, desc


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    bad = 0
    only = sys.argv[2] if len(sys.argv) > 2 else ""
    for name, fn, base in [c for c in (("exact", exact_case, 9000), ("vfe", vfe_case, 12000)) if only in ("", c[0])]:
        ok = 0
        for seed in range(n):
            try:
                fn(np.random.default_rng(base + seed))
                ok += 1
            except Exception as e:  # noqa: BLE001  report and continue
                bad += 1
                print("FAIL", name, "seed", base + seed, repr(e)[:400], flush=True)
        print(f"{name}: {ok}/{n} passed", flush=True)
    sys.exit(min(bad, 100))
