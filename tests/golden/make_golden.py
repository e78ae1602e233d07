"""Generates tests/golden/*.npz — small input/output vectors for the hot path.

The reference ships NO golden vectors and Julia is not available here, so these are
produced by the CPU oracle (oracle/gp_oracle.py), whose logpdf / α are cross-checked at generation time
against a 60-digit mpmath evaluation (oracle/mp_check.py) where N is small enough.  Re-run:
    python tests/golden/make_golden.py
"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from oracle import gp_oracle as o  # noqa: E402
from oracle import mp_check  # noqa: E402

OUT = Path(__file__).resolve().parent

CASES = [
    # name, n, d, kind, variance, scale, sigma2 (scalar or "vec"), mean, n_star
    ("c1_se_1d_256", 256, 1, o.SE, 1.0, None, 0.01, None, 64),  # BASELINE config 1
    ("se_3d_200", 200, 3, o.SE, 1.0, None, 0.01, None, 50),
    ("mat32_8d_scale_130", 130, 8, o.MATERN32, 1.0, 0.5, 0.01, None, 40),
    ("mat32_8d_ard_130", 130, 8, o.MATERN32, 1.0, "ard", 0.01, 0.3, 40),
    ("mat52_2d_var_97", 97, 2, o.MATERN52, 2.5, 1.7, "vec", None, 33),
    ("mat12_1d_65", 65, 1, o.MATERN12, 0.7, 0.9, 0.05, -1.0, 20),
    ("se_1d_tiny_24", 24, 1, o.SE, 1.3, 1.1, 0.02, 0.5, 8),
]


def build(name, n, d, kind, variance, scale, sigma2, mean, ns):
    seed = abs(hash(name)) % (2**31)
    seed = sum(ord(c) * (i + 1) for i, c in enumerate(name))  # deterministic across processes
    x, y = o.synth_c1() if name == "c1_se_1d_256" else o.synth_inputs(n, d, seed)   # C1: SURVEY.md §8(d)'s literal recipe (seed 1, sin(3x))
    rng = np.random.default_rng(seed + 1)
    xs = rng.standard_normal((ns, d))
    xs = xs[:, 0].copy() if d == 1 else xs
    if isinstance(scale, str):
        scale = np.linspace(0.25, 1.0, d)
    s2 = (0.01 + 0.05 * rng.random(n)) if isinstance(sigma2, str) else sigma2
    k = o.Kernel(kind, variance, scale)
    f = o.GP(k, mean)
    fx = o.FiniteGP(f, x, s2)
    lp = o.logpdf(fx, y)
    Y = np.stack([y, np.cos(y), y**2 - 1.0], axis=1)
    lpY = o.logpdf(fx, Y)
    post = o.posterior(fx, y)
    pm, pv = post.mean_and_var(xs)
    pc = post.cov(xs)
    if n <= 32:  # pin the oracle itself against 60-digit arithmetic
        X2 = o.as_points(x).tolist()
        lp_mp, a_mp = mp_check.logpdf_alpha(kind, variance, k.scale_vec(d).tolist(), X2,
                                            o.noise_diag(s2, n).tolist(), o.mean_vector(mean, x).tolist(), y.tolist())
        assert abs(lp - lp_mp) <= 1e-12 * abs(lp_mp), (lp, lp_mp)
        assert np.max(np.abs(post.alpha - np.array(a_mp))) <= 1e-9 * np.max(np.abs(a_mp))
    # VFE with a subset of the inputs as inducing points
    m_ind = max(8, n // 4)
    z = x[:m_ind].copy()
    jitter = 1e-6
    e = o.elbo(f, z, jitter, fx, y)
    dtc = o.dtc_log_evidence(f, z, jitter, fx, y)
    ap = o.vfe_posterior(f, z, jitter, fx, y)
    vm, vv = ap.mean_and_var(xs)
    # gradient of the sparse objectives (oracle.elbo_grad: dense N×N calculus, pinned by central differences in tests/test_oracle.py); the pseudo-points here are
    # a subset of the inputs at jitter 1e-6 — the ill-conditioned end of what the device is held to
    ge, gd = o.elbo_grad(f, z, jitter, fx, y, vfe=True), o.elbo_grad(f, z, jitter, fx, y, vfe=False)
    gpack = {}
    for tag, g in (("elbo", ge), ("dtc", gd)):
        gpack[f"{tag}_grad_variance"] = g["variance"]
        gpack[f"{tag}_grad_scale"] = np.asarray(np.nan if g["scale"] is None else g["scale"], dtype=np.float64)
        gpack[f"{tag}_grad_noise"] = np.asarray(g["noise"])
        gpack[f"{tag}_grad_y"], gpack[f"{tag}_grad_z"], gpack[f"{tag}_grad_x"] = g["y"], g["z"], g["x"]
    np.savez_compressed(OUT / f"{name}.npz", x=x, y=y, Y=Y, xs=xs, kind=kind, variance=variance,
                        scale=np.asarray(np.nan if scale is None else scale, dtype=np.float64),
                        sigma2=np.asarray(s2), mean=np.asarray(np.nan if mean is None else mean),
                        logpdf=lp, logpdf_Y=lpY, alpha=post.alpha, post_mean=pm, post_var=pv, post_cov=pc,
                        z=z, jitter=jitter, elbo=e, dtc=dtc, vfe_alpha=ap.alpha, vfe_m_eps=ap.m_eps, vfe_mean=vm,
                        vfe_var=vv, **gpack)
    print(f"{name}: logpdf={lp:.12g} elbo={e:.12g}")


if __name__ == "__main__":
    for c in CASES:
        build(*c)
