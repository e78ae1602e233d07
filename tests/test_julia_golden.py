"""Reference-side golden vectors: tests/golden/julia/<case>.gpb are written by tests/golden/make_golden.jl, which runs the REAL
AbstractGPs.jl (logpdf, posterior, mean_and_var, cov, sequential posterior, VFE posterior / elbo / DTC evidence, both
update_posterior forms) on tests/golden/julia_inputs/<case>.gpb — the inputs of the committed fixtures, bit for bit.  Julia is not
installed in the image this repository is built in, so those files exist only once a maintainer has run the generator; until
then the comparisons SKIP with that reason (they never silently pass) and what runs is

  * the format round trip and the bit-equality of the exported inputs with the fixtures the Python tests read,
  * a static check that the loader expects exactly the fields the generator writes,
  * the comparison code itself on a stand-in file written from the oracle's own outputs (so that the first real file does not
    meet an untested loader).

With the files present: the CPU oracle (-m "not gpu") and the HIP path through the C ABI (-m gpu) are compared with the reference's
numbers at the SURVEY §8(c) tolerances — that is the pin that lifts "parity unpinned"."""
import glob
import re
from pathlib import Path

import numpy as np
import pytest

from oracle import gp_oracle as o
from tests.golden import gpb

HERE = Path(__file__).resolve().parent / "golden"
INPUTS = sorted(glob.glob(str(HERE / "julia_inputs" / "*.gpb")))
JULIA = {Path(p).stem: p for p in glob.glob(str(HERE / "julia" / "*.gpb"))}
SKIP = ("tests/golden/julia/{}.gpb is absent: run `julia tests/golden/make_golden.jl` with AbstractGPs.jl installed and commit its "
        "output (Julia is not available in the build image) — reference-pinned parity stays UNPINNED until then")

# field -> (kind of comparison, tolerance): "rel" scalar / per-entry relative, "relnorm" 2-norm relative, "abs" max abs
TOL_ORACLE = {
    "logpdf": ("rel", 1e-11), "logpdf_Y": ("rel", 1e-11), "alpha": ("relnorm", 1e-9), "delta": ("abs", 0.0), "U": ("abs", 1e-11),
    "post_mean": ("abs", 1e-9), "post_var": ("abs", 1e-10), "post_cov": ("abs", 1e-10), "post_cross_cov": ("abs", 1e-10),
    "seq_alpha": ("relnorm", 1e-8), "seq_U": ("abs", 1e-9),
    "elbo": ("rel", 1e-9), "dtc": ("rel", 1e-9), "vfe_alpha": ("relnorm", 1e-6), "vfe_m_eps": ("relnorm", 1e-7),
    "vfe_Lam_U": ("abs", 1e-7), "vfe_U": ("abs", 1e-9), "vfe_b_y": ("abs", 1e-12),
    "vfe_mean": ("abs", 1e-8), "vfe_var": ("abs", 1e-8), "vfe_cov": ("abs", 1e-8),
    "elbo_dir": ("relnorm", 1e-4),   # the reference side of this field is a central difference (t = 1e-4) of its own elbo
    "upd_obs_alpha": ("relnorm", 1e-5), "upd_obs_m_eps": ("relnorm", 1e-6), "upd_obs_mean": ("abs", 1e-7), "upd_obs_var": ("abs", 1e-7),
    "upd_z_alpha": ("relnorm", 1e-4), "upd_z_m_eps": ("relnorm", 1e-5), "upd_z_mean": ("abs", 1e-6), "upd_z_var": ("abs", 1e-6),
}
# the device against the reference: SURVEY §8(c) — logpdf rel 1e-10, α 1e-8, mean 1e-8, var 1e-9; sparse side as tests/test_gpu_parity.py
TOL_DEVICE = dict(TOL_ORACLE)
TOL_DEVICE.update({"logpdf": ("rel", 1e-10), "logpdf_Y": ("rel", 1e-10), "alpha": ("relnorm", 1e-8), "U": ("abs", 1e-10),
                   "post_mean": ("abs", 1e-8), "post_var": ("abs", 1e-9), "post_cov": ("abs", 1e-9), "post_cross_cov": ("abs", 1e-9),
                   "seq_alpha": ("relnorm", 1e-8), "seq_U": ("abs", 1e-9), "elbo": ("rel", 1e-8), "dtc": ("rel", 1e-8),
                   "vfe_alpha": ("relnorm", 1e-5), "vfe_m_eps": ("relnorm", 1e-6), "vfe_Lam_U": ("abs", 1e-6), "vfe_U": ("abs", 1e-8),
                   "vfe_mean": ("abs", 1e-7), "vfe_var": ("abs", 1e-7), "vfe_cov": ("abs", 1e-7)})


def _case(inp):
    scale = inp["scale"]
    scale = None if np.all(np.isnan(scale)) else (float(scale) if np.ndim(scale) == 0 else np.asarray(scale))
    mean = None if np.isnan(inp["mean"]) else float(inp["mean"])
    s2 = float(inp["sigma2"]) if np.ndim(inp["sigma2"]) == 0 else np.asarray(inp["sigma2"])
    return int(inp["kind"]), float(inp["variance"]), scale, mean, s2


def _noise(s2, sl):
    return s2 if np.ndim(s2) == 0 else s2[sl]


def _nan_fields(m: int, ns: int) -> dict:
    return {"upd_z_alpha": np.full(m, np.nan), "upd_z_m_eps": np.full(m, np.nan), "upd_z_mean": np.full(ns, np.nan),
            "upd_z_var": np.full(ns, np.nan)}


def _elbo_dir(g, var, scale, s2, z) -> np.ndarray:
    """The analytic ELBO gradient contracted with the generator's four directions: variance·(1 + t), transform parameters·(1 + t), noise·(1 + t),
    z + t·D with D_ij = sin(i + 3j) (1-based, as make_golden.jl builds it)."""
    zz = np.asarray(z, dtype=np.float64)
    m = zz.shape[0]
    i = np.arange(1, m + 1, dtype=np.float64)
    D = np.sin(i + 3.0) if zz.ndim == 1 else np.sin(i[:, None] + 3.0 * np.arange(1, zz.shape[1] + 1, dtype=np.float64)[None, :])
    d_scale = 0.0 if scale is None else float(np.sum(np.asarray(g["scale"]) * np.asarray(scale)))
    noise = g["noise_diag"] if "noise_diag" in g and np.ndim(s2) else g["noise"]
    return np.array([g["variance"] * var, d_scale, float(np.sum(np.asarray(noise) * np.asarray(s2))), float(np.sum(np.asarray(g["z"]) * D))])


def oracle_outputs(inp) -> dict:
    kind, var, scale, mean, s2 = _case(inp)
    f = o.GP(o.Kernel(kind, var, scale), mean)
    x, y, Y, xs, z, jitter = inp["x"], inp["y"], inp["Y"], inp["xs"], inp["z"], float(inp["jitter"])
    n, n1, m1 = x.shape[0], int(inp["n1"]), int(inp["m1"])
    fx = o.FiniteGP(f, x, s2)
    out = {"logpdf": o.logpdf(fx, y), "logpdf_Y": o.logpdf(fx, Y)}
    post = o.posterior(fx, y)
    out.update(alpha=post.alpha, delta=post.delta, U=np.triu(post.U))
    out["post_mean"], out["post_var"] = post.mean_and_var(xs)
    out["post_cov"] = post.cov(xs)
    out["post_cross_cov"] = post.cov(xs, x[:min(n, 7)])
    p1 = o.posterior(o.FiniteGP(f, x[:n1], _noise(s2, slice(0, n1))), y[:n1])
    p2 = o.posterior(o.FiniteGP(p1, x[n1:], _noise(s2, slice(n1, n))), y[n1:])
    out.update(seq_alpha=p2.alpha, seq_U=np.triu(p2.U))
    out["elbo"] = o.elbo(f, z, jitter, fx, y)
    out["dtc"] = o.dtc_log_evidence(f, z, jitter, fx, y)
    ap = o.vfe_posterior(f, z, jitter, fx, y)
    out.update(vfe_alpha=ap.alpha, vfe_m_eps=ap.m_eps, vfe_Lam_U=np.triu(ap.Lam_U), vfe_U=np.triu(ap.U), vfe_b_y=ap.b_y)
    out["vfe_mean"], out["vfe_var"] = ap.mean_and_var(xs)
    out["vfe_cov"] = ap.cov(xs)
    out["elbo_dir"] = _elbo_dir(o.elbo_grad(f, z, jitter, fx, y), var, scale, s2, z)
    a1 = o.vfe_posterior(f, z, jitter, o.FiniteGP(f, x[:n1], _noise(s2, slice(0, n1))), y[:n1])
    a2 = o.vfe_update_obs(a1, o.FiniteGP(f, x[n1:], _noise(s2, slice(n1, n))), y[n1:])
    out.update(upd_obs_alpha=a2.alpha, upd_obs_m_eps=a2.m_eps)
    out["upd_obs_mean"], out["upd_obs_var"] = a2.mean_and_var(xs)
    b1 = o.vfe_posterior(f, z[:m1], jitter, fx, y)
    try:
        b2 = o.vfe_update_z(b1, z[m1:])
        out.update(upd_z_alpha=b2.alpha, upd_z_m_eps=b2.m_eps)
        out["upd_z_mean"], out["upd_z_var"] = b2.mean_and_var(xs)
    except o.PosDefException:
        # the reference puts NO jitter on the new block C22 (src/sparse_approximations.jl:138), so appending pseudo-points to a
        # near-singular K_zz throws PosDefException in update_chol — in AbstractGPs.jl, in the oracle and on the device alike;
        # the generator records that outcome as NaN fields (make_golden.jl does the same) and the comparison requires it of both sides
        out.update(_nan_fields(z.shape[0], xs.shape[0]))
    return out


def device_outputs(agp, inp) -> dict:
    kind, var, scale, mean, s2 = _case(inp)
    k = var * agp.Kernel(kind)
    if scale is not None:
        k = k @ (agp.ScaleTransform(scale) if np.ndim(scale) == 0 else agp.ARDTransform(scale))
    f = agp.GP(k) if mean is None else agp.GP(mean, k)
    x, y, Y, xs, z, jitter = inp["x"], inp["y"], inp["Y"], inp["xs"], inp["z"], float(inp["jitter"])
    n, n1, m1 = x.shape[0], int(inp["n1"]), int(inp["m1"])
    fx = f(x, s2)
    out = {"logpdf": agp.logpdf(fx, y), "logpdf_Y": agp.logpdf(fx, Y)}
    post = agp.posterior(fx, y)
    out.update(alpha=post.data.alpha, delta=post.data.delta, U=np.triu(post.data.C.U))
    out["post_mean"], out["post_var"] = post.mean_and_var(xs)
    out["post_cov"] = post.cov(xs)
    out["post_cross_cov"] = post.cov(xs, x[:min(n, 7)])
    p1 = agp.posterior(f(x[:n1], _noise(s2, slice(0, n1))), y[:n1])
    p2 = agp.posterior(p1(x[n1:], _noise(s2, slice(n1, n))), y[n1:])
    out.update(seq_alpha=p2.data.alpha, seq_U=np.triu(p2.data.C.U))
    vfe = agp.VFE(f(z, jitter))
    out["elbo"] = agp.elbo(vfe, fx, y)
    out["dtc"] = agp.approx_log_evidence(agp.DTC(vfe.fz), fx, y)
    ap = agp.posterior(vfe, fx, y)
    d = ap.data
    out.update(vfe_alpha=d["alpha"], vfe_m_eps=d["m_eps"], vfe_Lam_U=d["Lam_U"], vfe_U=d["U"], vfe_b_y=d["b_y"])
    out["vfe_mean"], out["vfe_var"] = ap.mean_and_var(xs)
    out["vfe_cov"] = ap.cov(xs)
    out["elbo_dir"] = _elbo_dir(ap.objective_grad(), var, scale, s2, z)
    a1 = agp.posterior(vfe, f(x[:n1], _noise(s2, slice(0, n1))), y[:n1])
    a2 = agp.update_posterior(a1, f(x[n1:], _noise(s2, slice(n1, n))), y[n1:])
    out.update(upd_obs_alpha=a2.data["alpha"], upd_obs_m_eps=a2.data["m_eps"])
    out["upd_obs_mean"], out["upd_obs_var"] = a2.mean_and_var(xs)
    b1 = agp.posterior(agp.VFE(f(z[:m1], jitter)), fx, y)
    try:
        b2 = agp.update_posterior(b1, f(z[m1:], jitter))
        out.update(upd_z_alpha=b2.data["alpha"], upd_z_m_eps=b2.data["m_eps"])
        out["upd_z_mean"], out["upd_z_var"] = b2.mean_and_var(xs)
    except agp.PosDefException:
        out.update(_nan_fields(z.shape[0], xs.shape[0]))
    return out


def compare(ref: dict, mine: dict, tol: dict, who: str) -> None:
    assert set(ref) == set(tol), set(ref) ^ set(tol)
    for name, (kind, t) in tol.items():
        a, b = np.asarray(mine[name], dtype=np.float64), np.asarray(ref[name], dtype=np.float64)
        assert a.shape == b.shape, (who, name, a.shape, b.shape)
        if b.size and np.all(np.isnan(b)):
            # the reference side threw PosDefException at this step (a Schur complement that is singular to rounding): whether the
            # other side's last pivot lands at +1e-9 or −1e-9 is rounding, so either outcome is accepted and there is nothing to compare
            continue
        assert not (a.size and np.all(np.isnan(a))), f"{who} vs AbstractGPs.jl: {name}: PosDefException where the reference has values"
        if kind == "rel":
            err = float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-300)))
        elif kind == "relnorm":
            err = float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))
        else:
            err = float(np.max(np.abs(a - b))) if a.size else 0.0
        assert err <= t, f"{who} vs AbstractGPs.jl: {name} {kind} error {err:.3e} > {t:.1e}"


def generator_fields() -> list:
    src = (HERE / "make_golden.jl").read_text()
    return re.findall(r'push!\(out, "([a-z_A-Z0-9]+)" =>', src)


def test_loader_expects_exactly_what_the_generator_writes():
    fields = generator_fields()
    assert len(fields) == len(set(fields))
    assert set(fields) == set(TOL_ORACLE) == set(TOL_DEVICE)


@pytest.mark.parametrize("path", INPUTS, ids=lambda p: Path(p).stem)
def test_exported_inputs_are_the_fixture_bits(path):
    """make_golden.jl reads the same bits the Python oracle / device tests read from tests/golden/<case>.npz."""
    inp = gpb.read(path)
    g = np.load(HERE / (Path(path).stem + ".npz"))
    for k in ("x", "y", "Y", "xs", "kind", "variance", "scale", "sigma2", "mean", "z", "jitter"):
        assert np.array_equal(np.asarray(inp[k]), np.asarray(g[k], dtype=np.float64), equal_nan=True), k
    assert 2 <= int(inp["n1"]) < g["x"].shape[0] and 2 <= int(inp["m1"]) < g["z"].shape[0]


def test_comparison_code_on_a_stand_in_file(tmp_path):
    """The loader + comparison on a file in the generator's schema (written here from the oracle, NOT a reference pin): the tolerance
    table, shapes (column-major matrices, scalars) and field names are exercised before a real Julia file exists; a perturbed field
    must fail."""
    inp = gpb.read(INPUTS[-1])
    mine = oracle_outputs(inp)
    gpb.write(tmp_path / "standin.gpb", {k: mine[k] for k in generator_fields()})
    ref = gpb.read(tmp_path / "standin.gpb")
    compare(ref, mine, TOL_ORACLE, "oracle")
    bad = dict(mine)
    bad["post_var"] = mine["post_var"] + 1e-6
    with pytest.raises(AssertionError, match="post_var"):
        compare(ref, bad, TOL_ORACLE, "oracle")


@pytest.mark.parametrize("path", INPUTS, ids=lambda p: Path(p).stem)
def test_oracle_vs_julia_reference(path):
    name = Path(path).stem
    if name not in JULIA:
        pytest.skip(SKIP.format(name))
    compare(gpb.read(JULIA[name]), oracle_outputs(gpb.read(path)), TOL_ORACLE, "oracle")


@pytest.mark.gpu
@pytest.mark.parametrize("path", INPUTS, ids=lambda p: Path(p).stem)
def test_device_vs_julia_reference(agp, path):
    name = Path(path).stem
    if name not in JULIA:
        pytest.skip(SKIP.format(name))
    compare(gpb.read(JULIA[name]), device_outputs(agp, gpb.read(path)), TOL_DEVICE, "device")


@pytest.mark.gpu
@pytest.mark.parametrize("path", INPUTS[:3], ids=lambda p: Path(p).stem)
def test_device_vs_oracle_in_the_generator_schema(agp, path):
    """Every field the Julia generator writes, device against the oracle (runs with or without the Julia files): covers the
    fields the other GPU tests do not read — Λ_ε.U, U and b_y of the sparse cache through gp_vfe_get_factors / gp_vfe_get_by
    (test/sparse_approximations.jl:48-55), the cross-covariance, the sequential factor."""
    inp = gpb.read(path)
    compare(oracle_outputs(inp), device_outputs(agp, inp), TOL_DEVICE, "device(vs oracle)")
