"""Gradient of the sparse objective on the device (gp_vfe_grad through the API mirror's `objective_grad` / `elbo_and_grad`) against the oracle's
`elbo_grad` — dense N×N matrix calculus on the textbook form (oracle/gp_oracle.py), itself checked against central differences of the oracle's
`elbo` / `dtc_log_evidence` in tests/test_oracle.py.  The reference has no hand-written adjoint: its users differentiate
`elbo(VFE(f(z, jitter)), f(x, Σy), y)` by AD or finite differences (examples/0-intro-1d/script.jl:385-394), so the pinned quantity is the
derivative of the pinned value.

Tolerances: fp64 — every component within 1e-7 of the oracle's, relative to the largest component of that block (measured ≈ 1e-10);
fp32 handles — the backward pass runs in fp64 on the handle's fp32-rounded inputs and its fp32-streamed M×M state: within 2e-3 on the same scale against
the fp64 oracle (the objective itself is held to rel 1e-4 in fp32), except ∂/∂z, which an fp32 handle refuses (include/gpmi355.h gp_vfe_grad)."""
import numpy as np
import pytest

from oracle import gp_oracle as o

pytestmark = pytest.mark.gpu


def _kernels(agp, kind, tr, d, var):
    k = var * agp.Kernel(kind)
    if tr == "none":
        return k, o.Kernel(kind, var, None)
    if tr == "scale":
        return k @ agp.ScaleTransform(0.8), o.Kernel(kind, var, 0.8)
    v = np.linspace(0.5, 1.3, d)
    return k @ agp.ARDTransform(v), o.Kernel(kind, var, v)


def _data(n, m, d, seed, vector_noise):
    rng = np.random.default_rng(seed)
    X = rng.normal(size=(n, d))
    Z = rng.normal(size=(m, d))
    y = np.sin(X.sum(1)) + 0.1 * rng.normal(size=n)
    s2 = (0.05 + 0.1 * rng.random(n)) if vector_noise else 0.08
    return X, Z, y, s2


def _close(a, b, tol, what):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    if a.size == 1 and b.size == 1:  # an ARDTransform of one dimension marshals as one scale: the mirror returns a float, the oracle a length-1 vector
        a, b = a.reshape(()), b.reshape(())
    assert a.shape == b.shape, (what, a.shape, b.shape)
    err = np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1.0)
    assert err <= tol, (what, err)
    return err


def _compare(g, ref, tol, vector_noise, wrt_x=True):
    _close(g["variance"], ref["variance"], tol, "variance")
    if ref["scale"] is None:
        assert g["scale"] is None
    else:
        _close(g["scale"], ref["scale"], tol, "scale")
    if vector_noise:
        _close(g["noise_diag"], ref["noise"], tol, "noise_diag")
        _close(g["noise"], np.sum(ref["noise"]), tol, "noise sum")
    else:
        _close(g["noise"], ref["noise"], tol, "noise")
    _close(g["y"], ref["y"], tol, "y")
    _close(g["mean"], ref["mean"], tol, "mean")
    _close(g["z"], ref["z"], tol, "z")
    if wrt_x:
        _close(g["x"], ref["x"], tol, "x")


@pytest.mark.parametrize("approx", ["VFE", "DTC"])
@pytest.mark.parametrize("vector_noise", [False, True])
@pytest.mark.parametrize("tr", ["none", "scale", "ard"])
@pytest.mark.parametrize("kind", [0, 1, 2, 3])
def test_objective_grad_vs_oracle(agp, kind, tr, vector_noise, approx):
    n, m, d = 700, 45, 3
    X, Z, y, s2 = _data(n, m, d, 100 + kind, vector_noise)
    k, ok = _kernels(agp, kind, tr, d, 1.3)
    f, of = agp.GP(k), o.GP(ok)
    A = agp.VFE if approx == "VFE" else agp.DTC
    val, g = agp.elbo_and_grad(A(f(agp.RowVecs(Z), 1e-4)), f(agp.RowVecs(X), s2), y, wrt_x=True)
    ofx = o.FiniteGP(of, X, s2)
    ref_val = o.elbo(of, Z, 1e-4, ofx, y) if approx == "VFE" else o.dtc_log_evidence(of, Z, 1e-4, ofx, y)
    assert val == pytest.approx(ref_val, rel=1e-9)
    _compare(g, o.elbo_grad(of, Z, 1e-4, ofx, y, vfe=(approx == "VFE")), 1e-7, vector_noise)


@pytest.mark.parametrize("d,tr", [(1, "scale"), (6, "ard"), (20, "ard"), (20, "scale"), (17, "none")])
def test_objective_grad_any_input_dimension(agp, d, tr):
    """D = 1 as plain vectors; D = 6 takes the 16-dimension instance in one launch; D = 17 / 20 take two launches per tile (the per-dimension sums of
    dimensions 16.. come from the second)."""
    n, m = 520, 33
    X, Z, y, s2 = _data(n, m, d, 7 + d, True)
    X, Z = 0.5 * X, 0.5 * Z
    k, ok = _kernels(agp, 3, tr, d, 0.9)
    f, of = agp.GP(k), o.GP(ok)
    xin, zin = (X[:, 0], Z[:, 0]) if d == 1 else (agp.RowVecs(X), agp.RowVecs(Z))
    val, g = agp.elbo_and_grad(agp.VFE(f(zin, 1e-4)), f(xin, s2), y, wrt_x=True)
    oX, oZ = (X[:, 0], Z[:, 0]) if d == 1 else (X, Z)
    ofx = o.FiniteGP(of, oX, s2)
    assert val == pytest.approx(o.elbo(of, oZ, 1e-4, ofx, y), rel=1e-9)
    _compare(g, o.elbo_grad(of, oZ, 1e-4, ofx, y), 1e-7, True)


def test_objective_grad_containers_and_prior_mean(agp):
    """ColVecs / RowVecs for x and z in either memory order: "z" and "x" come back in the shape of the container's array; a constant prior mean
    shifts δ only ("mean" = −"y")."""
    n, m, d = 300, 20, 2
    X, Z, y, s2 = _data(n, m, d, 3, False)
    k, ok = _kernels(agp, 0, "ard", d, 1.1)
    f, of = agp.GP(0.3, k), o.GP(ok, 0.3)
    ref = o.elbo_grad(of, Z, 1e-3, o.FiniteGP(of, X, s2), y)
    for xin, zin, tx, tz in ((agp.RowVecs(X), agp.RowVecs(Z), False, False), (agp.ColVecs(np.ascontiguousarray(X.T)), agp.ColVecs(np.ascontiguousarray(Z.T)), True, True),
                             (agp.RowVecs(np.asfortranarray(X)), agp.ColVecs(np.asfortranarray(Z.T)), False, True)):
        _, g = agp.elbo_and_grad(agp.VFE(f(zin, 1e-3)), f(xin, s2), y, wrt_x=True)
        _close(g["x"].T if tx else g["x"], ref["x"], 1e-7, "x")
        _close(g["z"].T if tz else g["z"], ref["z"], 1e-7, "z")
        _close(g["variance"], ref["variance"], 1e-7, "variance")
        _close(g["mean"], ref["mean"], 1e-7, "mean")


def test_objective_grad_after_updates(agp):
    """The handle after update_posterior with new observations (second batch with its own noise) and after appended pseudo-points: the gradient of the
    objective THAT posterior reports — for the append, the reference's route gives the new pseudo-points no jitter (src/sparse_approximations.jl:138), which
    the oracle reproduces with a vector jitter."""
    n1, n2, m1, m2, d = 400, 250, 25, 10, 2
    X, Z, y, _ = _data(n1 + n2, m1 + m2, d, 11, False)
    rng = np.random.default_rng(5)
    s2 = np.concatenate([np.full(n1, 0.07), 0.05 + 0.1 * rng.random(n2)])
    k, ok = _kernels(agp, 2, "scale", d, 1.2)
    f, of = agp.GP(k), o.GP(ok)
    post = agp.posterior(agp.VFE(f(agp.RowVecs(Z[:m1]), 1e-4)), f(agp.RowVecs(X[:n1]), 0.07), y[:n1])
    post = agp.update_posterior(post, f(agp.RowVecs(X[n1:]), s2[n1:]), y[n1:])
    g = post.objective_grad(wrt_x=True)
    ofx = o.FiniteGP(of, X, s2)
    assert post.objective == pytest.approx(o.elbo(of, Z[:m1], 1e-4, ofx, y), rel=1e-9)
    _compare(g, o.elbo_grad(of, Z[:m1], 1e-4, ofx, y), 1e-7, True)
    post2 = agp.update_posterior(post, f(agp.RowVecs(Z[m1:]), 1e-4))
    g2 = post2.objective_grad(wrt_x=True)
    jit = np.concatenate([np.full(m1, 1e-4), np.zeros(m2)])
    assert post2.objective == pytest.approx(o.elbo(of, Z, jit, ofx, y), rel=1e-8)
    _compare(g2, o.elbo_grad(of, Z, jit, ofx, y), 1e-6, True)


@pytest.mark.parametrize("approx,chunk", [("VFE", 16384), ("DTC", 16384), ("VFE", 0)])
def test_objective_grad_streams_several_chunks(agp, approx, chunk):
    """N = 40 000, M = 700 (three chunks of 16 384 with a ragged tail, M padded to 768): the gradient along a random direction in (variance, scale, noise, z)
    against a central difference of the device's own objective (rel 1e-4), and the cheap identities Σ_i ∂/∂y_i·1 = −Σ "mean"."""
    n, m, d = 40000, 700, 3
    rng = np.random.default_rng(21)
    X = rng.normal(size=(n, d))
    Z = X[rng.choice(n, m, replace=False)] + 0.01 * rng.normal(size=(m, d))
    y = np.sin(X.sum(1)) + 0.1 * rng.normal(size=n)
    A = agp.VFE if approx == "VFE" else agp.DTC
    ctx = agp.Context(0)   # a private context: "vfe_chunk" = 16 384 keeps the three chunks (0, the default, streams this batch as one chunk of 49 152)
    ctx.set_param("vfe_chunk", chunk)

    def obj(var, sc, s2, Zc):
        f = agp.GP(var * agp.Matern52Kernel() @ agp.ScaleTransform(sc), ctx=ctx)
        return agp.approx_log_evidence(A(f(agp.RowVecs(Zc), 1e-4)), f(agp.RowVecs(X), s2), y)

    var, sc, s2 = 1.2, 0.7, 0.1
    f = agp.GP(var * agp.Matern52Kernel() @ agp.ScaleTransform(sc), ctx=ctx)
    val, g = agp.elbo_and_grad(A(f(agp.RowVecs(Z), 1e-4)), f(agp.RowVecs(X), s2), y)
    assert val == pytest.approx(obj(var, sc, s2, Z), rel=1e-12)
    dZ = rng.normal(size=Z.shape)
    dirs = np.array([0.3, -0.2, 0.05])
    h = 1e-5
    fd = (obj(var + h * dirs[0], sc + h * dirs[1], s2 + h * dirs[2], Z + h * dZ) - obj(var - h * dirs[0], sc - h * dirs[1], s2 - h * dirs[2], Z - h * dZ)) / (2 * h)
    an = g["variance"] * dirs[0] + g["scale"] * dirs[1] + g["noise"] * dirs[2] + float(np.sum(g["z"] * dZ))
    assert an == pytest.approx(fd, rel=1e-4), (an, fd)   # the difference quotient itself is good to ≈ 1e-5 here (h = 1e-4 … 1e-6 scatter that much around it)
    assert np.sum(g["noise_diag"]) == pytest.approx(g["noise"], rel=1e-9)
    ctx.close()


def test_automatic_chunk_for_few_pseudo_points(agp):
    """"vfe_chunk" = 0 (the default): M = 64 pseudo-points stream N = 300 000 observations in chunks of 262 144 (16 × 16 384; the second chunk ragged) — objective
    against the oracle (fp64 1e-9, fp32 1e-4), the gradient along a direction against differences of fits (rel 1e-4), an update with 40 000 more observations (the handle keeps
    its chunk), and the same numbers from a context pinned to 16 384-point chunks (19 chunks)."""
    n, n2, m, d = 300000, 40000, 64, 2
    rng = np.random.default_rng(8)
    X = rng.normal(size=(n + n2, d))
    y = np.sin(X.sum(1)) + 0.2 * rng.normal(size=n + n2)
    Z = rng.normal(size=(m, d))
    of = o.GP(o.Kernel(o.MATERN32, 1.1, 0.9))
    ref = o.elbo(of, Z, 1e-4, o.FiniteGP(of, X[:n], 0.1), y[:n])
    ref2 = o.elbo(of, Z, 1e-4, o.FiniteGP(of, X, 0.1), y)
    vals = {}
    for chunk in (0, 16384):
        ctx = agp.Context(0)
        ctx.set_param("vfe_chunk", chunk)
        try:
            def build(var, sc, s2, Zc, dt=np.float64):
                f = agp.GP(var * agp.Matern32Kernel() @ agp.ScaleTransform(sc), ctx=ctx)
                return agp.VFE(f(agp.RowVecs(Zc.astype(dt)), 1e-4)), f(agp.RowVecs(X[:n].astype(dt)), dt(s2))

            post = agp.posterior(*build(1.1, 0.9, 0.1, Z), y[:n])
            assert float(post.objective) == pytest.approx(ref, rel=1e-9)
            g = post.objective_grad()
            vals[chunk] = (float(post.objective), g["variance"], g["scale"], g["noise"], g["z"].copy())
            if chunk == 0:
                dZ, dirs, h = rng.normal(size=Z.shape), np.array([0.3, -0.2, 0.05]), 1e-5
                fd = (agp.approx_log_evidence(*build(1.1 + h * dirs[0], 0.9 + h * dirs[1], 0.1 + h * dirs[2], Z + h * dZ), y[:n])
                      - agp.approx_log_evidence(*build(1.1 - h * dirs[0], 0.9 - h * dirs[1], 0.1 - h * dirs[2], Z - h * dZ), y[:n])) / (2 * h)
                an = g["variance"] * dirs[0] + g["scale"] * dirs[1] + g["noise"] * dirs[2] + float(np.sum(g["z"] * dZ))
                assert an == pytest.approx(fd, rel=1e-4), (an, fd)   # the difference quotient itself is good to ≈ 1e-5 here (h = 1e-4 … 1e-6 scatter that much around it)
                post2 = agp.update_posterior(post, post.prior(agp.RowVecs(X[n:]), 0.1), y[n:])
                assert float(post2.objective) == pytest.approx(ref2, rel=1e-9)
                p32 = agp.posterior(*build(1.1, 0.9, 0.1, Z, np.float32), y[:n].astype(np.float32))
                assert float(p32.objective) == pytest.approx(ref, rel=1e-4)
        finally:
            ctx.close()
    a, b = vals[0], vals[16384]
    assert a[0] == pytest.approx(b[0], rel=1e-11)
    for i in (1, 2, 3):
        assert a[i] == pytest.approx(b[i], rel=1e-8)
    assert np.max(np.abs(a[4] - b[4])) <= 1e-8 * np.max(np.abs(b[4]))


def test_objective_grad_fp32(agp):
    """The fp32 handle (BASELINE config C5's dtype): value and gradient against the fp64 oracle."""
    n, m, d = 3000, 120, 3
    X, Z, y, s2 = _data(n, m, d, 33, False)
    X, Z = X.astype(np.float32), Z.astype(np.float32)
    k, ok = _kernels(agp, 0, "scale", d, 1.0)
    f, of = agp.GP(k), o.GP(ok)
    val, g = agp.elbo_and_grad(agp.VFE(f(agp.RowVecs(Z), 1e-4)), f(agp.RowVecs(X), np.float32(0.1)), y.astype(np.float32), wrt_x=True)
    ofx = o.FiniteGP(of, X.astype(np.float64), 0.1)
    assert val == pytest.approx(o.elbo(of, Z.astype(np.float64), 1e-4, ofx, y.astype(np.float32).astype(np.float64)), rel=1e-4)
    ref = o.elbo_grad(of, Z.astype(np.float64), 1e-4, ofx, y.astype(np.float32).astype(np.float64))
    errs = {}
    assert "z" not in g   # ∂/∂z needs an fp64 handle (the C ABI returns −7: below)
    for key in ("variance", "scale", "noise", "y", "x"):
        a, b = np.asarray(g[key], dtype=np.float64), np.asarray(ref[key], dtype=np.float64)
        errs[key] = float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1.0))
    print("fp32 gradient errors", errs)
    assert max(errs.values()) <= 2e-3, errs
    post = agp.posterior(agp.VFE(f(agp.RowVecs(Z), 1e-4)), f(agp.RowVecs(X), np.float32(0.1)), y.astype(np.float32))
    dz = np.empty((m, d))
    assert post._state.ctx.lib.gp_vfe_grad(post._state.handle, None, None, None, None, None, dz.ctypes.data, 1, None, 0) == -7


def test_objective_grad_argument_errors(agp):
    n, m, d = 200, 10, 2
    X, Z, y, s2 = _data(n, m, d, 1, False)
    f = agp.GP(agp.SqExponentialKernel())
    post = agp.posterior(agp.VFE(f(agp.RowVecs(Z), 1e-4)), f(agp.RowVecs(X), s2), y)
    import ctypes as C
    lib = post._state.ctx.lib
    dz = np.empty((m, d))
    assert lib.gp_vfe_grad(post._state.handle, None, None, None, None, None, dz.ctypes.data, 0, None, 0) == -8     # layout 0 needs D = 1
    assert lib.gp_vfe_grad(post._state.handle, None, None, None, None, None, dz.ctypes.data, 3, None, 0) == -8
    assert lib.gp_vfe_grad(post._state.handle, None, None, None, None, None, None, 0, dz.ctypes.data, 7) == -10
    assert lib.gp_vfe_grad(None, None, None, None, None, None, None, 0, None, 0) == -1
    assert lib.gp_vfe_grad(post._state.handle, None, None, None, None, None, None, 0, None, 0) == 0                # every output may be NULL
    with pytest.raises(TypeError):
        agp.elbo_and_grad(agp.ExactInference(), f(agp.RowVecs(X), s2), y)


# GPMI_TEST_RANDOM_CASES / GPMI_TEST_RANDOM_SEED: as in tests/test_gpu_random.py
_NCASES = max(1, int(__import__("os").environ.get("GPMI_TEST_RANDOM_CASES", "24")) // 2)
_SEED0 = int(__import__("os").environ.get("GPMI_TEST_RANDOM_SEED", "1000")) + 1700000


@pytest.mark.parametrize("case", range(_NCASES))
def test_objective_grad_random_sweep(agp, case):
    """Seeded random configurations (sizes around the 128-tile and padding edges, every kernel / transform / noise form, VFE and DTC, containers, prior
    means, jitter 1e-6 … 1e-3): value and every gradient block against the oracle."""
    rng = np.random.default_rng(_SEED0 + case)
    n = int(rng.choice([1, 2, 127, 128, 129, 300, 511, 640, 1000, 1700]))
    m = int(rng.choice([1, 2, 17, 64, 127, 128, 129, 200]))
    d = int(rng.integers(1, 6))
    kind = int(rng.integers(0, 4))
    variance = float(rng.uniform(0.3, 2.5))
    tr = int(rng.integers(0, 3))
    scale = None if tr == 0 else (float(rng.uniform(0.4, 1.6)) if tr == 1 else rng.uniform(0.4, 1.6, d))
    X, Z = rng.standard_normal((n, d)), rng.standard_normal((m, d))
    y = np.sin(X.sum(1)) + 0.1 * rng.standard_normal(n)
    vec = rng.random() < 0.5
    s2 = rng.uniform(0.03, 0.3, n) if vec else float(rng.uniform(0.03, 0.3))
    mean = None if rng.random() < 0.5 else float(rng.normal())
    jit = float(10.0 ** rng.uniform(-6, -3))
    vfe = rng.random() < 0.7
    kern = variance * agp.Kernel(kind)
    if scale is not None:
        kern = kern @ (agp.ScaleTransform(scale) if np.ndim(scale) == 0 else agp.ARDTransform(scale))
    f = agp.GP(kern) if mean is None else agp.GP(mean, kern)
    of = o.GP(o.Kernel(kind, variance, scale), mean)
    col = rng.random() < 0.5
    wrap = (lambda a: agp.ColVecs(np.ascontiguousarray(a.T))) if col else agp.RowVecs
    desc = f"n={n} m={m} d={d} kind={kind} tr={tr} mean={mean} noise={'vec' if vec else 'scalar'} col={col} jitter={jit:.1e} vfe={vfe}"
    A = agp.VFE if vfe else agp.DTC
    val, g = agp.elbo_and_grad(A(f(wrap(Z), jit)), f(wrap(X), s2), y, wrt_x=True)
    ofx = o.FiniteGP(of, X, s2)
    ref_val = o.elbo(of, Z, jit, ofx, y) if vfe else o.dtc_log_evidence(of, Z, jit, ofx, y)
    assert val == pytest.approx(ref_val, rel=1e-8, abs=1e-8), desc
    ref = o.elbo_grad(of, Z, jit, ofx, y, vfe)
    tol = max(1e-7, 1e-11 / jit)                # K_zz⁻¹ reaches 1/jitter: both sides lose digits with it (900 cases in profiles/r6/random_sweep_vfe_grad.log)
    try:
        g = dict(g, x=g["x"].T if col else g["x"], z=g["z"].T if col else g["z"])
        _compare(g, ref, tol, vec)
    except AssertionError as e:
        raise AssertionError(f"{desc}: {e}") from None


def test_objective_grad_at_c5_size(agp):
    """BASELINE config C5's size (N = 262 144, M = 4 096, D = 3) on an fp64 posterior: the gradient along a random direction in (variance, scale, noise, z)
    against a central difference of two fits (rel 1e-4; measured 4e-6), the fp32 posterior's hyper-parameter / noise gradients against the fp64 one's (5e-3;
    measured 1.4e-3), and ∂/∂z refused on the fp32 posterior."""
    n, m, d = 262144, 4096, 3
    rng = np.random.default_rng(5)
    X = rng.normal(size=(n, d))
    y = np.sin(X.sum(1)) + 0.1 * rng.normal(size=n)
    Z = X[rng.choice(n, m, replace=False)].copy()

    def build(var, sc, s2, Zc, dt=np.float64):
        f = agp.GP(var * agp.SqExponentialKernel() @ agp.ScaleTransform(sc))
        return agp.VFE(f(agp.RowVecs(Zc.astype(dt)), 1e-4)), f(agp.RowVecs(X.astype(dt)), dt(s2))

    val, g = agp.elbo_and_grad(*build(1.0, 1.0, 0.1, Z), y)
    dZ, dirs, h = rng.normal(size=Z.shape), np.array([0.3, -0.2, 0.05]), 1e-5
    fd = (agp.approx_log_evidence(*build(1 + h * dirs[0], 1 + h * dirs[1], 0.1 + h * dirs[2], Z + h * dZ), y)
          - agp.approx_log_evidence(*build(1 - h * dirs[0], 1 - h * dirs[1], 0.1 - h * dirs[2], Z - h * dZ), y)) / (2 * h)
    an = g["variance"] * dirs[0] + g["scale"] * dirs[1] + g["noise"] * dirs[2] + float(np.sum(g["z"] * dZ))
    assert an == pytest.approx(fd, rel=1e-4), (an, fd)
    val32, g32 = agp.elbo_and_grad(*build(1.0, 1.0, 0.1, Z, np.float32), y.astype(np.float32))
    assert float(val32) == pytest.approx(float(val), rel=1e-4) and "z" not in g32
    for key in ("variance", "scale", "noise"):
        assert g32[key] == pytest.approx(g[key], rel=5e-3), key
    assert np.max(np.abs(g32["y"] - g["y"])) <= 5e-3 * np.max(np.abs(g["y"]))


def test_lbfgs_on_the_elbo_like_the_reference_example(agp):
    """examples/0-intro-1d/script.jl:359-420 on the device (tools/train_sparse_example.py): LBFGS over softplus(variance), softplus(inverse lengthscale) and
    logistic(pseudo-points) with `elbo_and_grad` as value / gradient.  The optimiser must make progress with these gradients (a wrong chain would stall the
    line search), and the optimum's value must be the oracle's ELBO at the optimum's parameters."""
    import importlib.util
    from pathlib import Path

    from scipy.optimize import minimize

    spec = importlib.util.spec_from_file_location("train_sparse_example", Path(__file__).resolve().parent.parent / "tools" / "train_sparse_example.py")
    ex = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ex)
    rng = np.random.default_rng(0)
    n, m = 3000, 10
    x = rng.random(n)
    y = np.sin(4 * np.pi * x) + np.cos(11 * x) * x + 0.3 * rng.standard_normal(n)
    calls = []
    fun = ex.make_objective(x, y, 0.09, calls=calls)
    p0 = rng.random(2 + m)
    v0 = -fun(p0)[0]
    res = minimize(fun, p0, jac=True, method="L-BFGS-B", options={"maxiter": 30})
    v1 = -float(res.fun)
    assert v1 > v0 + 100.0, (v0, v1)
    var, sc, z = ex.softplus(res.x[0]), ex.softplus(res.x[1]), ex.logistic(res.x[2:])
    of = o.GP(o.Kernel(o.MATERN52, float(var), float(sc)))
    assert v1 == pytest.approx(o.elbo(of, z, 1e-6, o.FiniteGP(of, x, 0.09), y), rel=1e-7)
    # the chain-ruled gradient at the optimum against central differences of the objective in the unconstrained parameters
    g = fun(res.x)[1]
    for i in (0, 1, 2, 2 + m // 2):
        e = np.zeros_like(res.x)
        e[i] = 1e-5
        fd = (fun(res.x + e)[0] - fun(res.x - e)[0]) / 2e-5
        assert g[i] == pytest.approx(fd, rel=1e-3, abs=2e-3), (i, g[i], fd)
