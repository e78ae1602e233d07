"""GPU: each HIP kernel family against a plain fp64 reference of the same op, through the C ABI's
device-level entry points (gpd_*).  torch is used only for device memory / the reference matmul."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def P(t):
    return C.c_void_p(t.data_ptr())


@pytest.fixture(scope="module")
def lib(agp):
    return agp._lib.load()


@pytest.fixture(scope="module")
def h(ctx):
    return ctx.handle


def _sync(lib, h):
    from abstractgps_jl_amd._lib import check

    check(lib.gpd_sync(h))


def test_mfma_f64_lane_maps(lib, h):
    """A=I-style check with ASYMMETRIC operands (cdna_hip_programming.md §3)."""
    rng = np.random.default_rng(0)
    A = rng.standard_normal((16, 4))
    B = rng.standard_normal((4, 16))
    D = np.zeros((16, 16))
    assert lib.gp_probe_mfma_f64(h, A.ctypes.data, B.ctypes.data, D.ctypes.data) == 0
    np.testing.assert_allclose(D, A @ B, rtol=1e-14, atol=1e-14)


@pytest.mark.parametrize("m,n,k", [(128, 128, 16), (256, 128, 64), (64, 64, 64), (192, 64, 128), (320, 448, 272),
                                    (1024, 1024, 1024), (2176, 2304, 48), (4224, 1152, 32)])
def test_gemm_nt_rect(lib, h, m, n, k):
    from abstractgps_jl_amd._lib import check

    g = torch.Generator(device="cuda").manual_seed(m * 7 + n * 3 + k)
    ld = max(n, k) + 32
    rows = max(m, n) + 128  # slack rows for the over-read contract
    A = torch.randn(rows, ld, dtype=torch.float64, device="cuda", generator=g)
    B = torch.randn(rows, ld, dtype=torch.float64, device="cuda", generator=g)
    Cm = torch.randn(rows, ld, dtype=torch.float64, device="cuda", generator=g)
    ref = Cm.clone()
    ref[:m, :n] -= A[:m, :k] @ B[:n, :k].T
    torch.cuda.synchronize()
    check(lib.gpd_gemm_nt(h, P(Cm), ld, P(A), ld, P(B), ld, m, n, k, None, 0, 0))
    _sync(lib, h)
    err = (Cm - ref).abs().max().item()
    assert err <= 1e-12 * max(1.0, k ** 0.5) * 10, err
    # nothing outside the m×n window was touched
    assert torch.equal(Cm[m:], ref[m:]) and torch.equal(Cm[:, n:], ref[:, n:])


@pytest.mark.parametrize("m,n,off,coff", [(256, 256, 0, 0), (320, 192, 64, 64), (512, 128, 128, 128),
                                          (2304, 2304, 0, 0), (2432, 2176, 384, 128), (3200, 2560, 1152, 0),
                                          (2048, 2048, 64, 64)])
def test_gemm_nt_lower_skips_upper(lib, h, m, n, off, coff):
    """lower mode: 64×64 sub-tiles strictly above the diagonal are not updated; everything on/below is.
    The large cases run in the XCD-aware super-tile order (≥ 256 tiles), also as a trapezoid (row0 > col0)."""
    from abstractgps_jl_amd._lib import check, gp_grid

    k = 64
    g = torch.Generator(device="cuda").manual_seed(m + n + off + coff)
    ld = n + 32
    A = torch.randn(m + 128, k + 32, dtype=torch.float64, device="cuda", generator=g)
    B = torch.randn(n + 128, k + 32, dtype=torch.float64, device="cuda", generator=g)
    Cm = torch.zeros(m + 128, ld, dtype=torch.float64, device="cuda")
    full = -(A[:m, :k] @ B[:n, :k].T)
    grid = gp_grid(1, 0, 1, 0, 1, 1)
    torch.cuda.synchronize()
    check(lib.gpd_gemm_nt(h, P(Cm), ld, P(A), k + 32, P(B), k + 32, m, n, k, C.byref(grid), off, coff))
    _sync(lib, h)
    r = torch.arange(m, device="cuda")[:, None] + off
    c = torch.arange(n, device="cuda")[None, :] + coff
    need = c <= r
    got = Cm[:m, :n]
    assert (got - full)[need].abs().max().item() < 1e-11
    skipped = (c // 64) > (r // 64)
    if skipped.any():
        assert got[skipped].abs().max().item() == 0.0


@pytest.mark.parametrize("n,extra", [(64, 0), (64, 192), (128, 64), (192, 128), (256, 0), (1024, 256),
                                      (64, 128 * 700), (128, 128 * 300 + 64)])  # > 256 workgroups: late starters
def test_potrf_and_trsm(lib, h, n, extra):
    from abstractgps_jl_amd._lib import check

    g = torch.Generator(device="cuda").manual_seed(n + extra)
    m = n + extra
    ld = n + 32
    G = torch.randn(n, n, dtype=torch.float64, device="cuda", generator=g)
    S = G @ G.T / n + torch.eye(n, dtype=torch.float64, device="cuda")
    X = torch.randn(extra, n, dtype=torch.float64, device="cuda", generator=g)
    A = torch.full((m + 128, ld), float("nan"), dtype=torch.float64, device="cuda")
    A[:n, :n] = S
    A[n:m, :n] = X
    info = torch.zeros(1, dtype=torch.int32, device="cuda")
    logdet = torch.zeros(1, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    check(lib.gpd_potrf(h, P(A), ld, m, n, P(info), 0, n, P(logdet)))
    _sync(lib, h)
    L = torch.linalg.cholesky(S)
    assert info.item() == 0
    got = torch.tril(A[:n, :n])
    assert (got - L).abs().max().item() < 1e-11
    assert abs(logdet.item() - torch.log(torch.diagonal(L)).sum().item()) < 1e-10
    if extra:
        ref = torch.linalg.solve_triangular(L, X.T, upper=False).T  # X L⁻ᵀ
        assert (A[n:m, :n] - ref).abs().max().item() < 1e-10


def test_potrf_reports_first_bad_pivot(lib, h):
    from abstractgps_jl_amd._lib import check

    n, ld = 256, 288
    A = torch.zeros(n + 128, ld, dtype=torch.float64, device="cuda")
    A[:n, :n] = torch.eye(n, dtype=torch.float64, device="cuda")
    A[130, 130] = -2.0
    info = torch.zeros(1, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    check(lib.gpd_potrf(h, P(A), ld, n, n, P(info), 0, n, None))
    _sync(lib, h)
    assert info.item() == 131  # LAPACK-style 1-based leading-minor order


@pytest.mark.parametrize("m,n", [(64, 64), (192, 256), (384, 1088)])
def test_trsm_rec(lib, h, m, n):
    from abstractgps_jl_amd._lib import check

    g = torch.Generator(device="cuda").manual_seed(m + n)
    G = torch.randn(n, n, dtype=torch.float64, device="cuda", generator=g)
    L = torch.linalg.cholesky(G @ G.T / n + torch.eye(n, dtype=torch.float64, device="cuda"))
    ld = n + 32
    Lp = torch.zeros(n + 128, ld, dtype=torch.float64, device="cuda")
    Lp[:n, :n] = L
    X = torch.randn(m + 128, ld, dtype=torch.float64, device="cuda", generator=g)
    ref = torch.linalg.solve_triangular(L, X[:m, :n].T, upper=False).T
    torch.cuda.synchronize()
    check(lib.gpd_trsm(h, P(X), ld, m, P(Lp), ld, n))
    _sync(lib, h)
    assert (X[:m, :n] - ref).abs().max().item() < 1e-10


@pytest.mark.parametrize("np_,nrhs", [(128, 1), (1024, 2), (2432, 1), (4096, 3)])
def test_trsv_forward_backward(lib, h, np_, nrhs):
    from abstractgps_jl_amd._lib import check

    g = torch.Generator(device="cuda").manual_seed(np_)
    G = torch.randn(np_, np_, dtype=torch.float64, device="cuda", generator=g)
    L = torch.linalg.cholesky(G @ G.T / np_ + torch.eye(np_, dtype=torch.float64, device="cuda"))
    ld = np_ + 32
    Lp = torch.full((np_, ld), float("nan"), dtype=torch.float64, device="cuda")
    Lp[:, :np_] = torch.tril(L) + torch.triu(torch.full_like(L, float("nan")), 1)  # upper part must never be read
    R = torch.randn(nrhs, np_, dtype=torch.float64, device="cuda", generator=g)
    for fwd in (1, 0):
        W = R.clone()
        torch.cuda.synchronize()
        check(lib.gpd_trsv(h, P(Lp), ld, np_, P(W), np_, nrhs, fwd))
        _sync(lib, h)
        ref = torch.linalg.solve_triangular(L if fwd else L.T, R.T, upper=not fwd).T
        assert (W - ref).abs().max().item() < 1e-9 * max(1.0, ref.abs().max().item()), (np_, fwd)


def test_rowsumsq(lib, h):
    from abstractgps_jl_amd._lib import check

    X = torch.randn(5, 1000, dtype=torch.float64, device="cuda")
    out = torch.zeros(5, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    check(lib.gpd_rowsumsq(h, P(X), 1000, 5, 777, P(out)))
    _sync(lib, h)
    np.testing.assert_allclose(out.cpu().numpy(), (X[:, :777] ** 2).sum(1).cpu().numpy(), rtol=1e-13)


@pytest.mark.parametrize("group", [64, 128, 256, 512])
def test_potrf_leaf_groups(lib, h, group):
    """left-looking leaf groups: a leaf applies the kpre = 0..group/64−1 tiles to its left itself before factoring (64 = every
    leaf is followed by its own GEMM).  Same contract as test_potrf_and_trsm, on shapes that exercise every kpre, ragged row
    counts and more than 256 workgroups."""
    from abstractgps_jl_amd._lib import check

    check(lib.gp_ctx_set_param(h, b"leaf_group", group))
    try:
        for n, extra in ((512, 0), (576, 200 * 128 + 64), (1024, 320), (64, 64)):
            test_potrf_and_trsm(lib, h, n, extra)
        test_potrf_reports_first_bad_pivot(lib, h)
    finally:
        check(lib.gp_ctx_set_param(h, b"leaf_group", 128))


@pytest.mark.parametrize("rt,maxk", [(1, 512), (2, 512), (4, 512), (0, 1024), (0, 0)])
def test_potrf_in_panel_update_tiles(lib, h, rt, maxk):
    """the in-panel updates C[m×N] −= P·P[0:N]ᵀ (K = N = 128 / 256 / 512) through panel_updk_kernel<RT> (csrc/leaf.hpp) with every workgroup
    tile forced in turn (16 / 32 / 64 rows), with K = 1 024 admitted as well, and through the tile GEMM only (maxk = 0, upd128 = 0): same
    contract as test_potrf_and_trsm, up to 27 712 rows below 2 048 columns."""
    from abstractgps_jl_amd._lib import check

    for k, v in ((b"updk_rt", rt), (b"updk_max_k", maxk), (b"updk_tall_k", 1024 if maxk else 256), (b"upd128", 1 if maxk else 0)):
        check(lib.gp_ctx_set_param(h, k, v))
    try:
        for n, extra in ((256, 0), (512, 192), (1024, 320), (2048, 200 * 128 + 64)):
            test_potrf_and_trsm(lib, h, n, extra)
        test_potrf_reports_first_bad_pivot(lib, h)
    finally:
        for k, v in ((b"updk_rt", 0), (b"updk_max_k", 512), (b"updk_tall_k", 256), (b"upd128", 1)):
            check(lib.gp_ctx_set_param(h, k, v))


@pytest.mark.parametrize("nbv", [128, 256, 512, 1024])
def test_trsv_block_sizes(lib, h, nbv):
    """diagonal block of the vector solves (one workgroup) = 128 … 1024; the rest goes to the multi-CU update kernels"""
    from abstractgps_jl_amd._lib import check

    check(lib.gp_ctx_set_param(h, b"trsv_nb", nbv))
    try:
        for np_, nrhs in ((128, 1), (1152, 2), (4096, 1)):
            test_trsv_forward_backward(lib, h, np_, nrhs)
    finally:
        check(lib.gp_ctx_set_param(h, b"trsv_nb", 256))


@pytest.mark.parametrize("nb", [128, 256, 384, 1024, 2048])
def test_block_inverse_and_inverse_block_solve(lib, h, nb):
    """The two pieces of the multi-device panel step through their device-level entry points: gpd_inv_lower — W = −inv(L) of an nb×nb lower block, level by level in
    batched launches (nb = 64·2^m with the second scratch) or by the restricted-row recursion (nb = 384; or no second scratch) — against torch, both forms against each
    other, and gpd_trsm_inv — X ← X L⁻ᵀ as one triangular-k GEMM with that W — against gpd_trsm (substitution) and torch, with a ragged row count."""
    from abstractgps_jl_amd._lib import check

    g = torch.Generator(device="cuda").manual_seed(nb)
    G = torch.randn(nb, 64, dtype=torch.float64, device="cuda", generator=g)
    L = torch.linalg.cholesky(G @ G.T / 64 + 2.0 * torch.eye(nb, dtype=torch.float64, device="cuda"))
    ld = nb + 32
    Lp = torch.zeros(nb + 128, ld, dtype=torch.float64, device="cuda")
    Lp[:nb, :nb] = torch.tril(L) + torch.triu(torch.full_like(L, float("nan")), 1)      # the strictly upper part must never be read
    ref = -torch.linalg.inv(L)
    out = {}
    for tag, two in (("levels", True), ("recursion", False)):
        W = torch.zeros(nb + 128, ld, dtype=torch.float64, device="cuda")
        S1 = torch.zeros(nb + 128, ld, dtype=torch.float64, device="cuda")
        S2 = torch.zeros(nb + 128, ld, dtype=torch.float64, device="cuda")
        torch.cuda.synchronize()
        check(lib.gpd_inv_lower(h, P(Lp), ld, nb, P(W), ld, P(S1), P(S2) if two else None))
        _sync(lib, h)
        assert torch.isfinite(W).all().item()
        assert (W[:nb, :nb] - ref).abs().max().item() < 1e-11 * ref.abs().max().item() * nb, (nb, tag)
        assert W[:nb, :nb].triu(1).abs().max().item() == 0.0
        out[tag] = W
    assert (out["levels"][:nb, :nb] - out["recursion"][:nb, :nb]).abs().max().item() < 1e-11 * ref.abs().max().item() * nb
    m = 64 * 37                                                                           # not a multiple of 128
    X0 = torch.randn(m + 128, ld, dtype=torch.float64, device="cuda", generator=g)
    want = torch.linalg.solve_triangular(L, X0[:m, :nb].T, upper=False).T
    Xa, Xb = X0.clone(), X0.clone()
    S = torch.zeros(m + 128, ld, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    check(lib.gpd_trsm_inv(h, P(Xa), ld, m, P(out["levels"]), ld, nb, P(S), ld))
    check(lib.gpd_trsm(h, P(Xb), ld, m, P(Lp), ld, nb))
    _sync(lib, h)
    scale = want.abs().max().item()
    assert (Xa[:m, :nb] - want).abs().max().item() < 1e-10 * scale
    assert (Xb[:m, :nb] - want).abs().max().item() < 1e-10 * scale
    assert torch.equal(Xa[m:], X0[m:]) and torch.equal(Xa[:m, nb:], X0[:m, nb:])          # nothing outside the m × nb block is touched


@pytest.mark.parametrize("sk", [0, 1])
def test_fit_with_and_without_streamk(agp, sk):
    """the few-tile GEMMs cut along k over all CUs (stream-K, hardware atomics) or not: same logpdf / α as the oracle"""
    from oracle import gp_oracle as o

    x, y = o.synth_inputs(4500, 3, 21)
    fo = o.FiniteGP(o.GP(o.Kernel(o.MATERN52)), x, 0.02)
    ref = float(o.logpdf(fo, y))
    ctx = agp.Context(0)
    ctx.set_param("gemm_streamk", sk)
    ctx.set_param("nb", 1024)
    post = agp.posterior(agp.GP(agp.Matern52Kernel(), ctx=ctx)(agp.RowVecs(x), 0.02), y)
    assert float(post.logpdf_value) == pytest.approx(ref, rel=1e-10)
    alpha = o.posterior(fo, y).alpha
    assert np.linalg.norm(post.data.alpha - alpha) <= 1e-8 * np.linalg.norm(alpha)
    ctx.close()
