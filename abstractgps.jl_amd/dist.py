"""Multi-process 2D block-cyclic driver for the logpdf + posterior pair (one process per GPU).

The N×N matrix K + Σy is partitioned in NB×NB blocks over a P×Q process grid (block (i, j) lives on
rank (i mod P, j mod Q)); only blocks on/below the diagonal are touched.  Every numeric step is a call
into the HIP library through the C ABI's device-level entry points (gpd_* in include/gpmi355.h); this
module owns the schedule and the RCCL traffic (torch.distributed, backend "nccl" = RCCL over xGMI):

  per block column k:   diag owner: Cholesky of its diagonal block + X L⁻ᵀ of its rows below
                        L_kk -> broadcast down the process column; the other owners X L⁻ᵀ their rows
                        panel pieces (one per process row) -> broadcast to all ranks
                        every rank: ONE local trailing update  C -= A Bᵀ  (MFMA gemm with the
                        block-cyclic lower-triangle predicate built into the kernel)
  y − m rides along as an extra block row (forward substitution for free); the backward substitution
  is a distributed block sweep with one 8 KiB reduce + one 8 KiB broadcast per block.

Reference semantics: src/finite_gp_projection.jl:306-311 and src/exact_gpr_posterior.jl:29-35 (the
reference has no distributed path at all — SURVEY.md §5).  The tile backend is injected so the schedule
can be exercised on CPU with gloo in tests/; the product backend is HipTileBackend (no fallback).
"""
from __future__ import annotations

import ctypes as C
import math
import os
from typing import Optional

import numpy as np
import torch
import torch.distributed as dist

LOG2PI = math.log(2.0 * math.pi)
RHS_ROWS = 128


def choose_grid(world: int) -> tuple[int, int]:
    """P×Q with P <= Q, both powers of two where possible: 1→1×1, 2→1×2, 4→2×2, 8→2×4."""
    p = 1
    while (p * 2) * (p * 2) <= world and world % (p * 2) == 0:
        p *= 2
    return p, world // p


class HipTileBackend:
    """Product backend: torch CUDA tensors for memory, libgpmi355 for every operation."""

    def __init__(self, device: int):
        from . import _lib
        from .api import Context

        self._lib = _lib
        self.device = torch.device("cuda", device)
        # a dedicated (non-null) stream shared by torch ops, the RCCL collectives and the library's main stream,
        # so kernels, copies and broadcasts are ordered without host synchronisation
        self.stream = torch.cuda.Stream(self.device)
        self.ctx = Context(device, stream=self.stream.cuda_stream)
        self.lib = self.ctx.lib
        self.h = self.ctx.handle
        self._timing = False
        # ("gemm_pad_lds" = 20480 would pin the GEMM to one workgroup per CU and leave room for the RCCL kernels of the
        # look-ahead; on one GPU that costs 4-7 % of the whole factorisation, so it stays off until measured on a node.)
        if os.environ.get("GPMI_DIST_GEMM_PAD"):
            self.ctx.set_param("gemm_pad_lds", int(os.environ["GPMI_DIST_GEMM_PAD"]))

    @staticmethod
    def _p(t: Optional[torch.Tensor]):
        return None if t is None else C.c_void_p(t.data_ptr())

    def zeros(self, *shape, dtype=torch.float64):
        return torch.zeros(*shape, dtype=dtype, device=self.device)

    def empty(self, *shape, dtype=torch.float64):
        return torch.empty(*shape, dtype=dtype, device=self.device)

    def from_numpy(self, a: np.ndarray):
        return torch.from_numpy(np.ascontiguousarray(a)).to(self.device)

    def assemble(self, kernel_desc, x_dev, n_valid, n_pad, d, noise_dev, grid, a_loc, lda, m_loc, n_loc):
        kind, variance, scale = kernel_desc
        s = None if scale is None else np.ascontiguousarray(scale, dtype=np.float64)
        kk = self._lib.gp_kernel(kind, 0, variance, 0 if s is None else s.shape[0],
                                 None if s is None else s.ctypes.data_as(C.POINTER(C.c_double)))
        # inputs are pre-scaled by the caller: the kernel descriptor's scale is not applied by gpd_assemble
        kk.nscale, kk.scale = 0, None
        g = self._lib.gp_grid(*grid)
        self._lib.check(self.lib.gpd_assemble(self.h, C.byref(kk), self._p(x_dev), n_valid, n_pad, d, self._p(noise_dev),
                                              C.byref(g), self._p(a_loc), lda, m_loc, n_loc))

    def potrf(self, a, lda, m, n, info, col0, n_valid, logdet):
        self._lib.check(self.lib.gpd_potrf(self.h, self._p(a), lda, m, n, self._p(info), col0, n_valid, self._p(logdet)))

    def trsm(self, x, ldx, m, l, ldl, n):
        self._lib.check(self.lib.gpd_trsm(self.h, self._p(x), ldx, m, self._p(l), ldl, n))

    def gemm_nt(self, c, ldc, a, lda, b, ldb, m, n, k, grid, row0, col0):
        g = self._lib.gp_grid(*grid)
        if self._timing:  # only the trailing-update launches are timed (their algorithmic flops are counted by the driver)
            self.ctx.set_param("time_kernels", 1)
        self._lib.check(self.lib.gpd_gemm_nt(self.h, self._p(c), ldc, self._p(a), lda, self._p(b), ldb, m, n, k,
                                             C.byref(g), row0, col0))
        if self._timing:
            self.ctx.set_param("time_kernels", 0)

    def trsv(self, l, ldl, np_, r, ldr, nrhs, forward):
        self._lib.check(self.lib.gpd_trsv(self.h, self._p(l), ldl, np_, self._p(r), ldr, nrhs, 1 if forward else 0))

    def gemv_t(self, l, ldl, nrows, ncols, a, r):
        self._lib.check(self.lib.gpd_gemv_t(self.h, self._p(l), ldl, nrows, ncols, self._p(a), self._p(r)))

    def rowsumsq(self, x, ldx, nrows, ncols, out):
        self._lib.check(self.lib.gpd_rowsumsq(self.h, self._p(x), ldx, nrows, ncols, self._p(out)))

    def sync(self):
        self._lib.check(self.lib.gpd_sync(self.h))
        torch.cuda.synchronize(self.device)

    def time_kernels(self, on: bool):
        self._timing = bool(on)

    def gemm_time(self):
        """(Σ ms, launches) of the MFMA update launches since the last call (needs time_kernels)."""
        ms, cnt = C.c_double(), C.c_int64()
        self._lib.check(self.lib.gpd_gemm_time(self.h, C.byref(ms), C.byref(cnt)))
        return ms.value, cnt.value


def _sub(t: torch.Tensor, r0: int, c0: int) -> torch.Tensor:
    """View of the 2-D row-major tensor t starting at (r0, c0) (same leading dimension)."""
    return t[r0:, c0:]


class BlockCyclicEngine:
    """fit(kernel, x, sigma2, y[, mean]) -> {'logpdf', 'alpha', 'info'} on a P×Q grid of ranks."""

    def __init__(self, device: int = 0, nb: int = 1024, backend=None, grid: Optional[tuple[int, int]] = None):
        if nb % 128:
            raise ValueError("nb must be a multiple of 128")
        self.coll = dist.is_initialized()   # issue the collectives whenever a process group exists (also world 1: API check)
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.P, self.Q = grid or choose_grid(self.world)
        if self.P * self.Q != self.world:
            raise ValueError("grid does not match the world size")
        self.p, self.q = self.rank // self.Q, self.rank % self.Q
        self.nb = nb
        self.be = backend if backend is not None else HipTileBackend(device)
        # process-column groups (ranks sharing q); created in the same order on every rank
        self.col_groups = []
        self.my_col_group = None
        if self.world > 1:
            for qq in range(self.Q):
                ranks = [pp * self.Q + qq for pp in range(self.P)]
                g = dist.new_group(ranks) if self.P > 1 else None
                self.col_groups.append((ranks, g))
            self.my_col_group = self.col_groups[self.q]

    # ---- helpers -------------------------------------------------------------------------------
    def _rank_of(self, p, q):
        return p * self.Q + q

    def _bcast_world(self, t, src):
        if self.coll:
            dist.broadcast(t, src=src)

    def _bcast_col(self, t, src_p, q):
        """broadcast within process column q from (src_p, q); no-op when P == 1."""
        if self.P > 1 and q == self.q:
            ranks, g = self.col_groups[q]
            dist.broadcast(t, src=self._rank_of(src_p, q), group=g)

    def _reduce_col(self, t, dst_p, q):
        """sum over process column q (8 KiB; an all-reduce so it also runs on gloo with device tensors)."""
        if self.P > 1 and q == self.q:
            ranks, g = self.col_groups[q]
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=g)

    def _nlb_before(self, k, p, P):
        """number of global blocks i <= k with i ≡ p (mod P)"""
        return (k - p) // P + 1 if k >= p else 0

    # ---- the pair --------------------------------------------------------------------------------
    def fit(self, kernel, x, sigma2, y, mean=None):
        """kernel: abstractgps api.Kernel; x: (N, D) array (RowVecs) or (N,) vector; sigma2 scalar or (N,)."""
        stream = getattr(self.be, "stream", None)
        if stream is None:
            return self._fit(kernel, x, sigma2, y, mean)
        with torch.cuda.stream(stream):
            return self._fit(kernel, x, sigma2, y, mean)

    def _fit(self, kernel, x, sigma2, y, mean=None):
        be, P, Q, p, q, NB = self.be, self.P, self.Q, self.p, self.q, self.nb
        self.gemm_flops = 0.0
        X = np.asarray(x, dtype=np.float64)
        X = X[:, None] if X.ndim == 1 else X
        n, d = X.shape
        # kernel descriptor + host-side input scaling (k ∘ ScaleTransform / ARDTransform)
        scale = None
        tr = getattr(kernel, "transform", None)
        if tr is not None:
            scale = np.full(d, tr.s) if hasattr(tr, "s") else np.asarray(tr.v, dtype=np.float64)
        Xs = X if scale is None else X * scale
        lcm = P * Q // math.gcd(P, Q)
        nblk = -(-n // NB)
        nblk = -(-nblk // lcm) * lcm          # every rank owns the same number of block rows / columns
        npad = nblk * NB
        nlb_r, nlb_c = nblk // P, nblk // Q   # local block rows / cols
        p_rhs = nblk % P                      # process row that carries the RHS block row (= 0)
        m_loc = nlb_r * NB + (RHS_ROWS if p == p_rhs else 0)
        n_loc = nlb_c * NB
        ldl = n_loc + 32
        tb = NB // 128
        grid = (P, p, Q, q, tb, 1)

        xs_h = np.zeros((d, npad))
        xs_h[:, :n] = Xs.T
        noise_h = np.zeros(npad)
        noise_h[:n] = np.broadcast_to(np.asarray(sigma2, dtype=np.float64), (n,))
        delta = np.zeros(npad)
        delta[:n] = np.asarray(y, dtype=np.float64) - (0.0 if mean is None else np.asarray(mean, dtype=np.float64))

        A = be.zeros(m_loc + 128, ldl)        # +128 slack rows (over-read contract of gemm_nt)
        xs_dev, noise_dev = be.from_numpy(xs_h), be.from_numpy(noise_h)
        info = be.zeros(1, dtype=torch.int32)
        scal = be.zeros(8)
        be.assemble((kernel.kind, kernel.variance, None), xs_dev, n, npad, d, noise_dev, grid, A, ldl, nlb_r * NB, n_loc)
        if p == p_rhs:                         # RHS block row: row 0 = δᵀ restricted to my local columns
            dloc = np.concatenate([delta[(lj * Q + q) * NB:(lj * Q + q + 1) * NB] for lj in range(nlb_c)])
            A[nlb_r * NB, :n_loc] = be.from_numpy(dloc)

        max_rows = nlb_r * NB + RHS_ROWS
        # two sets of panel buffers: panel k+1 is factored and travels while the bulk of update k still runs
        # operand buffers carry 32 padding columns: a power-of-two row stride (NB·8 B) would alias HBM channels
        LDP = NB + 32
        Pbufs = [[be.zeros(max_rows, LDP) for _ in range(P)] for _ in range(2)]
        Bbuf = be.zeros(n_loc + 128, LDP)                     # B operand gathered in my local column order
        Lkk = be.zeros(NB + 128, LDP)                         # (+128 slack rows: operand over-read contract of the GEMM)

        def rows_below(k):
            """per process row pp: (first local row below block k, number of local rows from there incl. RHS rows)"""
            out = []
            for pp in range(P):
                r0p = self._nlb_before(k, pp, P) * NB
                out.append((r0p, nlb_r * NB + (RHS_ROWS if pp == p_rhs else 0) - r0p))
            return out

        def panel(k, Pbuf):
            """factor block column k on its owners, copy my piece into Pbuf[p]; returns the async broadcast handles"""
            pk, qk = k % P, k % Q
            lbk_r, lbk_c = k // P, k // Q                     # local block indices on the owners
            if q == qk:
                c0 = lbk_c * NB
                if P == 1:                                    # one owner: diagonal block and every row below in one call
                    r0 = lbk_r * NB
                    be.potrf(_sub(A, r0, c0), ldl, m_loc - r0, NB, info, k * NB, n, scal[0:1])
                else:
                    # several row owners: factor ONLY the diagonal block, ship L_kk down the process column at once, then
                    # all owners (the diagonal owner included) solve their rows below concurrently
                    if p == pk:
                        r0 = lbk_r * NB
                        be.potrf(_sub(A, r0, c0), ldl, NB, NB, info, k * NB, n, scal[0:1])
                        Lkk[:NB, :NB].copy_(A[r0:r0 + NB, c0:c0 + NB])
                    self._bcast_col(Lkk, pk, qk)
                    r0 = self._nlb_before(k, p, P) * NB       # my first local row with global block > k
                    if m_loc - r0 > 0:
                        be.trsm(_sub(A, r0, c0), ldl, m_loc - r0, Lkk, LDP, NB)
            works = []
            for pp, (r0p, mp) in enumerate(rows_below(k)):    # one piece per process row, to everyone
                if mp <= 0:
                    continue
                piece = Pbuf[pp][:mp]
                if p == pp and q == qk:
                    piece[:, :NB].copy_(A[r0p:r0p + mp, lbk_c * NB:(lbk_c + 1) * NB])
                if self.coll:
                    works.append(dist.broadcast(piece, src=self._rank_of(pp, qk), async_op=True))
            return works

        def update(k, Pbuf, lj_lo, lj_hi):
            """A[rows below k, local block columns lj_lo..lj_hi) -= panel_k(rows) · panel_k(cols)ᵀ (lower part only)"""
            ncols = (lj_hi - lj_lo) * NB
            rows = rows_below(k)
            r0, mrows = rows[p]
            if ncols <= 0 or mrows <= 0:
                return
            for lj in range(lj_lo, lj_hi):                    # B operand: panel rows of global block gj, my column order
                gj = lj * Q + q
                pp = gj % P
                off = (gj // P) * NB - rows[pp][0]
                Bbuf[(lj - lj_lo) * NB:(lj - lj_lo + 1) * NB].copy_(Pbuf[pp][off:off + NB])
            be.gemm_nt(_sub(A, r0, lj_lo * NB), ldl, Pbuf[p], LDP, Bbuf, LDP, mrows, ncols, NB, grid, r0, lj_lo * NB)
            # algorithmic flops of this launch: local elements on/below the global diagonal × 2·NB
            cnt = 0
            for lj in range(lj_lo, lj_hi):
                gj = lj * Q + q
                for li_ in range(r0 // NB, nlb_r):
                    gi = li_ * P + p
                    cnt += NB * NB if gi > gj else (NB * (NB + 1) // 2 if gi == gj else 0)
                if p == p_rhs:
                    cnt += RHS_ROWS * NB
            self.gemm_flops += 2.0 * NB * cnt

        for w_ in panel(0, Pbufs[0]):
            w_.wait()
        for k in range(nblk):
            cur, nxt = Pbufs[k % 2], Pbufs[(k + 1) % 2]
            lj0 = self._nlb_before(k, q, Q)                   # my first local block column with global index > k
            works = []
            lj_rest = lj0
            if k + 1 < nblk:
                if q == (k + 1) % Q:                          # look-ahead: bring block column k+1 up to date first ...
                    update(k, cur, lj0, lj0 + 1)
                    lj_rest = lj0 + 1
                works = panel(k + 1, nxt)                     # ... factor it and put it on the wire (asynchronous)
            update(k, cur, lj_rest, nlb_c)                    # the bulk of the trailing update overlaps the transfers
            for w_ in works:
                w_.wait()

        # ---- scalars: logdet (diag owners), ‖z‖² (RHS row pieces), info
        if p == p_rhs:
            be.rowsumsq(_sub(A, nlb_r * NB, 0), ldl, 1, n_loc, scal[1:2])
        red = torch.stack([scal[0], scal[1]])
        # LAPACK info = the FIRST failing leading minor (PosDefException(info) in the reference): ranks that saw no failure
        # contribute a sentinel, then MIN over ranks (after the first bad pivot NaNs make later diagonal owners flag too)
        info_f = info.to(torch.float64)
        info_f = torch.where(info_f > 0, info_f, torch.full_like(info_f, 2.0**52))
        if self.coll:
            dist.all_reduce(red, op=dist.ReduceOp.SUM)
            dist.all_reduce(info_f, op=dist.ReduceOp.MIN)
        logdet_half, sq = float(red[0].item()), float(red[1].item())
        info_v = int(info_f.item()) if float(info_f.item()) < 2.0**52 else 0
        if info_v != 0:
            be.sync()
            from ._lib import PosDefException

            raise PosDefException(info_v)
        logpdf = -0.5 * (n * LOG2PI + 2.0 * logdet_half + sq)

        # ---- backward substitution  α = L⁻ᵀ z  (block sweep, last block first)
        alpha = be.zeros(npad)
        acc = be.zeros(n_loc)                                  # my partial −Σ_k L[k][j]ᵀ α_k per local column
        rk = be.zeros(NB)
        for k in range(nblk - 1, -1, -1):
            pk, qk = k % P, k % Q
            lbk_r, lbk_c = k // P, k // Q
            if q == qk:
                c0 = lbk_c * NB
                rk.copy_(acc[c0:c0 + NB])
                if p == p_rhs:
                    rk.add_(A[nlb_r * NB, c0:c0 + NB])        # z_k from the RHS row
                self._reduce_col(rk, pk, qk)
                if p == pk:
                    r0 = lbk_r * NB
                    be.trsv(_sub(A, r0, c0), ldl, NB, rk, NB, 1, False)
            self._bcast_world(rk, self._rank_of(pk, qk))
            alpha[k * NB:(k + 1) * NB].copy_(rk)
            if p == pk:                                        # my block row k: acc_j −= L[k][j]ᵀ α_k for local j < k
                ncb = self._nlb_before(k - 1, q, Q) if k > 0 else 0
                if ncb > 0:
                    be.gemv_t(_sub(A, lbk_r * NB, 0), ldl, NB, ncb * NB, rk, acc)
        be.sync()
        out = {"logpdf": logpdf, "info": info_v, "alpha": alpha[:n].cpu().numpy(), "grid": (P, Q), "nb": NB,
               "gemm_flops": self.gemm_flops}
        return out
