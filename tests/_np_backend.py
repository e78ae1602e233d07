"""TEST INFRASTRUCTURE: a NumPy stand-in for HipTileBackend so the block-cyclic schedule and the
collective pattern of abstractgps.jl_amd/dist.py can run on CPU with gloo.  Same method contract as the
product backend (row-major views + leading dimensions); never used by the product path."""
import numpy as np
import scipy.linalg as sla
import torch

from oracle import gp_oracle as o


def _glob(loc, nb, P, p):
    return ((loc // nb) * P + p) * nb + (loc % nb)


class NumpyTileBackend:
    def __init__(self):
        self.calls = {"potrf": 0, "trsm": 0, "gemm": 0}

    def zeros(self, *shape, dtype=torch.float64):
        return torch.zeros(*shape, dtype=dtype)

    def empty(self, *shape, dtype=torch.float64):
        return torch.full(shape, float("nan"), dtype=dtype)  # poison: unwritten data must never be consumed

    def from_numpy(self, a):
        return torch.from_numpy(np.ascontiguousarray(a))

    def sync(self):
        pass

    def time_kernels(self, on):
        pass

    def gemm_time(self):
        return 0.0, 0

    def assemble(self, kernel_desc, x_dev, n_valid, n_pad, d, noise_dev, grid, a_loc, lda, m_loc, n_loc):
        kind, variance, _ = kernel_desc
        P, p, Q, q, tb, lower = grid
        nb = tb * 128
        X = x_dev.numpy().reshape(d, n_pad).T  # pre-scaled inputs
        gi = _glob(np.arange(m_loc), nb, P, p)
        gj = _glob(np.arange(n_loc), nb, Q, q)
        K = o.kernelmatrix(o.Kernel(kind, variance, None), X[gi], X[gj])
        K = np.where((gi[:, None] >= n_valid) | (gj[None, :] >= n_valid), (gi[:, None] == gj[None, :]) * 1.0, K)
        diag = gi[:, None] == gj[None, :]
        noise = noise_dev.numpy()
        ii, jj = np.nonzero(diag)
        for a, b in zip(ii, jj):
            if gi[a] < n_valid:
                K[a, b] += noise[gi[a]]
        if lower:  # leave 128-tiles strictly above the global diagonal untouched (as the HIP kernel does)
            view = a_loc.numpy()
            for bi in range(m_loc // 128):
                for bj in range(n_loc // 128):
                    if gj[bj * 128] > gi[bi * 128] + 127:
                        continue
                    view[bi * 128:(bi + 1) * 128, bj * 128:(bj + 1) * 128] = K[bi * 128:(bi + 1) * 128, bj * 128:(bj + 1) * 128]
        else:
            a_loc.numpy()[:m_loc, :n_loc] = K

    def potrf(self, a, lda, m, n, info, col0, n_valid, logdet):
        self.calls["potrf"] += 1
        A = a.numpy()
        S = np.tril(A[:n, :n]) + np.tril(A[:n, :n], -1).T
        try:
            L = np.linalg.cholesky(S)
        except np.linalg.LinAlgError:
            info[0] = col0 + 1
            return
        A[:n, :n] = np.tril(L) + np.triu(A[:n, :n], 1)
        if m > n:
            A[n:m, :n] = sla.solve_triangular(L, A[n:m, :n].T, lower=True).T
        if logdet is not None:
            idx = col0 + np.arange(n) < n_valid
            logdet[0] += float(np.sum(np.log(np.diag(L))[idx]))

    def trsm(self, x, ldx, m, l, ldl, n):
        self.calls["trsm"] += 1
        X, L = x.numpy(), np.tril(l.numpy()[:n, :n])
        X[:m, :n] = sla.solve_triangular(L, X[:m, :n].T, lower=True).T

    def gemm_nt(self, c, ldc, a, lda, b, ldb, m, n, k, grid, row0, col0):
        self.calls["gemm"] += 1
        P, p, Q, q, tb, lower = grid
        nb = tb * 128
        C, A, B = c.numpy(), a.numpy(), b.numpy()
        upd = A[:m, :k] @ B[:n, :k].T
        if lower:
            gi = _glob(row0 + np.arange(m), nb, P, p)
            gj = _glob(col0 + np.arange(n), nb, Q, q)
            mask = (gj[None, :] // 64) <= (gi[:, None] // 64)  # 64×64 sub-tiles on/below the diagonal
            C[:m, :n] = np.where(mask, C[:m, :n] - np.where(mask, upd, 0.0), C[:m, :n])
        else:
            C[:m, :n] -= upd

    def trsv(self, l, ldl, np_, r, ldr, nrhs, forward):
        L = np.tril(l.numpy()[:np_, :np_])
        R = r.numpy()
        R[:np_] = sla.solve_triangular(L, R[:np_], lower=True, trans="N" if forward else "T")

    def gemv_t(self, l, ldl, nrows, ncols, a, r):
        r.numpy()[:ncols] -= l.numpy()[:nrows, :ncols].T @ a.numpy()[:nrows]

    def rowsumsq(self, x, ldx, nrows, ncols, out):
        out.numpy()[:nrows] = (x.numpy()[:nrows, :ncols] ** 2).sum(axis=1)
