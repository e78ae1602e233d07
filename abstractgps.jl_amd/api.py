"""Host-side mirror of the AbstractGPs.jl API surface for the accelerated path.

Line-for-line Python counterpart of the Julia shim (julia/HipGPs.jl): same names, argument meaning
and error behaviour as the reference (file:line cited per function, relative to the upstream repo),
every numeric step a call through the C ABI (include/gpmi355.h) into the HIP library.  Nothing here
computes Gram matrices, factorisations or solves on the host (held-out logpdf and posterior sampling included:
gp_posterior_logpdf / gp_posterior_rand / gp_vfe_logpdf / gp_vfe_rand).

    f   = GP(SqExponentialKernel())                    # src/base_gp.jl:57-64
    fx  = f(x, 0.01)                                   # src/finite_gp_projection.jl:32
    lp  = logpdf(fx, y)                                # src/finite_gp_projection.jl:306
    fp  = posterior(fx, y)                             # src/exact_gpr_posterior.jl:29
    m, v = mean_and_var(fp(xs))                        # src/exact_gpr_posterior.jl:85
"""
from __future__ import annotations

import ctypes as C
from collections.abc import Mapping
import math
import threading
import weakref
from dataclasses import dataclass, field
from typing import Callable, Optional, Union

import numpy as np

from . import _lib
from ._lib import PosDefException, check, gp_kernel, gp_noise, gp_points, gp_timings

default_sigma2 = 1e-18  # src/finite_gp_projection.jl:17


# --------------------------------------------------------------------------------------------
# Kernels and transforms (KernelFunctions.jl surface used by the path)
# --------------------------------------------------------------------------------------------
@dataclass(frozen=True)
class ScaleTransform:
    s: float


@dataclass(frozen=True)
class ARDTransform:
    v: tuple

    def __init__(self, v):
        object.__setattr__(self, "v", tuple(float(t) for t in np.asarray(v).ravel()))


@dataclass(frozen=True)
class Kernel:
    kind: int
    variance: float = 1.0
    transform: Union[None, ScaleTransform, ARDTransform] = None

    def __matmul__(self, t):  # k ∘ ScaleTransform(s)  /  k ∘ ARDTransform(v)
        if not isinstance(t, (ScaleTransform, ARDTransform)):
            raise TypeError("only ScaleTransform / ARDTransform are accelerated")
        if self.transform is not None:
            raise TypeError("nested transforms are not accelerated")
        return Kernel(self.kind, self.variance, t)

    def __rmul__(self, a):  # α * k  (ScaledKernel)
        return Kernel(self.kind, self.variance * float(a), self.transform)

    __mul__ = __rmul__


def SqExponentialKernel():
    return Kernel(0)


SEKernel = RBFKernel = GaussianKernel = SqExponentialKernel


def Matern12Kernel():
    return Kernel(1)


ExponentialKernel = Matern12Kernel


def Matern32Kernel():
    return Kernel(2)


def Matern52Kernel():
    return Kernel(3)


def compose(k: Kernel, t) -> Kernel:
    return k @ t


def with_lengthscale(k: Kernel, l) -> Kernel:
    """with_lengthscale(k, ℓ) ≡ k ∘ ScaleTransform(1/ℓ) (scalar) or k ∘ ARDTransform(1 ./ ℓ)."""
    l = np.asarray(l, dtype=np.float64)
    return k @ (ScaleTransform(1.0 / float(l)) if l.ndim == 0 else ARDTransform(1.0 / l))


# --------------------------------------------------------------------------------------------
# Input containers (KernelFunctions ColVecs / RowVecs; src/finite_gp_projection.jl:32-37)
# --------------------------------------------------------------------------------------------
@dataclass
class ColVecs:
    X: np.ndarray  # D × N, points are columns

    def __len__(self):
        return self.X.shape[1]


@dataclass
class RowVecs:
    X: np.ndarray  # N × D, points are rows

    def __len__(self):
        return self.X.shape[0]


def _as_input(x):
    if isinstance(x, (ColVecs, RowVecs)):
        return x
    x = np.asarray(x)
    if x.ndim == 1:
        return x
    if x.ndim == 2:
        return RowVecs(x)  # a bare matrix is taken as N × D
    raise TypeError("inputs must be a vector, RowVecs or ColVecs")


def _npoints(x) -> int:
    return len(x)


class _Marshal:
    """Keeps the numpy buffers behind the ctypes structs alive for the duration of a call."""

    def __init__(self, dtype):
        self.dtype = np.dtype(dtype)
        self.keep = []

    def arr(self, a, order="C"):
        a = np.ascontiguousarray(a, dtype=self.dtype) if order == "C" else np.asfortranarray(a, dtype=self.dtype)
        self.keep.append(a)
        return a

    def ptr(self, a):
        return None if a is None else C.c_void_p(a.ctypes.data)

    def points(self, x) -> gp_points:
        """gp_points of an input container WITHOUT a host copy when the array already lies in one of the two ABI layouts (round 6: the transposing copy of
        RowVecs(X) with a C-ordered X cost 0.6–1.9 ms per call at N = 262 144 — inside every timed sparse fit — and has no counterpart in the Julia shim, where
        RowVecs(X) is column-major N×D = layout 2 as it lies).  layout 1 = element (dimension dd, point i) at data[dd + i·D] — a C-ordered (N, D) array, or a
        Fortran-ordered (D, N) one; layout 2 = data[i + dd·N] — a C-ordered (D, N) array, or a Fortran-ordered (N, D) one."""
        x = _as_input(x)
        if isinstance(x, (ColVecs, RowVecs)):
            A = np.asarray(x.X)
            n, d = (A.shape[1], A.shape[0]) if isinstance(x, ColVecs) else (A.shape[0], A.shape[1])
            point_major = A.flags.c_contiguous if isinstance(x, RowVecs) else A.flags.f_contiguous      # memory order [point][dimension]
            dim_major = A.flags.f_contiguous if isinstance(x, RowVecs) else A.flags.c_contiguous        # memory order [dimension][point]
            if A.dtype == self.dtype and (point_major or dim_major) and A.size:
                self.keep.append(A)
                return gp_points(A.ctypes.data, n, d, 1 if point_major else 2)
            if isinstance(x, ColVecs):  # D×N column-major  == (N, D) C-contiguous
                X = self.arr(A.T)
                return gp_points(X.ctypes.data, X.shape[0], X.shape[1], 1)
            X = self.arr(A.T)           # N×D column-major == (D, N) C-contiguous
            return gp_points(X.ctypes.data, X.shape[1], X.shape[0], 2)
        v = self.arr(x)
        return gp_points(v.ctypes.data, v.shape[0], 1, 0)

    def kernel(self, k: Kernel, d: int) -> gp_kernel:
        dt = 0 if self.dtype == np.float64 else 1
        if k.transform is None:
            return gp_kernel(k.kind, dt, k.variance, 0, None)
        if isinstance(k.transform, ScaleTransform):
            s = np.array([k.transform.s], dtype=np.float64)
        else:
            s = np.array(k.transform.v, dtype=np.float64)
            if s.shape[0] != d:
                raise ValueError(f"DimensionMismatch: ARDTransform has {s.shape[0]} scales, inputs have D={d}")
        self.keep.append(s)
        return gp_kernel(k.kind, dt, k.variance, s.shape[0], s.ctypes.data_as(C.POINTER(C.c_double)))

    def noise(self, sigma2, n: int) -> gp_noise:
        s = np.asarray(sigma2)
        if s.ndim == 0:
            return gp_noise(0, float(s), None)
        if s.ndim == 1:
            if s.shape[0] != n:
                raise ValueError("DimensionMismatch: noise vector length != number of points")
            v = self.arr(s)
            return gp_noise(1, 0.0, v.ctypes.data)
        raise NotImplementedError("dense Σy is outside the accelerated path (falls back to stock AbstractGPs in the Julia shim)")


def _input_dim(x) -> int:
    x = _as_input(x)
    if isinstance(x, ColVecs):
        return x.X.shape[0]
    if isinstance(x, RowVecs):
        return x.X.shape[1]
    return 1


def _input_dtype(x):
    x = _as_input(x)
    a = x.X if isinstance(x, (ColVecs, RowVecs)) else x
    return np.float32 if np.asarray(a).dtype == np.float32 else np.float64


# --------------------------------------------------------------------------------------------
# Engine context (one per device)
# --------------------------------------------------------------------------------------------
class Context:
    """gp_ctx wrapper.  `stream` = a raw hipStream_t (int) to issue main-stream work on, or None.
    `devices=[d0, d1, ...]` builds a multi-device context (gp_ctx_create_multi): fp64 logpdf / posterior fits are then
    partitioned 2D block-cyclically over those devices inside the library (grid P×Q, 0 = chosen for the fabric; block nb);
    listing one device several times gives virtual ranks (the whole schedule on one GPU)."""

    def __init__(self, device: int = 0, stream: Optional[int] = None, devices=None, P: int = 0, Q: int = 0, nb: int = 0):
        self.lib = _lib.load()
        h = C.c_void_p()
        if devices is not None:
            devs = (C.c_int32 * len(devices))(*[int(v) for v in devices])
            check(self.lib.gp_ctx_create_multi(C.byref(h), devs, len(devices), int(P), int(Q), int(nb)))
            device = int(devices[0])
        else:
            check(self.lib.gp_ctx_create(C.byref(h), int(device), C.c_void_p(stream) if stream else None))
        self.handle = h
        self.device = device
        self._fin = weakref.finalize(self, self.lib.gp_ctx_destroy, h)

    def multi_info(self) -> dict:
        v = [C.c_int32() for _ in range(5)]
        check(self.lib.gp_ctx_multi_info(self.handle, *[C.byref(t) for t in v]))
        P, Q, nb, comm, depth = (t.value for t in v)
        return {"P": P, "Q": Q, "nb": nb, "comm": {0: "none", 1: "rccl", 2: "copies"}.get(comm, str(comm)), "lookahead_depth": depth}

    def multi_stats(self) -> dict:
        """fit attempts of the multi-device driver and how many were repetitions after a failed self-check (gp_ctx_multi_stats)"""
        f, r, sv = C.c_int64(), C.c_int64(), C.c_int64()
        check(self.lib.gp_ctx_multi_stats(self.handle, C.byref(f), C.byref(r), C.byref(sv)))
        return {"fits": f.value, "retries": r.value, "solves": sv.value}

    def set_param(self, name: str, value: int) -> None:
        check(self.lib.gp_ctx_set_param(self.handle, name.encode(), int(value)))

    def get_param(self, name: str) -> int:
        v = C.c_int64()
        check(self.lib.gp_ctx_get_param(self.handle, name.encode(), C.byref(v)))
        return int(v.value)

    def timings(self) -> dict:
        t = gp_timings()
        check(self.lib.gp_get_timings(self.handle, C.byref(t)))
        return {f: getattr(t, f) for f, _ in gp_timings._fields_}

    def trim(self) -> None:
        """Hand the ctx's cached free device blocks back to the HIP allocator (e.g. after freeing a 34 GB posterior)."""
        check(self.lib.gp_ctx_trim(self.handle))

    def close(self):
        self._fin()


_default_ctx = {}
_ctx_lock = threading.Lock()


def default_context(device: int = 0) -> Context:
    with _ctx_lock:
        if device not in _default_ctx:
            _default_ctx[device] = Context(device)
        return _default_ctx[device]


# --------------------------------------------------------------------------------------------
# GP types
# --------------------------------------------------------------------------------------------
Mean = Union[None, float, Callable]


def _mean_vector(mean: Mean, x, dtype) -> Optional[np.ndarray]:
    """mean_vector — src/mean_function.jl:27,40,52-55.  None = ZeroMean (stays lazy)."""
    if mean is None:
        return None
    n = _npoints(x)
    if isinstance(mean, (int, float, np.floating)):
        return np.full(n, mean, dtype=dtype)
    x = _as_input(x)
    if isinstance(x, ColVecs):
        return np.asarray([mean(c) for c in x.X.T], dtype=dtype)
    if isinstance(x, RowVecs):
        return np.asarray([mean(r) for r in x.X], dtype=dtype)
    return np.asarray([mean(v) for v in x], dtype=dtype)


class AbstractGP:
    def __call__(self, x, sigma2=default_sigma2) -> "FiniteGP":  # src/finite_gp_projection.jl:32
        return FiniteGP(self, _as_input(x), sigma2)

    def mean(self, x=None):  # src/abstract_gp.jl:66-87
        if x is None:
            raise TypeError("`mean(f)` for an AbstractGP requires inputs: use `mean(f(x))` or `mean(f, x)`")
        raise NotImplementedError


@dataclass(eq=False)
class GP(AbstractGP):
    """HipGP: GP(mean, kernel) whose FiniteGP methods run on the MI355X (src/base_gp.jl:57-64)."""

    kernel: Kernel
    mean_fn: Mean = None
    ctx: Optional[Context] = None

    def __init__(self, *args, ctx: Optional[Context] = None):
        if len(args) == 1:
            self.mean_fn, self.kernel = None, args[0]  # GP(kernel) -> ZeroMean  base_gp.jl:64
        elif len(args) == 2:
            self.mean_fn, self.kernel = args  # GP(c::Real, k) / GP(f, k)  base_gp.jl:62-63
        else:
            raise TypeError("GP(kernel) or GP(mean, kernel)")
        if not isinstance(self.kernel, Kernel):
            raise TypeError("kernel must be one of the accelerated kernels")
        self.ctx = ctx

    def context(self) -> Context:
        return self.ctx or default_context()

    # internal AbstractGP API (src/base_gp.jl:68-74)
    def mean(self, x=None):
        if x is None:
            return super().mean()
        m = _mean_vector(self.mean_fn, x, _input_dtype(x))
        return np.zeros(_npoints(x), dtype=_input_dtype(x)) if m is None else m

    def cov(self, x, z=None):
        return kernelmatrix(self.kernel, x, z, ctx=self.context())

    def var(self, x):
        return np.full(_npoints(x), self.kernel.variance, dtype=_input_dtype(x))  # kernelmatrix_diag

    def mean_and_var(self, x):
        return self.mean(x), self.var(x)


@dataclass(eq=False)
class FiniteGP:
    """src/finite_gp_projection.jl:7-21."""

    f: AbstractGP
    x: object
    sigma2: object = default_sigma2

    def __len__(self):
        return _npoints(self.x)

    def noise_vector(self):
        s = np.asarray(self.sigma2)
        return np.full(len(self), float(s)) if s.ndim == 0 else s

    def _prior_factor(self):
        """Device-resident factor of cov(fx) = K + Σy for a GP prior, computed once per FiniteGP object and reused by
        rand (the reference refactors on every call, src/finite_gp_projection.jl:235)."""
        fac = getattr(self, "_fac", None)
        if fac is None:
            fac = _posterior_exact(self, np.zeros(len(self), dtype=_input_dtype(self.x)), zero_mean=True).data.C
            object.__setattr__(self, "_fac", fac)
        return fac


def kernelmatrix(k: Kernel, x, z=None, ctx: Optional[Context] = None) -> np.ndarray:
    """KernelFunctions.kernelmatrix(k, x[, z]) on the device (src/base_gp.jl:70,74)."""
    ctx = ctx or default_context()
    dt = _input_dtype(x)
    m = _Marshal(dt)
    px = m.points(x)
    kk = m.kernel(k, px.d)
    n = px.n
    if z is None:
        out = np.empty((n, n), dtype=dt, order="F")
        check(ctx.lib.gp_kernelmatrix(ctx.handle, C.byref(kk), C.byref(px), None, out.ctypes.data))
    else:
        pz = m.points(z)
        out = np.empty((n, pz.n), dtype=dt, order="F")
        check(ctx.lib.gp_kernelmatrix(ctx.handle, C.byref(kk), C.byref(px), C.byref(pz), out.ctypes.data))
    return out


# --------------------------------------------------------------------------------------------
# FiniteGP API on GP priors: logpdf / posterior
# --------------------------------------------------------------------------------------------
def _check_y(fx: FiniteGP, y):
    y = np.asarray(y)
    if y.shape[0] != len(fx):
        raise ValueError(f"DimensionMismatch: length(fx) = {len(fx)} but y has {y.shape[0]} rows")
    return y


def logpdf(fx: FiniteGP, y):
    """logpdf(f::FiniteGP, Y) — src/finite_gp_projection.jl:306-311.  Vector -> scalar of the input
    eltype; matrix (N×S) -> length-S vector."""
    y = _check_y(fx, y)
    f = fx.f
    if isinstance(f, (PosteriorGP, ApproxPosteriorGP)):
        return _logpdf_posterior(fx, y)
    if not isinstance(f, GP):
        raise TypeError("logpdf: unsupported GP type")
    ctx = f.context()
    dt = np.result_type(_input_dtype(fx.x), np.float32 if y.dtype == np.float32 else np.float64).type
    m = _Marshal(dt)
    px = m.points(fx.x)
    kk = m.kernel(f.kernel, px.d)
    nz = m.noise(fx.sigma2, px.n)
    mean = _mean_vector(f.mean_fn, fx.x, dt)
    mean = None if mean is None else m.arr(mean)
    Y = m.arr(y if y.ndim == 2 else y[:, None], order="F")
    out = np.empty(Y.shape[1], dtype=dt)
    check(ctx.lib.gp_logpdf(ctx.handle, C.byref(kk), C.byref(px), C.byref(nz), m.ptr(mean), Y.ctypes.data,
                            Y.shape[0], Y.shape[1], out.ctypes.data))
    return out[0] if y.ndim == 1 else out


def _terms(fx: FiniteGP, y, want_logdet: bool, want_sqmahal: bool):
    """gp_logpdf_terms: logdet(cov(fx)) and / or sqmahal(fx, y) from ONE factorisation on the device."""
    f = fx.f
    if not isinstance(f, GP):
        raise TypeError("only GP priors are accelerated here (the shim falls back to the stock methods otherwise)")
    ctx = f.context()
    dt = _input_dtype(fx.x) if y is None else np.result_type(_input_dtype(fx.x), np.float32 if np.asarray(y).dtype == np.float32 else np.float64).type
    m = _Marshal(dt)
    px = m.points(fx.x)
    kk = m.kernel(f.kernel, px.d)
    nz = m.noise(fx.sigma2, px.n)
    mean = _mean_vector(f.mean_fn, fx.x, dt)
    mean = None if mean is None else m.arr(mean)
    Y = None if y is None else m.arr(y if y.ndim == 2 else y[:, None], order="F")
    ld = np.empty(1, dtype=dt)
    sq = np.empty(1 if Y is None else Y.shape[1], dtype=dt)
    check(ctx.lib.gp_logpdf_terms(ctx.handle, C.byref(kk), C.byref(px), C.byref(nz), m.ptr(mean), m.ptr(Y), px.n,
                                  0 if Y is None else Y.shape[1], ld.ctypes.data if want_logdet else None,
                                  sq.ctypes.data if want_sqmahal else None))
    return ld[0], sq


def sqmahal(fx: FiniteGP, y):
    """Distributions.sqmahal(f::FiniteGP, x) — src/finite_gp_projection.jl:315-326: (y − m)ᵀ C⁻¹ (y − m); one value per column."""
    y = _check_y(fx, y)
    sq = _terms(fx, y, False, True)[1]
    return sq[0] if y.ndim == 1 else sq


def logdetcov(fx: FiniteGP):
    """Distributions.logdetcov(f::FiniteGP) = logdet(cov(f)) — src/finite_gp_projection.jl:313."""
    return _terms(fx, None, True, False)[0]


def gradlogpdf(fx: FiniteGP, y):
    """Distributions.gradlogpdf(f::FiniteGP, x) = C \\ (m .- x) — src/finite_gp_projection.jl:328-337: −α for a vector; for a matrix
    the factor of the first column's fit is reused for all columns (gp_posterior_solve)."""
    y = _check_y(fx, y)
    f = fx.f
    if not isinstance(f, GP):
        raise TypeError("only GP priors are accelerated here")
    post = _posterior_exact(fx, y if y.ndim == 1 else y[:, 0])
    try:
        if y.ndim == 1:
            return -post.data.alpha
        dt = post.data.alpha.dtype
        mean = _mean_vector(f.mean_fn, fx.x, dt)
        D = np.asfortranarray((np.asarray(y, dtype=dt) - (0 if mean is None else np.asarray(mean, dtype=dt)[:, None])))
        out = np.empty_like(D, order="F")
        check(post.data.C.ctx.lib.gp_posterior_solve(post.data.C.handle, D.ctypes.data, D.shape[1], out.ctypes.data))
        return -out
    finally:
        post.data.C.free()


def logpdf_and_grad(fx: FiniteGP, y, wrt_x: bool = False) -> tuple:
    """Value and gradient of logpdf(fx, y) for the rrule of the accelerated path (the reference differentiates the same
    expression by AD — test/finite_gp_projection.jl:152-178).  Returns (logpdf, grads) with grads =
    {"variance": ∂/∂σ_k², "scale": ∂/∂s (ScaleTransform) or ∂/∂v (ARDTransform) or None, "noise": ∂/∂σ² (scalar Σy) or the
    vector ∂/∂Σy_ii, "y": −α, "mean": +α} and, with wrt_x, "x": ∂/∂x in the shape of the input container's array."""
    y = _check_y(fx, y)
    if y.ndim != 1:
        raise TypeError("logpdf_and_grad expects a vector of observations")
    f = fx.f
    if not isinstance(f, GP):
        raise TypeError("logpdf_and_grad: unsupported GP type")
    ctx = f.context()
    dt = np.result_type(_input_dtype(fx.x), np.float32 if y.dtype == np.float32 else np.float64).type
    m = _Marshal(dt)
    px = m.points(fx.x)
    kk = m.kernel(f.kernel, px.d)
    nz = m.noise(fx.sigma2, px.n)
    mean = _mean_vector(f.mean_fn, fx.x, dt)
    mean = None if mean is None else m.arr(mean)
    yv = m.arr(y)
    lp = np.empty(1, dtype=dt)
    dvar = C.c_double()
    dscale = (C.c_double * max(kk.nscale, 1))()
    dnoise = np.empty(1 if nz.kind == 0 else px.n, dtype=dt)
    dy = np.empty(px.n, dtype=dt)
    # ∂/∂x comes back in the ABI layout of the inputs: (n,) vector; ColVecs (N, D) C-order = D×N column-major; RowVecs (D, N)
    dxb = None
    if wrt_x:
        dxb = np.empty((px.n,) if px.layout == 0 else ((px.n, px.d) if px.layout == 1 else (px.d, px.n)), dtype=dt)
    check(ctx.lib.gp_logpdf_grad(ctx.handle, C.byref(kk), C.byref(px), C.byref(nz), m.ptr(mean), yv.ctypes.data,
                                 lp.ctypes.data, C.byref(dvar), dscale, dnoise.ctypes.data, dy.ctypes.data, m.ptr(dxb)))
    sc = None
    if kk.nscale == 1:
        sc = float(dscale[0])
    elif kk.nscale > 1:
        sc = np.array([dscale[i] for i in range(kk.nscale)])
    g = {"variance": dvar.value, "scale": sc, "noise": dnoise[0] if nz.kind == 0 else dnoise, "y": dy, "mean": -dy}
    if wrt_x:  # the buffer comes back in the ABI layout the inputs were passed in: (n, d) for layout 1, (d, n) for layout 2; the result has the shape of x.X
        if px.layout == 0:
            g["x"] = dxb
        else:
            nd = dxb if px.layout == 1 else dxb.T                       # (N, D) view
            g["x"] = np.ascontiguousarray(nd.T if isinstance(_as_input(fx.x), ColVecs) else nd)
    return lp[0], g


def loglikelihood(fx: FiniteGP, Y):  # src/finite_gp_projection.jl:304
    return np.sum(logpdf(fx, Y))


class _Factor:
    """Device-resident Cholesky factor handle (PosteriorGP.data.C)."""

    def __init__(self, ctx: Context, handle: C.c_void_p, n: int, dtype):
        self.ctx, self.handle, self.n, self.dtype = ctx, handle, n, dtype
        self._fin = weakref.finalize(self, ctx.lib.gp_posterior_free, handle)

    @property
    def U(self) -> np.ndarray:
        """C.U on the host (parity/debug; N×N copy)."""
        out = np.empty((self.n, self.n), dtype=self.dtype, order="F")
        check(self.ctx.lib.gp_posterior_get_factor(self.handle, out.ctypes.data))
        return out

    def solve(self, B) -> np.ndarray:
        """`C \\ B` by forward + backward sweeps over the resident factor (gp_posterior_solve) — on the block-cyclic pieces when the
        factor comes from a multi-device fit."""
        B = np.asarray(B, dtype=self.dtype)
        D = np.asfortranarray(B.reshape(self.n, -1))
        out = np.empty_like(D, order="F")
        check(self.ctx.lib.gp_posterior_solve(self.handle, D.ctypes.data, D.shape[1], out.ctypes.data))
        return np.ascontiguousarray(out).reshape(B.shape)

    def Ut_mul(self, xi) -> np.ndarray:
        """`C.U' * xi` (gp_posterior_factor_mul: the sampling transform of rand) — on the pieces for a multi-device factor."""
        xi = np.asarray(xi, dtype=self.dtype)
        D = np.asfortranarray(xi.reshape(self.n, -1))
        out = np.empty_like(D, order="F")
        check(self.ctx.lib.gp_posterior_factor_mul(self.handle, D.ctypes.data, D.shape[1], out.ctypes.data))
        return np.ascontiguousarray(out).reshape(xi.shape)

    def free(self):
        self._fin()


@dataclass
class _PostData:
    alpha: np.ndarray  # α
    C: _Factor
    x: object
    delta: np.ndarray  # δ


class PosteriorGP(AbstractGP):
    """src/exact_gpr_posterior.jl:1-4, with data = (α, C, x, δ) (:34); C lives on the device."""

    def __init__(self, prior: GP, data: _PostData, logpdf_value):
        self.prior, self.data, self.logpdf_value = prior, data, logpdf_value

    def _predict(self, x, what):
        ctx = self.data.C.ctx
        dt = self.data.C.dtype
        m = _Marshal(dt)
        px = m.points(x)
        ns = px.n
        pm = _mean_vector(self.prior.mean_fn, x, dt)
        pm = None if pm is None else m.arr(pm)
        mean = np.empty(ns, dtype=dt) if what & 1 else None
        var = np.empty(ns, dtype=dt) if what & 2 else None
        cov = np.empty((ns, ns), dtype=dt, order="F") if what & 4 else None
        check(ctx.lib.gp_posterior_predict(self.data.C.handle, C.byref(px), m.ptr(pm), what, m.ptr(mean), m.ptr(var),
                                           m.ptr(cov)))
        return mean, var, cov

    def mean(self, x=None):  # :60-62
        if x is None:
            return super().mean()
        return self._predict(x, 1)[0]

    def var(self, x):  # :68-70
        return self._predict(x, 2)[1]

    def cov(self, x, z=None):  # :64-66, :72-76
        if z is None:
            return self._predict(x, 4)[2]
        # cov(f, x, z) = block of the joint covariance over [x; z]
        xa, za = _stack_inputs(x, z)
        nx = _npoints(x)
        return np.asfortranarray(self._predict(xa, 4)[2][:nx, nx:]) if za else None

    def mean_and_var(self, x):  # :85-90
        m, v, _ = self._predict(x, 3)
        return m, v

    def mean_and_cov(self, x):  # :78-83
        m, _, c = self._predict(x, 5)
        return m, c


def _stack_inputs(x, z):
    x, z = _as_input(x), _as_input(z)
    if isinstance(x, ColVecs):
        return ColVecs(np.concatenate([x.X, (z.X if isinstance(z, ColVecs) else z.X.T)], axis=1)), True
    if isinstance(x, RowVecs):
        return RowVecs(np.concatenate([x.X, (z.X if isinstance(z, RowVecs) else z.X.T)], axis=0)), True
    return np.concatenate([x, z]), True


def _posterior_exact(fx: FiniteGP, y, zero_mean: bool = False) -> PosteriorGP:
    """posterior(fx::FiniteGP, y) — src/exact_gpr_posterior.jl:29-35.  One device call: Gram, Cholesky,
    α and logpdf(fx, y) (kept as .logpdf_value) from a single factorisation."""
    y = _check_y(fx, y)
    if y.ndim != 1:
        raise TypeError("posterior expects a vector of observations")
    f = fx.f
    if isinstance(f, PosteriorGP):
        return _posterior_sequential(fx, y)
    if not isinstance(f, GP):
        raise TypeError("posterior: unsupported GP type")
    ctx = f.context()
    dt = np.result_type(_input_dtype(fx.x), np.float32 if y.dtype == np.float32 else np.float64).type
    m = _Marshal(dt)
    px = m.points(fx.x)
    kk = m.kernel(f.kernel, px.d)
    nz = m.noise(fx.sigma2, px.n)
    mean = None if zero_mean else _mean_vector(f.mean_fn, fx.x, dt)
    yv = m.arr(y)
    delta = yv - mean if mean is not None else yv.copy()  # δ = y - m  (:32)
    mean = None if mean is None else m.arr(mean)
    alpha = np.empty(px.n, dtype=dt)
    lp = np.empty(1, dtype=dt)
    h = C.c_void_p()
    check(ctx.lib.gp_posterior_fit(ctx.handle, C.byref(kk), C.byref(px), C.byref(nz), m.ptr(mean), yv.ctypes.data,
                                   C.byref(h), alpha.ctypes.data, lp.ctypes.data))
    return PosteriorGP(f, _PostData(alpha, _Factor(ctx, h, px.n, dt), fx.x, delta), lp[0])


def _posterior_sequential(fx: FiniteGP, y) -> PosteriorGP:
    """posterior(fx::FiniteGP{<:PosteriorGP}, y) — src/exact_gpr_posterior.jl:46-56: the device-resident factor is
    extended by update_chol (src/util/common_covmat_ops.jl:38-42) instead of being recomputed."""
    post = fx.f
    prior = post.prior
    fac = post.data.C
    dt = fac.dtype
    m = _Marshal(dt)
    px = m.points(fx.x)
    nz = m.noise(fx.sigma2, px.n)
    m2 = _mean_vector(prior.mean_fn, fx.x, dt)
    d2 = m.arr(y) - m2 if m2 is not None else m.arr(y).copy()          # δ2 = y - m2           (:48-49)
    delta = m.arr(np.concatenate([post.data.delta, d2]))                # δ = vcat(δ_old, δ2)  (:52)
    n = delta.shape[0]
    alpha = np.empty(n, dtype=dt)
    lp = np.empty(1, dtype=dt)
    h = C.c_void_p()
    check(fac.ctx.lib.gp_posterior_update(fac.handle, C.byref(px), C.byref(nz), delta.ctypes.data, C.byref(h),
                                          alpha.ctypes.data, lp.ctypes.data))
    x_all, _ = _stack_inputs(post.data.x, fx.x)                         # x = vcat(x_old, x2)  (:54)
    return PosteriorGP(prior, _PostData(alpha, _Factor(fac.ctx, h, n, dt), x_all, delta), lp[0])


def _joint_call(fx: FiniteGP, name: str, payload: np.ndarray):
    """Shared marshalling of gp_{posterior,vfe}_{logpdf,rand}: payload is ns × ncols (y* columns or standard normals)."""
    f = fx.f
    if isinstance(f, PosteriorGP):
        ctx, h, dt, prior, fn = f.data.C.ctx, f.data.C.handle, f.data.C.dtype, f.prior, "gp_posterior_" + name
    elif isinstance(f, ApproxPosteriorGP):
        ctx, h, dt, prior, fn = f._state.ctx, f._state.handle, f._dtype, f.prior, "gp_vfe_" + name
    else:
        raise TypeError(f"{name}: unsupported GP type")
    m = _Marshal(dt)
    px = m.points(fx.x)
    nz = m.noise(fx.sigma2, px.n)
    pm = _mean_vector(prior.mean_fn, fx.x, dt)
    pm = None if pm is None else m.arr(pm)
    P = m.arr(payload, order="F")
    ncols = P.shape[1]
    if name == "logpdf":
        out = np.empty(ncols, dtype=dt)
        check(getattr(ctx.lib, fn)(h, C.byref(px), m.ptr(pm), C.byref(nz), P.ctypes.data, P.shape[0], ncols, out.ctypes.data))
    else:
        out = np.empty((px.n, ncols), dtype=dt, order="F")
        check(getattr(ctx.lib, fn)(h, C.byref(px), m.ptr(pm), C.byref(nz), P.ctypes.data, ncols, out.ctypes.data))
    return out


def rand(fx: FiniteGP, N: Optional[int] = None, rng=None, xi=None):
    """rand([rng,] fx[, N]) — src/finite_gp_projection.jl:233-240: m .+ C.U' * randn(rng, n, N).  Factor and triangular
    product run on the device for priors AND posteriors (exact / VFE: the N*×N* predictive covariance is built and
    factored in HBM against the resident factor — nothing is refitted); the standard normals are drawn on the host
    (`rng`: numpy Generator) or passed in as `xi` (n × N) for reproducible parity checks."""
    f = fx.f
    n = len(fx)
    ncols = 1 if N is None else int(N)
    if isinstance(f, GP):
        fac = fx._prior_factor()
        dt = fac.dtype
    elif isinstance(f, (PosteriorGP, ApproxPosteriorGP)):
        dt = f.data.C.dtype if isinstance(f, PosteriorGP) else f._dtype
    else:
        raise TypeError("rand: unsupported GP type")
    if xi is None:
        xi = (rng or np.random.default_rng()).standard_normal((n, ncols))
    xi = np.asfortranarray(np.asarray(xi, dtype=dt).reshape(n, ncols))
    if isinstance(f, GP):
        out = np.empty((n, ncols), dtype=dt, order="F")
        check(fac.ctx.lib.gp_posterior_factor_mul(fac.handle, xi.ctypes.data, ncols, out.ctypes.data))
        out += np.asarray(f.mean(fx.x), dtype=dt)[:, None]
    else:
        out = _joint_call(fx, "rand", xi)
    return out[:, 0] if N is None else out


def rand_(fx: FiniteGP, out: np.ndarray, rng=None, xi=None) -> np.ndarray:
    """rand!(rng, fx, y) — src/finite_gp_projection.jl:271-277 (via _rand!): fills the vector (n) or matrix (n × N) `out`
    in place and returns it."""
    out_arr = np.asarray(out)
    if out_arr.shape[0] != len(fx):
        raise ValueError(f"DimensionMismatch: length(fx) = {len(fx)} but the output has {out_arr.shape[0]} rows")
    res = rand(fx, None if out_arr.ndim == 1 else out_arr.shape[1], rng=rng, xi=xi)
    out[...] = res
    return out


def _logpdf_posterior(fx: FiniteGP, y):
    """logpdf(post(x*, Σy*), y*) — the generic FiniteGP path of the reference (src/finite_gp_projection.jl:306-311 over the
    posterior's mean_and_cov, src/exact_gpr_posterior.jl:78-83 / src/sparse_approximations.jl:205-210) as ONE device call."""
    out = _joint_call(fx, "logpdf", y if y.ndim == 2 else y[:, None])
    return out[0] if y.ndim == 1 else out


# Distribution-style accessors on FiniteGP (src/finite_gp_projection.jl:53,96,114,133,154,203)
def mean(fx_or_f, x=None):
    if isinstance(fx_or_f, FiniteGP):
        return fx_or_f.f.mean(fx_or_f.x)
    return fx_or_f.mean(x)


def var(fx_or_f, x=None):
    if isinstance(fx_or_f, FiniteGP):
        return fx_or_f.f.var(fx_or_f.x) + fx_or_f.noise_vector()
    return fx_or_f.var(x)


def cov(fx_or_f, x=None, z=None):
    if isinstance(fx_or_f, FiniteGP):
        fx = fx_or_f
        if isinstance(x, FiniteGP):  # cov(fx, gx)  :177-180
            assert fx.f is x.f
            return fx.f.cov(fx.x, x.x)
        Cm = np.array(fx.f.cov(fx.x))
        Cm[np.diag_indices_from(Cm)] += fx.noise_vector()
        return Cm
    return fx_or_f.cov(x, z)


def mean_and_var(fx_or_f, x=None):
    if isinstance(fx_or_f, FiniteGP):
        m, v = fx_or_f.f.mean_and_var(fx_or_f.x)
        return m, v + fx_or_f.noise_vector()
    return fx_or_f.mean_and_var(x)


def mean_and_cov(fx_or_f, x=None):
    if isinstance(fx_or_f, FiniteGP):
        return mean(fx_or_f), cov(fx_or_f)
    return fx_or_f.mean_and_cov(x)


def marginals(fx: FiniteGP):
    """marginals(fx) = Normal.(m, sqrt.(c)) -> (mean, std) arrays (src/finite_gp_projection.jl:203-206)."""
    m, c = mean_and_var(fx)
    return m, np.sqrt(c)


# --------------------------------------------------------------------------------------------
# Sparse approximations: VFE / DTC (src/sparse_approximations.jl)
# --------------------------------------------------------------------------------------------
@dataclass(eq=False)
class VFE:
    """VFE(fz::FiniteGP) — src/sparse_approximations.jl:12-14."""

    fz: FiniteGP


@dataclass(eq=False)
class DTC:
    """DTC(fz::FiniteGP) — src/sparse_approximations.jl:21-23."""

    fz: FiniteGP


class ExactInference:  # src/exact_gpr_posterior.jl:6-12
    pass


class _VfeState:
    def __init__(self, ctx, handle):
        self.ctx, self.handle = ctx, handle
        self._fin = weakref.finalize(self, ctx.lib.gp_vfe_free, handle)


def _vfe_call(approx, fx: FiniteGP, y, want_post: bool):
    y = _check_y(fx, y)
    if y.ndim != 1:
        raise TypeError("expected a vector of observations")
    f = fx.f
    if approx.fz.f is not f:  # @assert vfe.fz.f === fx.f  (:59, :249, :283)
        raise AssertionError("vfe.fz.f === fx.f")
    if not isinstance(f, GP):
        raise TypeError("VFE/DTC: unsupported prior type")
    ctx = f.context()
    dt = np.result_type(_input_dtype(fx.x), np.float32 if y.dtype == np.float32 else np.float64).type
    m = _Marshal(dt)
    px, pz = m.points(fx.x), m.points(approx.fz.x)
    kk = m.kernel(f.kernel, px.d)
    nz = m.noise(fx.sigma2, px.n)
    jit = np.asarray(approx.fz.sigma2)
    if jit.ndim != 0:
        raise NotImplementedError("inducing-point noise must be a scalar jitter")
    mean = _mean_vector(f.mean_fn, fx.x, dt)
    mean = None if mean is None else m.arr(mean)
    yv = m.arr(y)
    obj = np.empty(1, dtype=dt)
    h = C.c_void_p()
    check(ctx.lib.gp_vfe_fit(ctx.handle, C.byref(kk), C.byref(px), C.byref(pz), C.byref(nz), float(jit), m.ptr(mean),
                             yv.ctypes.data, 0 if isinstance(approx, VFE) else 1,
                             C.byref(h) if want_post else None, obj.ctypes.data))
    return (h if want_post else None), obj[0], ctx, dt, pz.n


class _VfeData(Mapping):
    """Lazy read-only Mapping over an ApproxPosteriorGP's device-resident cache (field names of src/sparse_approximations.jl:73, ASCII):
    iteration, len, keys / items / values / get behave like the plain dict this property once returned; every read of a field fetches it
    from the device (nothing is cached on the host)."""

    _FIELDS = ("alpha", "m_eps", "U", "Lam_U", "b_y")

    def __init__(self, post):
        self._post = post

    def __iter__(self):
        return iter(self._FIELDS)

    def __len__(self):
        return len(self._FIELDS)

    def __contains__(self, k):
        return k in self._FIELDS

    def __getattr__(self, k):
        if k.startswith("_") or k not in self._FIELDS:
            raise AttributeError(k)
        return self[k]

    def __getitem__(self, k):
        p = self._post
        st, dt, m = p._state, p._dtype, p._m
        lib = st.ctx.lib
        if k in ("alpha", "m_eps"):
            a = np.empty(m, dtype=dt)
            check(lib.gp_vfe_get(st.handle, a.ctypes.data if k == "alpha" else None, a.ctypes.data if k == "m_eps" else None))
            return a
        if k in ("U", "Lam_U"):
            out = np.empty((m, m), dtype=dt, order="F")
            check(lib.gp_vfe_get_factors(st.handle, out.ctypes.data if k == "U" else None, out.ctypes.data if k == "Lam_U" else None))
            return out
        if k == "b_y":
            n = int(lib.gp_vfe_n(st.handle))
            out = np.empty(max(n, 0), dtype=dt)
            check(lib.gp_vfe_get_by(st.handle, out.ctypes.data))
            return out
        raise KeyError(k)


class ApproxPosteriorGP(AbstractGP):
    """src/sparse_approximations.jl:25-29; the cache (:73) lives on the device."""

    def __init__(self, approx, prior: GP, state: _VfeState, dtype, m: int, objective):
        self.approx, self.prior, self._state, self._dtype, self._m = approx, prior, state, dtype, m
        self.objective = objective

    @property
    def data(self):
        """The cache of src/sparse_approximations.jl:73 — (m_ε, Λ_ε, U, α, b_y, …) — read from the device on demand: `alpha`, `m_eps`
        (M-vectors), `U` = cholesky(cov(fz)).U and `Lam_U` = Λ_ε.U (M×M upper, column-major), `b_y` (N-vector).  Mapping access
        (`data["alpha"]`) and attribute access (`data.alpha`) both work; B_εf is never materialised (it is N×M)."""
        return _VfeData(self)

    def _predict(self, x, what):
        st = self._state
        mm = _Marshal(self._dtype)
        px = mm.points(x)
        pm = _mean_vector(self.prior.mean_fn, x, self._dtype)
        pm = None if pm is None else mm.arr(pm)
        mean = np.empty(px.n, dtype=self._dtype) if what & 1 else None
        var = np.empty(px.n, dtype=self._dtype) if what & 2 else None
        cov = np.empty((px.n, px.n), dtype=self._dtype, order="F") if what & 4 else None
        check(st.ctx.lib.gp_vfe_predict(st.handle, C.byref(px), mm.ptr(pm), what, mm.ptr(mean), mm.ptr(var), mm.ptr(cov)))
        return mean, var, cov

    def mean(self, x=None):  # :183-185
        if x is None:
            return super().mean()
        return self._predict(x, 1)[0]

    def var(self, x):  # :192-195
        return self._predict(x, 2)[1]

    def cov(self, x, z=None):  # :187-190, :197-203
        if z is None:
            return self._predict(x, 4)[2]
        xa, _ = _stack_inputs(x, z)  # cov(f, x, z) = off-diagonal block of the joint covariance over [x; z]
        nx = _npoints(x)
        return np.asfortranarray(self._predict(xa, 4)[2][:nx, nx:])

    def mean_and_var(self, x):  # :212-217
        return self._predict(x, 3)[:2]

    def mean_and_cov(self, x):  # :205-210
        m, _, c = self._predict(x, 5)
        return m, c


    def objective_grad(self, wrt_x: bool = False) -> dict:
        """Gradient of the objective this posterior was fitted with (`self.objective`: elbo for VFE, approx_log_evidence for DTC —
        src/sparse_approximations.jl:248-254, :282-286) at its own parameters, from the state resident on the device (gp_vfe_grad): what an AD backend
        computes for examples/0-intro-1d/script.jl:385-394.  Keys as `logpdf_and_grad`: "variance", "scale" (None / float / vector), "noise" (the sum
        Σ_i ∂/∂Σy_ii — the gradient for a scalar σ²) and "noise_diag" (the vector), "y", "mean" (= −"y"), "z" in the shape of the pseudo-input
        container's array (fp64 posteriors only), and with wrt_x "x" in the shape of the observations' container (all observations seen so far, arrival order, as a
        RowVecs-shaped (N, D) array when the posterior has been updated)."""
        st = self._state
        dt = self._dtype
        n = int(st.ctx.lib.gp_vfe_n(st.handle))
        zin = _as_input(self.approx.fz.x)
        mm = _Marshal(dt)
        pz = mm.points(self.approx.fz.x)
        nscale = mm.kernel(self.prior.kernel, pz.d).nscale
        dvar, dns = C.c_double(), C.c_double()
        dscale = (C.c_double * max(nscale, 1))()
        dnoise = np.empty(n, dtype=dt)
        dy = np.empty(n, dtype=dt)
        dz = None   # fp64 posteriors only: an fp32 fit's streamed B Bᵀ does not carry ∂/∂z (the C ABI refuses it, include/gpmi355.h)
        if np.dtype(dt) == np.float64:
            dz = np.empty((pz.n,) if pz.layout == 0 else ((pz.n, pz.d) if pz.layout == 1 else (pz.d, pz.n)), dtype=np.float64)
        dxb = np.empty((n, pz.d), dtype=dt) if wrt_x else None          # layout 1: point-contiguous (N, D)
        check(st.ctx.lib.gp_vfe_grad(st.handle, C.byref(dvar), dscale, C.byref(dns), dnoise.ctypes.data, dy.ctypes.data, None if dz is None else dz.ctypes.data, pz.layout,
                                     None if dxb is None else dxb.ctypes.data, 1))
        sc = None if nscale == 0 else (float(dscale[0]) if nscale == 1 else np.array([dscale[i] for i in range(nscale)]))
        g = {"variance": dvar.value, "scale": sc, "noise": dns.value, "noise_diag": dnoise, "y": dy, "mean": -dy}
        if dz is not None:
            if pz.layout == 0:
                g["z"] = dz
            else:
                nd = dz if pz.layout == 1 else dz.T
                g["z"] = np.ascontiguousarray(nd.T if isinstance(zin, ColVecs) else nd)
        if wrt_x:
            g["x"] = dxb[:, 0] if pz.d == 1 and pz.layout == 0 else dxb
        return g


def elbo_and_grad(a, fx: FiniteGP, y, wrt_x: bool = False) -> tuple:
    """(approx_log_evidence(a, fx, y), its gradient) in one fit + one backward pass on the device — the value / pullback pair an rrule of
    `elbo(VFE(f(z, jitter)), f(x, Σy), y)` needs (the reference differentiates the same expression by AD: examples/0-intro-1d/script.jl:385-394).
    "x" comes back in the shape of fx.x's array."""
    if not isinstance(a, (VFE, DTC)):
        raise TypeError("elbo_and_grad: VFE or DTC")
    post = posterior(a, fx, y)
    g = post.objective_grad(wrt_x)
    if wrt_x and "x" in g and np.ndim(g["x"]) == 2 and isinstance(_as_input(fx.x), ColVecs):
        g["x"] = np.ascontiguousarray(g["x"].T)
    return post.objective, g


def inducing_points(f: ApproxPosteriorGP):  # :219
    return f.approx.fz.x


def update_posterior(f_post_approx: ApproxPosteriorGP, fx: FiniteGP, y=None):
    """update_posterior(f_post_approx, fx, y)  — new observations, same pseudo-points (src/sparse_approximations.jl:87-121):
    the device continues its streamed reductions with the new points and re-finalises the M×M side.
    update_posterior(f_post_approx, fz)      — append pseudo-points (:131-176): bordered Cholesky of K_zz on the resident
    factor (update_chol), then the observations retained on the device are streamed once more for the NEW block rows of
    B Bᵀ / B b_y only (gp_vfe_append)."""
    if f_post_approx.prior is not fx.f:
        raise AssertionError("f_post_approx.prior === fx.f")
    st = f_post_approx._state
    dt = f_post_approx._dtype
    mm = _Marshal(dt)
    obj = np.empty(1, dtype=dt)
    h = C.c_void_p()
    if y is None:  # new pseudo-points: fx is fz
        pz = mm.points(fx.x)
        check(st.ctx.lib.gp_vfe_append(st.handle, C.byref(pz), C.byref(h), obj.ctypes.data))
        z_all, _ = _stack_inputs(f_post_approx.approx.fz.x, fx.x)                                     # z_new = vcat(z_old, z)  (:160)
        approx = type(f_post_approx.approx)(f_post_approx.prior(z_all, f_post_approx.approx.fz.sigma2))  # _update_approx (:178-179)
        return ApproxPosteriorGP(approx, f_post_approx.prior, _VfeState(st.ctx, h), dt, f_post_approx._m + pz.n, obj[0])
    y = _check_y(fx, y)
    px = mm.points(fx.x)
    nz = mm.noise(fx.sigma2, px.n)
    mean = _mean_vector(f_post_approx.prior.mean_fn, fx.x, dt)
    mean = None if mean is None else mm.arr(mean)
    yv = mm.arr(y)
    check(st.ctx.lib.gp_vfe_update(st.handle, C.byref(px), C.byref(nz), mm.ptr(mean), yv.ctypes.data, C.byref(h), obj.ctypes.data))
    return ApproxPosteriorGP(f_post_approx.approx, f_post_approx.prior, _VfeState(st.ctx, h), dt, f_post_approx._m, obj[0])


def posterior(*args):
    """posterior(fx, y)                      — src/exact_gpr_posterior.jl:29-35
    posterior(VFE(fz)|DTC(fz), fx, y)     — src/sparse_approximations.jl:58-75
    posterior(ExactInference(), fx, y)    — src/exact_gpr_posterior.jl:8"""
    if len(args) == 2:
        return _posterior_exact(*args)
    if len(args) == 3:
        a, fx, y = args
        if isinstance(a, ExactInference):
            return _posterior_exact(fx, y)
        if isinstance(a, (VFE, DTC)):
            h, obj, ctx, dt, m = _vfe_call(a, fx, y, True)
            return ApproxPosteriorGP(a, fx.f, _VfeState(ctx, h), dt, m, obj)
    raise TypeError("posterior(fx, y) or posterior(approx, fx, y)")


def approx_log_evidence(a, fx: FiniteGP, y):
    """src/sparse_approximations.jl:248-252 (VFE), :282-286 (DTC); src/exact_gpr_posterior.jl:10-12."""
    if isinstance(a, ExactInference):
        return logpdf(fx, y)
    if not isinstance(a, (VFE, DTC)):
        raise TypeError("approx_log_evidence: unsupported approximation")
    return _vfe_call(a, fx, y, False)[1]


def elbo(a, fx: FiniteGP, y):
    """elbo(vfe, fx, y) = approx_log_evidence(vfe, fx, y) (:254); elbo(::DTC) is deprecated upstream."""
    if not isinstance(a, VFE):
        raise TypeError("elbo is defined for VFE")
    return approx_log_evidence(a, fx, y)
