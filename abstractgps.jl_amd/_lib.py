"""ctypes binding of libgpmi355.so (include/gpmi355.h).  No fallback: if the shared library is not
built or cannot be loaded this raises — the product path never runs on the CPU."""
from __future__ import annotations

import ctypes as C
import os
import re
from pathlib import Path

PKG_DIR = Path(__file__).resolve().parent
CSRC = PKG_DIR / "csrc"
LIB_PATH = CSRC / "libgpmi355.so"
HEADER = PKG_DIR.parent / "include" / "gpmi355.h"


class gp_kernel(C.Structure):
    _fields_ = [("kind", C.c_int32), ("dtype", C.c_int32), ("variance", C.c_double), ("nscale", C.c_int32),
                ("scale", C.POINTER(C.c_double))]


class gp_points(C.Structure):
    _fields_ = [("data", C.c_void_p), ("n", C.c_int64), ("d", C.c_int32), ("layout", C.c_int32)]


class gp_noise(C.Structure):
    _fields_ = [("kind", C.c_int32), ("s", C.c_double), ("diag", C.c_void_p)]


class gp_timings(C.Structure):
    _fields_ = [("assemble_ms", C.c_double), ("potrf_ms", C.c_double), ("solve_ms", C.c_double),
                ("total_ms", C.c_double), ("gemm_ms", C.c_double), ("gemm_flops", C.c_double),
                ("gemm_launches", C.c_int64), ("gemm_bytes", C.c_double)]


class gp_grid(C.Structure):
    _fields_ = [("P", C.c_int32), ("p", C.c_int32), ("Q", C.c_int32), ("q", C.c_int32), ("tb", C.c_int32),
                ("lower", C.c_int32)]


vp, i32, i64, dbl = C.c_void_p, C.c_int32, C.c_int64, C.c_double
PK, PP, PN, PG = C.POINTER(gp_kernel), C.POINTER(gp_points), C.POINTER(gp_noise), C.POINTER(gp_grid)

# name -> (restype, argtypes); must cover every function declared in include/gpmi355.h
PROTOTYPES = {
    "gp_ctx_create": (i32, [C.POINTER(vp), i32, vp]),
    "gp_ctx_create_multi": (i32, [C.POINTER(vp), C.POINTER(i32), i32, i32, i32, i32]),
    "gp_ctx_multi_info": (i32, [vp, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)]),
    "gp_multi_schedule_trace": (i32, [i32, i32, i32, i32, i32, C.c_char_p]),
    "gp_ctx_multi_stats": (i32, [vp, C.POINTER(i64), C.POINTER(i64), C.POINTER(i64)]),
    "gp_multi_solve_trace": (i32, [i32, i32, i32, C.c_char_p]),
    "gp_multi_solve_trace_ex": (i32, [i32, i32, i32, i32, C.c_char_p]),
    "gp_ctx_destroy": (i32, [vp]),
    "gp_ctx_set_param": (i32, [vp, C.c_char_p, i64]),
    "gp_ctx_get_param": (i32, [vp, C.c_char_p, C.POINTER(i64)]),
    "gp_get_timings": (i32, [vp, C.POINTER(gp_timings)]),
    "gp_last_error": (C.c_char_p, []),
    "gp_abi_version": (i32, []),
    "gp_kernelmatrix": (i32, [vp, PK, PP, PP, vp]),
    "gp_logpdf": (i32, [vp, PK, PP, PN, vp, vp, i64, i32, vp]),
    "gp_logpdf_terms": (i32, [vp, PK, PP, PN, vp, vp, i64, i32, vp, vp]),
    "gp_posterior_logdet": (i32, [vp, C.POINTER(dbl)]),
    "gp_posterior_fit": (i32, [vp, PK, PP, PN, vp, vp, C.POINTER(vp), vp, vp]),
    "gp_posterior_predict": (i32, [vp, PP, vp, i32, vp, vp, vp]),
    "gp_posterior_get_factor": (i32, [vp, vp]),
    "gp_logpdf_grad": (i32, [vp, PK, PP, PN, vp, vp, vp, C.POINTER(dbl), C.POINTER(dbl), vp, vp, vp]),
    "gp_posterior_update": (i32, [vp, PP, PN, vp, C.POINTER(vp), vp, vp]),
    "gp_posterior_factor_mul": (i32, [vp, vp, i32, vp]),
    "gp_posterior_solve": (i32, [vp, vp, i32, vp]),
    "gp_posterior_n": (i64, [vp]),
    "gp_posterior_free": (i32, [vp]),
    "gp_vfe_fit": (i32, [vp, PK, PP, PP, PN, dbl, vp, vp, i32, C.POINTER(vp), vp]),
    "gp_vfe_update": (i32, [vp, PP, PN, vp, vp, C.POINTER(vp), vp]),
    "gp_vfe_predict": (i32, [vp, PP, vp, i32, vp, vp, vp]),
    "gp_vfe_append": (i32, [vp, PP, C.POINTER(vp), vp]),
    "gp_vfe_logpdf": (i32, [vp, PP, vp, PN, vp, i64, i32, vp]),
    "gp_vfe_rand": (i32, [vp, PP, vp, PN, vp, i32, vp]),
    "gp_vfe_m": (i64, [vp]),
    "gp_posterior_logpdf": (i32, [vp, PP, vp, PN, vp, i64, i32, vp]),
    "gp_posterior_rand": (i32, [vp, PP, vp, PN, vp, i32, vp]),
    "gp_ctx_trim": (i32, [vp]),
    "gp_vfe_get": (i32, [vp, vp, vp]),
    "gp_vfe_get_factors": (i32, [vp, vp, vp]),
    "gp_vfe_n": (i64, [vp]),
    "gp_vfe_get_by": (i32, [vp, vp]),
    "gp_vfe_grad": (i32, [vp, C.POINTER(dbl), C.POINTER(dbl), C.POINTER(dbl), vp, vp, vp, i32, vp, i32]),
    "gp_vfe_free": (i32, [vp]),
    "gpd_assemble": (i32, [vp, PK, vp, i64, i64, i32, vp, PG, vp, i64, i64, i64]),
    "gpd_potrf": (i32, [vp, vp, i64, i64, i64, vp, i32, i64, vp]),
    "gpd_trsm": (i32, [vp, vp, i64, i64, vp, i64, i64]),
    "gpd_inv_lower": (i32, [vp, vp, i64, i64, vp, i64, vp, vp]),
    "gpd_trsm_inv": (i32, [vp, vp, i64, i64, vp, i64, i64, vp, i64]),
    "gpd_gemm_nt": (i32, [vp, vp, i64, vp, i64, vp, i64, i64, i64, i64, PG, i64, i64]),
    "gpd_trsv": (i32, [vp, vp, i64, i64, vp, i64, i32, i32]),
    "gpd_gemv_t": (i32, [vp, vp, i64, i64, i64, vp, vp]),
    "gpd_rowsumsq": (i32, [vp, vp, i64, i64, i64, vp]),
    "gpd_sync": (i32, [vp]),
    "gpd_gemm_time": (i32, [vp, C.POINTER(dbl), C.POINTER(i64)]),
    "gp_rccl_selftest": (i32, [i32, i64, C.POINTER(dbl)]),
    "gp_probe_mfma_f64": (i32, [vp, vp, vp, vp]),
    "gp_bench_mfma_f64": (i32, [vp, i32, C.POINTER(dbl)]),
    "gp_bench_mfma_f32": (i32, [vp, i32, i32, C.POINTER(dbl)]),
}


def header_functions() -> list[str]:
    """Names of every function declared in include/gpmi355.h (used by the CPU symbol test)."""
    txt = HEADER.read_text()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(gpd?_[a-z0-9_]+)\s*\(", txt)))


_lib = None


def load() -> C.CDLL:
    """Load libgpmi355.so and attach prototypes.  Raises (never falls back) when it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    path = Path(os.environ.get("GPMI355_LIB", LIB_PATH))
    if not path.exists():
        raise ImportError(f"{path} is not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
    lib = C.CDLL(str(path), mode=C.RTLD_GLOBAL)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype, fn.argtypes = res, args
    if lib.gp_abi_version() != 4:
        raise ImportError("libgpmi355.so ABI version mismatch")
    _lib = lib
    return lib


class PosDefException(Exception):
    """Mirror of LinearAlgebra.PosDefException(info) (reference src/finite_gp_projection.jl:308)."""

    def __init__(self, info: int):
        super().__init__(f"matrix is not positive definite; leading minor of order {info}")
        self.info = info


class GpmiError(RuntimeError):
    pass


def check(rc: int) -> None:
    if rc == 0:
        return
    if rc > 0:
        raise PosDefException(rc)
    msg = load().gp_last_error().decode(errors="replace")
    if rc > -1000:
        raise ValueError(f"libgpmi355: {msg} (status {rc})")
    raise GpmiError(f"libgpmi355: {msg} (status {rc})")
