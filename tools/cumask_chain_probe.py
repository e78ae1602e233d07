"""Round 5 (the review's item 5c, measured on ONE device): what a CU-masked stream pair would buy a rank of the multi-device driver.

In the multi-device schedule the diagonal owner factors the NB×NB diagonal block of column k+d (a chain of NB/128 leaf launches + few-tile
updates: latency-bound, at most NB/64 workgroups) WHILE the bulk update of step k occupies the same GPU.  Round 4 measured on one GPU that a
co-resident 64-column leaf shares each CU's one DP pipe with a GEMM wave (≈ 5× slower) and that the 128-column leaf does not start before a GEMM
launch ends; tools/grid_model.py prices that guess (`cores=5`: 8 devices 74 % -> 54 % of peak).  This probe replaces the guess by two measured
numbers per (NB, r): the chain's duration on a stream masked to r CUs (hipExtStreamCreateWithCUMask: bit i = CU i/8 of XCC i%8, so r/8 CUs of every
XCD) while the bulk update runs on the other 256 − r, and the bulk update's slow-down there — against the unmasked pair (chain on a high-priority
stream beside the update) and both kernels alone.  Both run through the library's own device-level entry points (gpd_potrf / gpd_gemm_nt) on
contexts created over caller-owned streams, with the rank-context settings (no stream-K tails; leaf_cols 64 and 128).
Bulk = one rank's share of a C4 step on an 8×1 grid: C[7 168 × 57 344] −= A·Bᵀ, K = NB.   One JSON line per case.
"""
import ctypes as C
import json
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import abstractgps_jl_amd as agp  # noqa: E402
from abstractgps_jl_amd._lib import check  # noqa: E402

hip = C.CDLL("libamdhip64.so")
vp = C.c_void_p


def hck(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what}: hip error {rc}")


def malloc(nbytes):
    p = vp()
    hck(hip.hipMalloc(C.byref(p), C.c_size_t(nbytes)), "hipMalloc")
    return p


def masked_stream(bits):
    words = 8
    m = (C.c_uint32 * words)(*[sum(1 << (i % 32) for i in bits if i // 32 == w) for w in range(words)])
    s = vp()
    hck(hip.hipExtStreamCreateWithCUMask(C.byref(s), C.c_uint32(words), m), "hipExtStreamCreateWithCUMask")
    return s


def prio_stream(high):
    lo, hi_ = C.c_int(), C.c_int()
    hck(hip.hipDeviceGetStreamPriorityRange(C.byref(lo), C.byref(hi_)), "prio range")
    s = vp()
    hck(hip.hipStreamCreateWithPriority(C.byref(s), C.c_uint(1), hi_.value if high else lo.value), "hipStreamCreateWithPriority")
    return s


def event():
    e = vp()
    hck(hip.hipEventCreate(C.byref(e)), "hipEventCreate")
    return e


def elapsed(e0, e1):
    ms = C.c_float()
    hck(hip.hipEventElapsedTime(C.byref(ms), e0, e1), "hipEventElapsedTime")
    return float(ms.value)


def main():
    hck(hip.hipSetDevice(0), "hipSetDevice")
    M, N = 7168, 57344
    rng = np.random.default_rng(0)
    lib = agp._lib.load()
    info = malloc(64)
    logdet = malloc(1024)
    ev = [event() for _ in range(4)]
    for NB in (1024, 512):
        ld = NB + 32
        # SPD diagonal block + a pristine copy
        idx = np.arange(NB)
        blk = np.exp(-np.abs(idx[:, None] - idx[None, :]) / 40.0) + 0.05 * np.eye(NB)
        host = np.zeros((NB + 128, ld))
        host[:NB, :NB] = blk
        a_keep, a_work = malloc(host.nbytes), malloc(host.nbytes)
        hck(hip.hipMemcpy(a_keep, host.ctypes.data_as(vp), C.c_size_t(host.nbytes), 1), "H2D")
        # bulk operands
        lda, ldc = NB + 32, N + 32
        opA = rng.standard_normal((N + 128, lda))
        A2 = malloc(opA.nbytes)
        hck(hip.hipMemcpy(A2, opA.ctypes.data_as(vp), C.c_size_t(opA.nbytes), 1), "H2D")
        Cm = malloc((M + 128) * ldc * 8)
        hck(hip.hipMemset(Cm, 0, C.c_size_t((M + 128) * ldc * 8)), "memset")
        flops = 2.0 * M * N * NB

        def run_case(tag, s_chain, s_bulk, leaf_cols, together):
            cc = agp.Context(0, stream=s_chain.value)
            cb = agp.Context(0, stream=s_bulk.value)
            for c_ in (cc, cb):
                c_.set_param("gemm_streamk", 0)  # rank contexts: multi_gemm_streamk = 0
            cc.set_param("leaf_cols", leaf_cols)

            def chain():
                hck(hip.hipMemcpyAsync(a_work, a_keep, C.c_size_t(host.nbytes), 3, s_chain), "D2D")
                hck(hip.hipMemsetAsync(info, 0, C.c_size_t(4), s_chain), "memset")
                hck(hip.hipEventRecord(ev[0], s_chain), "rec")
                check(lib.gpd_potrf(cc.handle, a_work, ld, NB, NB, info, 0, NB, logdet))
                hck(hip.hipEventRecord(ev[1], s_chain), "rec")

            def bulk():
                hck(hip.hipEventRecord(ev[2], s_bulk), "rec")
                check(lib.gpd_gemm_nt(cb.handle, Cm, ldc, A2, lda, A2, lda, M, N, NB, None, 0, 0))
                hck(hip.hipEventRecord(ev[3], s_bulk), "rec")

            tc, tb = [], []
            for rep in range(6):
                if together:
                    bulk()
                    time.sleep(0.0008)  # the update is running when the chain is queued
                    chain()
                else:
                    chain()
                    hck(hip.hipDeviceSynchronize(), "sync")
                    bulk()
                hck(hip.hipDeviceSynchronize(), "sync")
                if rep:
                    tc.append(elapsed(ev[0], ev[1]))
                    tb.append(elapsed(ev[2], ev[3]))
            i32 = C.c_int32()
            hck(hip.hipMemcpy(C.byref(i32), info, C.c_size_t(4), 2), "D2H")
            out = {"NB": NB, "case": tag, "leaf_cols": leaf_cols, "together": together, "chain_ms": float(np.median(tc)), "bulk_ms": float(np.median(tb)),
                   "bulk_tflops": flops / float(np.median(tb)) / 1e9, "potrf_info": int(i32.value)}
            print(json.dumps(out), flush=True)
            cc.close()
            cb.close()
            return out

        plain_hi, plain_lo = prio_stream(True), prio_stream(False)
        for lc in (64, 128):
            run_case("unmasked, alone", plain_hi, plain_lo, lc, False)
            run_case("unmasked, chain beside the update (high-priority stream)", plain_hi, plain_lo, lc, True)
        for r in (8, 16, 32):
            sc = masked_stream(range(0, r))
            sb = masked_stream(range(r, 256))
            for lc in (64, 128):
                run_case(f"chain on {r} CUs alone / update on {256 - r} CUs alone", sc, sb, lc, False)
                run_case(f"chain on {r} CUs BESIDE the update on {256 - r} CUs", sc, sb, lc, True)
        for p in (a_keep, a_work, A2, Cm):
            hip.hipFree(p)


if __name__ == "__main__":
    main()
