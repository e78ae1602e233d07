// gpmi355.hip — host engine + C ABI (include/gpmi355.h) of libgpmi355.so.
//
// Host orchestration of the kernels in kernels.hpp:
//   assemble (kmat)  ->  right-looking blocked Cholesky with a recursive panel and look-ahead
//   (potf2_64 / trsm_64 / MFMA gemm_nt_sub)  ->  logdet + forward solve (carried as extra RHS rows
//   of the factorisation)  ->  backward substitution  ->  logpdf scalar and α.
// Mirrors, on the device, reference src/finite_gp_projection.jl:306-311 and
// src/exact_gpr_posterior.jl:29-35, 60-90 (see include/gpmi355.h for the per-entry citations).
#include "kernels.hpp"  // includes engine.hpp (shared structs, registry, error helpers)

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <set>
#include <memory>
#include <string>
#include <type_traits>
#include <unordered_map>
#include <vector>

using namespace gpmi;

// ------------------------------------------------------------------------------------------------
// errors
// ------------------------------------------------------------------------------------------------
static thread_local std::string g_err;
int32_t set_err_text(int32_t status, const std::string& msg) {
    g_err = msg;
    return status;
}
int32_t set_hip_err(hipError_t e, const char* what, int line) {
    char buf[512];
    snprintf(buf, sizeof buf, "HIP error %d (%s) at %s:%d", (int)e, hipGetErrorString(e), what, line);
    g_err = buf;
    return -1000 - (int)e;
}
int32_t set_arg_err(int i, const char* msg) {
    g_err = std::string("invalid argument ") + std::to_string(i) + ": " + msg;
    return -i;
}
std::mutex g_reg_mu;
std::set<void*> g_live;
void reg_add(void* p) {
    std::lock_guard<std::mutex> l(g_reg_mu);
    g_live.insert(p);
}
bool reg_take(void* p) {
    std::lock_guard<std::mutex> l(g_reg_mu);
    return g_live.erase(p) > 0;
}
bool reg_has(void* p) {
    std::lock_guard<std::mutex> l(g_reg_mu);
    return g_live.count(p) > 0;
}
static void pool_drop(gp_ctx* c, size_t i) {
    c->pool_bytes -= c->pool[i].bytes;
    (void)hipFree(c->pool[i].p);
    c->pool.erase(c->pool.begin() + i);
}
int32_t ctx_alloc(gp_ctx* c, size_t bytes, void** out) {
    size_t best = (size_t)-1;
    int bi = -1;
    for (size_t i = 0; i < c->pool.size(); ++i)
        if (c->pool[i].bytes >= bytes && c->pool[i].bytes < best && c->pool[i].bytes <= bytes + bytes / 4 + (1 << 20)) {
            best = c->pool[i].bytes;
            bi = (int)i;
        }
    if (bi >= 0) {
        *out = c->pool[bi].p;
        c->blk[*out] = c->pool[bi].bytes;  // the block keeps its true size
        c->pool_bytes -= c->pool[bi].bytes;
        c->pool.erase(c->pool.begin() + bi);
        return 0;
    }
    hipError_t e = hipMalloc(out, bytes);
    if (e != hipSuccess) {  // drop the cache and retry once
        (void)hipGetLastError();
        while (!c->pool.empty()) pool_drop(c, c->pool.size() - 1);
        e = hipMalloc(out, bytes);
    }
    if (e != hipSuccess) return set_hip_err(e, "hipMalloc", __LINE__);
    c->blk[*out] = bytes;
    return 0;
}
void ctx_release(gp_ctx* c, void* p, size_t /*requested*/) {
    if (!p) return;
    size_t bytes = 0;
    auto it = c->blk.find(p);
    if (it != c->blk.end()) {
        bytes = it->second;
        c->blk.erase(it);
    }
    if (c->dead || bytes == 0 || c->pool_cap == 0) {  // pool_cap_mb = 0: nothing is cached
        (void)hipFree(p);
        return;
    }
    if (bytes > c->pool_cap) {
        // ONE block larger than the cap may stay cached: the factor of N > 110 000 points is 100+ GB and re-allocating it costs seconds per fit (N = 131 072: 17.2 s per
        // pair with the block freed and allocated again against ≈ 12 s of work).  It takes the whole cache (everything else goes, a previous oversize block included);
        // ctx_alloc drops the cache and retries when an allocation fails, and gp_ctx_trim returns it like any other block.
        while (!c->pool.empty()) pool_drop(c, c->pool.size() - 1);
        c->pool.push_back({p, bytes});
        c->pool_bytes += bytes;
        return;
    }
    c->pool.push_back({p, bytes});
    c->pool_bytes += bytes;
    // bound the cache by bytes and by count (a VFE fit cycles through ~16 buffers, its gradient pass through ~25 more); oldest blocks go first — an oversize block (above) is the last to go and
    // keeps 4 GiB of room beside it for the small buffers of the calls that follow
    size_t big = 0;
    for (const auto& b : c->pool) big = std::max(big, b.bytes);
    const size_t limit = big > c->pool_cap ? big + ((size_t)4 << 30) : c->pool_cap;
    while (c->pool.size() > 1 && (c->pool_bytes > limit || c->pool.size() > 96)) pool_drop(c, (c->pool[0].bytes > c->pool_cap) ? 1 : 0);
}
void ctx_unref(gp_ctx* c) {
    if (--c->refs != 0) return;
    (void)hipSetDevice(c->device);
    for (auto& b : c->pool) (void)hipFree(b.p);
    for (auto& kv : c->blk) (void)hipFree(kv.first);
    for (auto e : c->ev_pool) (void)hipEventDestroy(e);
    for (auto e : c->ev_phase)
        if (e) (void)hipEventDestroy(e);
    if (c->info_dev) (void)hipFree(c->info_dev);
    if (c->pin) (void)hipHostFree(c->pin);
    if (c->ticket_dev) (void)hipFree(c->ticket_dev);
    if (c->w_ws) (void)hipFree(c->w_ws);
    if (c->scal_dev) (void)hipFree(c->scal_dev);
    if (c->sp) (void)hipStreamDestroy(c->sp);
    if (c->sq) (void)hipStreamDestroy(c->sq);
    if (c->own_sm && c->sm) (void)hipStreamDestroy(c->sm);
    delete c;
}
// the ctx's third stream (high priority, like the panel stream), created and primed on first use
int32_t ctx_third_stream(gp_ctx* c, hipStream_t* out) {
    if (!c->sq) {
        HIPCHK(hipSetDevice(c->device));
        int lo = 0, hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
        HIPCHK(hipStreamCreateWithPriority(&c->sq, hipStreamNonBlocking, hi));
        RC(ctx_prime_stream(c, c->sq));
    }
    *out = c->sq;
    return 0;
}
void* ctx_pinned(gp_ctx* c, size_t bytes) {
    if (bytes > ((size_t)1 << 30)) return nullptr;
    if (c->pin && c->pin_bytes >= bytes) return c->pin;
    if (c->pin) (void)hipHostFree(c->pin);
    c->pin = nullptr;
    c->pin_bytes = 0;
    const size_t want = bytes + bytes / 4 + 4096;
    if (hipHostMalloc(&c->pin, want, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        c->pin = nullptr;
        return nullptr;
    }
    c->pin_bytes = want;
    return c->pin;
}
int32_t ctx_event(gp_ctx* c, hipEvent_t* out, bool timing) {
    // timing events are separate objects (created on demand, pooled)
    if (c->ev_used == c->ev_pool.size()) {
        hipEvent_t e;
        HIPCHK(hipEventCreate(&e));
        c->ev_pool.push_back(e);
    }
    (void)timing;
    *out = c->ev_pool[c->ev_used++];
    return 0;
}
int32_t ctx_scal(gp_ctx* c, long n) {
    if (c->scal_cap >= n) return 0;
    if (c->scal_dev) (void)hipFree(c->scal_dev);
    c->scal_cap = round_up(n, 1024);
    HIPCHK(hipMalloc((void**)&c->scal_dev, sizeof(double) * c->scal_cap));
    return 0;
}

__global__ void prime_kernel(int* p) {
    if (p && threadIdx.x == 0) *p = 0;
}
// Everything a ctx creates lazily, created NOW: the workspaces (leaf tickets, info, scalars, trtri tiles for nb_hint columns) and
// — by launching one empty kernel on each stream — the hardware queues behind its streams.  HIP creates the HSA queue of a stream
// at its first use, and every queue creation makes the hardware scheduler unmap and remap ALL queues of the process (running waves
// are context-switched out and back in).  The multi-device driver primes every rank context when it is created, so that no queue
// appears while kernels of other ranks run (the first-fit item).
int32_t ctx_prime(gp_ctx* c, long nb_hint) {
    HIPCHK(hipSetDevice(c->device));
    if (!c->info_dev) HIPCHK(hipMalloc((void**)&c->info_dev, sizeof(int)));
    RC(ctx_scal(c, 8 + 128));
    if (!c->ticket_dev) {
        HIPCHK(hipMalloc((void**)&c->ticket_dev, sizeof(int) * 64));
        HIPCHK(hipMemset(c->ticket_dev, 0, sizeof(int) * 64));
    }
    const size_t need = sizeof(double) * (size_t)(std::max(nb_hint, 256L) / 64 + 2) * 4096;
    if (c->w_ws_bytes < need) {
        if (c->w_ws) (void)hipFree(c->w_ws);
        c->w_ws_bytes = 0;
        HIPCHK(hipMalloc(&c->w_ws, need));
        HIPCHK(hipMemset(c->w_ws, 0, need));
        c->w_ws_bytes = need;
    }
    hipLaunchKernelGGL(prime_kernel, dim3(1), dim3(64), 0, c->sm, c->info_dev);
    hipLaunchKernelGGL(prime_kernel, dim3(1), dim3(64), 0, c->sp, (int*)nullptr);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(c->sm));
    HIPCHK(hipStreamSynchronize(c->sp));
    return 0;
}
int32_t ctx_prime_stream(gp_ctx* c, hipStream_t s) {
    HIPCHK(hipSetDevice(c->device));
    hipLaunchKernelGGL(prime_kernel, dim3(1), dim3(64), 0, s, (int*)nullptr);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(s));
    return 0;
}

// ------------------------------------------------------------------------------------------------
// launches
// ------------------------------------------------------------------------------------------------
static double lower_count(long M, long N, long row0, long col0) {
    // number of (r, c) in [row0,row0+M) × [col0,col0+N) with c <= r  (closed form)
    const long a = row0 - col0 + 1;  // count in the first row before clamping to [0, N]
    const long i1 = std::min(std::max(1 - a, 0L), M);           // rows contributing 0
    const long i2 = std::max(i1, std::min(std::max(N - a, 0L), M));  // rows from i2 on contribute N
    const double mid = (double)(i2 - i1) * (double)a + 0.5 * (double)(i1 + i2 - 1) * (double)(i2 - i1);
    return mid + (double)(M - i2) * (double)N;
}

template <typename T, typename CT = T>
static int32_t launch_gemm(gp_ctx* c, hipStream_t s, CT* C, long ldc, const T* A, long lda, const T* B, long ldb,
                           long M, long N, long K, GridMap g) {
    static_assert(std::is_same<T, CT>::value, "operands and result share one dtype (fp64 sums of fp32 products are formed by the callers)");
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    gp_ctx::GemmRec rec{};
    const bool timed = c->time_kernels != 0;
    if (timed) {
        RC(ctx_event(c, &rec.a, true));
        RC(ctx_event(c, &rec.b, true));
        double elems = (g.lower && g.P == 1 && g.Q == 1) ? lower_count(M, N, g.row0, g.col0) : (double)M * (double)N;
        if (g.nbatch > 1) elems *= g.nbatch;
        rec.flops = g.ktri == 1 ? (double)N * (double)M * (double)(2 * g.ktri_off + M + 128)
                                : (g.ktri == 2 ? (g.ktri_off > 0 ? 2.0 * (double)g.ktri_off * (double)N * (double)K + (double)N * (double)K * (double)K
                                                                 : (double)M * (double)M * (double)M / 3.0)
                                               : (g.ktri == 3 ? (double)M * (double)N * (double)(N + 128) : 2.0 * (double)K * elems));
        rec.bytes = 2.0 * sizeof(CT) * elems + sizeof(T) * (double)K * (double)(M + N);
        rec.M = M; rec.N = N; rec.K = K;
        rec.stream = (s == c->sp);
        HIPCHK(hipEventRecord(rec.a, s));
    }
    dim3 grid((unsigned)((N + 127) / 128), (unsigned)((M + 127) / 128));
    const long tm = (M + 127) / 128, tn = (N + 127) / 128;
    const bool single_lower = g.lower && g.P == 1 && g.Q == 1 && g.row0 >= g.col0;
    const long dt = single_lower ? (g.row0 - g.col0 + 127) / 128 : 0;
    g.tm = (int)tm;
    g.tn = (int)tn;
    g.dt = (int)dt;
    if (c->xcd_swizzle && tm * tn >= c->xcd_min_tiles) {  // XCD-aware 8×8 super-tile order (kernels.hpp xcd_tile)
        const long tms = (tm + 7) / 8, tns = (tn + 7) / 8, dts = (dt + 7) / 8;
        long nsuper;
        if (single_lower) {
            const long tri = std::min(tms, std::max(0L, tns - dts));
            nsuper = tri * (dts + 1) + tri * (tri - 1) / 2 + (tms - tri) * tns;
            g.compact = 3;
        } else {
            nsuper = tms * tns;
            g.compact = 2;
        }
        grid = dim3((unsigned)(round_up(nsuper, 8) * 64), 1);
    } else if (single_lower) {  // enumerate only the tiles on/below the diagonal
        const long tri = std::min(tm, std::max(0L, tn - dt));
        const long total = tri * (dt + 1) + tri * (tri - 1) / 2 + (tm - tri) * tn;
        g.compact = 1;
        grid = dim3((unsigned)total, 1);
    }
    if (g.nbatch > 1) grid.z = (unsigned)g.nbatch;
    if (!c->deterministic && K >= c->sk_min_k && (c->gemm_streamk || c->sk_scope > 0) && !g.beta0 && !g.ktri && g.nbatch <= 1 && g.P == 1 && g.Q == 1 && g.compact <= 1 &&
        (g.compact == 1 ? (long)grid.x : tm * tn) <= c->sk_max_tiles) {
        // persistent grid + stream-K tail (kernels.hpp gemm_nt_sk_kernel)
        const long nk = K / (128 / (long)sizeof(T));  // BK = 16 (f64) / 32 (f32)
        const long ntiles = g.compact == 1 ? (long)grid.x : tm * tn;
        const long Gmax = 2L * c->num_cus;
        const long G = Gmax;  // fewer tiles than workgroups: every tile is cut along k
        const long R = ntiles - (ntiles / G) * G;
        // tail shares: never fewer workgroups than tail tiles; beyond that at least 16 k-steps per share
        long G2 = std::min(G, std::max(R, R * nk / 16));
        if (R == 0) G2 = 0;
        if (c->gemm_pipe)
            hipLaunchKernelGGL((gemm_nt_sk_kernel<T, 1>), dim3((unsigned)G), dim3(256), 0, s, (T*)C, ldc, A, lda, B, ldb, (int)M, (int)N,
                               (int)K, g, ntiles, (int)G2);
        else
            hipLaunchKernelGGL((gemm_nt_sk_kernel<T, 0>), dim3((unsigned)G), dim3(256), 0, s, (T*)C, ldc, A, lda, B, ldb, (int)M, (int)N,
                               (int)K, g, ntiles, (int)G2);
    } else {
        // residency: two workgroups per CU (fp64: measured best over a whole factorisation; fp32: since the pipelined k loop, "gemm_pad_f32");
        // a dynamic-LDS request of 20 KiB on top of the 64 KiB static image pins ONE per CU.  "gemm_pad_lds" overrides both.
        const long pad = c->gemm_pad_user ? c->gemm_pad_lds : (sizeof(T) == 4 ? c->gemm_pad_f32 : 0);
        if (pad > 0 && !c->gemm_pad_set) {
            HIPCHK(hipFuncSetAttribute((const void*)gemm_nt_dma_kernel<double, double, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 32768));
            HIPCHK(hipFuncSetAttribute((const void*)gemm_nt_dma_kernel<float, float, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 32768));
            HIPCHK(hipFuncSetAttribute((const void*)gemm_nt_dma_kernel<double, double, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 32768));
            HIPCHK(hipFuncSetAttribute((const void*)gemm_nt_dma_kernel<float, float, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 32768));
            c->gemm_pad_set = true;
        }
        if (c->gemm_pipe)
            hipLaunchKernelGGL((gemm_nt_dma_kernel<T, CT, 1>), grid, dim3(256), (size_t)pad, s, C, ldc, A, lda, B, ldb, (int)M, (int)N, (int)K, g);
        else
            hipLaunchKernelGGL((gemm_nt_dma_kernel<T, CT, 0>), grid, dim3(256), (size_t)pad, s, C, ldc, A, lda, B, ldb, (int)M, (int)N, (int)K, g);
    }
    HIPCHK(hipGetLastError());
    if (timed) {
        HIPCHK(hipEventRecord(rec.b, s));
        c->gemm_recs.push_back(rec);
    }
    return 0;
}

GridMap gpmi::plain_map(int lower, long row0, long col0) {
    GridMap g;
    g.lower = lower;
    g.P = 1; g.p = 0; g.Q = 1; g.q = 0;
    g.nb = 128;
    g.row0 = row0;
    g.col0 = col0;
    g.compact = 0;
    g.tn = 0;
    g.dt = 0;
    g.tm = 0;
    g.beta0 = 0;
    g.ktri = 0;
    g.nbatch = 1;
    g.cstride = 0;
    g.astride = 0;
    g.bstride = 0;
    g.ktri_off = 0;
    return g;
}

static long split_half(long n) {  // largest multiple of 64 that is <= n/2 (>= 64)
    long h = (n / 128) * 64;
    return h < 64 ? 64 : h;
}

// one fused 64-column leaf (panel64_kernel): tile Cholesky (replicated per workgroup) + X L⁻ᵀ of all rows below, after the
// in-leaf update by the kpre column tiles to its left
template <typename T>
static int32_t launch_leaf(gp_ctx* c, hipStream_t s, T* A, long lda, long j0, long mtot, int* info_dev, long gcol0, long n_valid,
                           double* logdet_dev, int kpre) {
    const long mrows = mtot - j0 - 64;
    const unsigned nblk = (unsigned)std::max(1L, (mrows + 127) / 128);
    if (!c->ticket_dev) {
        HIPCHK(hipMalloc((void**)&c->ticket_dev, sizeof(int) * 64));
        // null-stream memsets are not ordered against the (non-blocking) ctx streams: zero it and wait
        HIPCHK(hipMemset(c->ticket_dev, 0, sizeof(int) * 64));
        HIPCHK(hipDeviceSynchronize());
    }
    int* const tk = c->ticket_dev + (s == c->sp ? 32 : 0);
    if constexpr (std::is_same<T, double>::value) {
        if (c->leaf_v2) {  // register-resident leaf (leaf.hip: panel64v2_kernel)
            HIPCHK((hipError_t)launch_leaf_v2(s, (double*)(A + j0 * lda + j0), lda, mrows, info_dev, (int)(gcol0 + j0), (int)n_valid, logdet_dev, tk, kpre,
                                              c->leaf_xr, c->num_cus, 64));
            return 0;
        }
    }
    hipLaunchKernelGGL(panel64_kernel<T>, dim3(nblk), dim3(256), 0, s, A + j0 * lda + j0, lda, (int)mrows, info_dev,
                       (int)(gcol0 + j0), (int)n_valid, logdet_dev, tk, kpre);
    HIPCHK(hipGetLastError());
    return 0;
}

// Factor columns [j0, j0+n) of the row-major matrix A (n multiple of 64) including all rows below
// (rows [j0, mtot)).  Recursive: left half, MFMA update of the right half, right half.
template <typename T>
static int32_t potrf_rec(gp_ctx* c, hipStream_t s, T* A, long lda, long j0, long n, long mtot, int* info_dev,
                         long gcol0, long n_valid, double* logdet_dev) {
    if (n <= 64) return launch_leaf<T>(c, s, A, lda, j0, mtot, info_dev, gcol0, n_valid, logdet_dev, 0);
    if constexpr (std::is_same<T, double>::value) {
        if (n == 128 && c->leaf_v2 && c->leaf_cols == 128 && c->leaf_group >= 128) {  // one 128-column register-resident leaf (leaf.hip)
            if (!c->ticket_dev) {
                HIPCHK(hipMalloc((void**)&c->ticket_dev, sizeof(int) * 64));
                HIPCHK(hipMemset(c->ticket_dev, 0, sizeof(int) * 64));
                HIPCHK(hipDeviceSynchronize());
            }
            int* const tk = c->ticket_dev + (s == c->sp ? 32 : 0);
            HIPCHK((hipError_t)launch_leaf_v2(s, (double*)(A + j0 * lda + j0), lda, mtot - j0 - 128, info_dev, (int)(gcol0 + j0), (int)n_valid, logdet_dev, tk, 0,
                                              c->leaf_xr, c->num_cus, 128));
            return 0;
        }
    }
    if (n <= c->leaf_group) {  // left-looking group: leaf t first applies the t tiles to its left itself
        for (long t = 0; t < n / 64; ++t)
            RC(launch_leaf<T>(c, s, A, lda, j0 + 64 * t, mtot, info_dev, gcol0, n_valid, logdet_dev, (int)t));
        return 0;
    }
    const long h = split_half(n);
    RC(potrf_rec<T>(c, s, A, lda, j0, h, mtot, info_dev, gcol0, n_valid, logdet_dev));
    bool skinny = false;
    if constexpr (std::is_same<T, double>::value) {
        const long mrows = mtot - j0 - h;
        const bool k128 = h == 128 && n - h == 128 && c->upd128;
        const bool kmid = h >= 256 && h <= c->updk_max_k && h % 32 == 0 && (n - h) % 128 == 0 && n - h <= h && (h <= c->updk_tall_k || mrows <= c->updk_tall_m);
        if (c->leaf_v2 && (k128 || kmid)) {  // the skinny in-panel updates through the register chain (leaf.hip panel_updk_kernel)
            HIPCHK((hipError_t)launch_panel_updk(s, (double*)(A + (j0 + h) * lda + (j0 + h)), lda, (const double*)(A + (j0 + h) * lda + j0), lda, mrows, n - h, h,
                                                 c->updk_rt, c->num_cus));
            skinny = true;
        }
    }
    if (!skinny)
        RC(launch_gemm<T>(c, s, A + (j0 + h) * lda + (j0 + h), lda, A + (j0 + h) * lda + j0, lda,
                          A + (j0 + h) * lda + j0, lda, mtot - j0 - h, n - h, h, plain_map(1, j0 + h, j0 + h)));
    RC(potrf_rec<T>(c, s, A, lda, j0 + h, n - h, mtot, info_dev, gcol0, n_valid, logdet_dev));
    return 0;
}

// W_j = I − inv(L_jj) for every 64×64 diagonal tile of a lower factor (one batched launch; the 64-wide steps of the vector solves)
template <typename T> static int32_t trtri_tiles(gp_ctx* c, hipStream_t s, const T* L, long ldl, long n, T** Wout) {
    const size_t need = sizeof(T) * (size_t)(n / 64 + 2) * 4096;  // + slack tiles (B-operand over-read of gemm_nt)
    if (c->w_ws_bytes < need) {
        HIPCHK(hipStreamSynchronize(c->sm));
        HIPCHK(hipStreamSynchronize(c->sp));
        if (c->w_ws) (void)hipFree(c->w_ws);
        c->w_ws_bytes = 0;
        HIPCHK(hipMalloc(&c->w_ws, need));
        HIPCHK(hipMemsetAsync(c->w_ws, 0, need, s));  // same stream as the trtri launch that follows
        c->w_ws_bytes = need;
    }
    hipLaunchKernelGGL(trtri_64_kernel<T>, dim3((unsigned)(n / 64)), dim3(64), 0, s, L, ldl, (T*)c->w_ws);
    HIPCHK(hipGetLastError());
    *Wout = (T*)c->w_ws;
    return 0;
}
// 64-wide TRSM leaf on the matrix pipe (trsm64_mfma, one workgroup per 128 rows)
template <typename T>
static int32_t launch_trsm64(gp_ctx* c, hipStream_t s, T* X, long ldx, long M, const T* L, long ldl) {
    (void)c;
    if (M <= 0) return 0;
    hipLaunchKernelGGL(trsm64_mfma_kernel<T>, dim3((unsigned)((M + 127) / 128)), dim3(256), 0, s, X, ldx, (int)M, L, ldl);
    HIPCHK(hipGetLastError());
    return 0;
}

// Inverse diagonal blocks ("DIB") of a resident factor.  A forward solve X ← X L⁻ᵀ by recursion ends in N/64 latency-bound leaf launches
// and as many few-tile GEMMs (C4, 4 096 test points: 2 047 launches; the levels below 2 048 columns hold 3 % of the flops and a third of
// the time).  With W_b = −inv(L_bb) of every diagonal block (j0, n <= nbi) at hand, the leaf of the recursion at that size is ONE
// triangular-k MFMA GEMM  S = −X_b W_bᵀ = X_b L_bb⁻ᵀ  (B operand lower triangular: GridMap::ktri = 3) into a scratch panel that is copied
// back; everything above it stays the recursion's large-K GEMMs.  The blocks are the leaves the recursion itself reaches (dib_ranges).
template <typename T> struct DibArgs {
    const T* W = nullptr;  // −inv(L_bb), rows shifted along with L (block (j0, n): rows [j0, j0 + n), columns [0, n))
    long ldw = 0, nbi = 0;
    T* S = nullptr;        // scratch panel, at least (rows + 128) × lds
    long lds = 0;
};
static void dib_ranges(long j0, long n, long nbi, std::vector<std::pair<long, long>>& out) {
    if (n <= nbi) {
        out.push_back({j0, n});
        return;
    }
    const long h = split_half(n);
    dib_ranges(j0, h, nbi, out);
    dib_ranges(j0 + h, n - h, nbi, out);
}
// S = −X W_bᵀ (M × n), then X ← S
template <typename T>
static int32_t dib_apply(gp_ctx* c, hipStream_t s, T* X, long ldx, long M, long n, const DibArgs<T>& dib) {
    if (M <= 0) return 0;
    GridMap g = plain_map(0, 0, 0);
    g.beta0 = 1;
    g.ktri = 3;
    RC(launch_gemm<T>(c, s, dib.S, dib.lds, X, ldx, dib.W, dib.ldw, M, n, n, g));
    HIPCHK(hipMemcpy2DAsync(X, sizeof(T) * ldx, dib.S, sizeof(T) * dib.lds, sizeof(T) * n, M, hipMemcpyDeviceToDevice, s));
    return 0;
}

// X[M×n] ← X · L⁻ᵀ with L the n×n row-major lower factor (n, M multiples of 64): left half; X_right −= X_left · L_21ᵀ (MFMA GEMM);
// right half; leaves: the explicit inverse block (dib, n <= nbi) or 64-wide trsm64_mfma.
template <typename T>
static int32_t trsm_rec_v(gp_ctx* c, hipStream_t s, T* X, long ldx, long M, const T* L, long ldl, long n, DibArgs<T> dib = DibArgs<T>()) {
    if (dib.W && n <= dib.nbi) return dib_apply<T>(c, s, X, ldx, M, n, dib);
    if (n <= 64) return launch_trsm64<T>(c, s, X, ldx, M, L, ldl);
    const long h = split_half(n);
    RC(trsm_rec_v<T>(c, s, X, ldx, M, L, ldl, h, dib));
    RC(launch_gemm<T>(c, s, X + h, ldx, X, ldx, L + h * ldl, ldl, M, n - h, h, plain_map(0, 0, 0)));
    if (dib.W) dib.W += h * dib.ldw;
    RC(trsm_rec_v<T>(c, s, X + h, ldx, M, L + h * ldl + h, ldl, n - h, dib));
    return 0;
}
// W ← W L⁻ᵀ for W that is upper triangular on entry AND exit (W = I gives L⁻ᵀ): columns [j0, j0+n) only ever have
// non-zeros in rows [0, j0+n), so every step is restricted to those rows — N³/3 flops instead of the N³ of the general
// solve.  Same recursion as trsm_rec_v.  With dib (W indexed by GLOBAL row here): the diagonal blocks of X already hold L_bb⁻ᵀ (written
// by dib_build), a leaf only multiplies the rows above its block.
template <typename T>
static int32_t trsm_upper_rec(gp_ctx* c, hipStream_t s, T* X, long ldx, const T* L, long ldl, long j0, long n, DibArgs<T> dib = DibArgs<T>()) {
    if (dib.W && n <= dib.nbi) {
        DibArgs<T> d2 = dib;
        d2.W = dib.W + j0 * dib.ldw;
        return dib_apply<T>(c, s, X + j0, ldx, j0, n, d2);
    }
    if (n <= 64) return launch_trsm64<T>(c, s, X + j0, ldx, j0 + 64, L + j0 * ldl + j0, ldl);
    const long h = split_half(n);
    RC(trsm_upper_rec<T>(c, s, X, ldx, L, ldl, j0, h, dib));
    {
        // A = X[0 : j0+h, j0 : j0+h]: rows below j0 hold the h×h UPPER-triangular block just solved — its leading zeros are skipped on launches
        // large enough for the hardware-dispatched kernel (until round 5 this product ran over the whole k range: N³/2 flops for L⁻ᵀ instead
        // of N³/3; the few-tile launches keep the full range and their stream-K cut)
        GridMap g = plain_map(0, 0, 0);
        if (h >= 512 && ((j0 + h) / 128) * ((n - h + 127) / 128) > c->sk_max_tiles / 8) {
            g.ktri = 2;
            g.ktri_off = (int)j0;
        }
        RC(launch_gemm<T>(c, s, X + j0 + h, ldx, X + j0, ldx, L + (j0 + h) * ldl + j0, ldl, j0 + h, n - h, h, g));
    }
    RC(trsm_upper_rec<T>(c, s, X, ldx, L, ldl, j0 + h, n - h, dib));
    return 0;
}

template <typename T>
static int32_t trsm_rec(gp_ctx* c, hipStream_t s, T* X, long ldx, long M, const T* L, long ldl, long n, DibArgs<T> dib = DibArgs<T>()) {
    if (M <= 0) return 0;
    return trsm_rec_v<T>(c, s, X, ldx, M, L, ldl, n, dib);
}

// W (rows [0, np + 128) × ldw, zero on entry is NOT required) ← −inv(L_bb) for every block of dib_ranges(0, np, nbi); Iw: scratch of the same
// shape.  Per block: Iw_b = I, Iw_b ← Iw_b L_bb⁻ᵀ (upper; the restricted-row recursion above, without dib), W_b = −Iw_bᵀ.  up != nullptr:
// the upper blocks L_bb⁻ᵀ are also written onto the diagonal of `up` (the gradient's L⁻ᵀ).
template <typename T>
static int32_t dib_build(gp_ctx* c, hipStream_t s, const T* L, long ldl, long np, long nbi, T* W, long ldw, T* Iw, T* up, long ldu) {
    std::vector<std::pair<long, long>> blocks;
    dib_ranges(0, np, nbi, blocks);
    HIPCHK(hipMemsetAsync(Iw, 0, sizeof(T) * (size_t)(np + 128) * ldw, s));
    HIPCHK(hipMemsetAsync(W, 0, sizeof(T) * (size_t)(np + 128) * ldw, s));
    bool uniform = blocks.size() > 1;
    for (const auto& b : blocks) uniform = uniform && b.second == blocks[0].second;
    if (uniform) {  // equal blocks (np a power-of-two multiple of the block): the whole batch per launch
        const long n = blocks[0].second, nb = (long)blocks.size(), xs = n * ldw, ls = n * ldl + n;
        hipLaunchKernelGGL(diag_ones_kernel<T>, dim3((unsigned)((n + 255) / 256), (unsigned)nb), dim3(256), 0, s, Iw, ldw, n, xs);
        HIPCHK(hipGetLastError());
        RC(trsm_upper_rec_batched<T>(c, s, Iw, ldw, L, ldl, 0, n, nb, xs, ls));
        hipLaunchKernelGGL(transpose_scale_kernel<T>, dim3((unsigned)((n + 31) / 32), (unsigned)((n + 31) / 32), (unsigned)nb), dim3(256), 0, s, (const T*)Iw, ldw,
                           W, ldw, n, T(-1), xs, xs);
        HIPCHK(hipGetLastError());
        if (up)
            for (const auto& b : blocks)
                HIPCHK(hipMemcpy2DAsync(up + b.first * ldu + b.first, sizeof(T) * ldu, Iw + b.first * ldw, sizeof(T) * ldw, sizeof(T) * n, n, hipMemcpyDeviceToDevice, s));
        return 0;
    }
    for (const auto& b : blocks) {
        const long j0 = b.first, n = b.second;
        T* Ib = Iw + j0 * ldw;
        hipLaunchKernelGGL(identity_kernel<T>, dim3((unsigned)((n + 255) / 256), (unsigned)n), dim3(256), 0, s, Ib, ldw, n);
        HIPCHK(hipGetLastError());
        RC(trsm_upper_rec<T>(c, s, Ib, ldw, L + j0 * ldl + j0, ldl, 0, n));
        hipLaunchKernelGGL(transpose_scale_kernel<T>, dim3((unsigned)((n + 31) / 32), (unsigned)((n + 31) / 32)), dim3(256), 0, s, (const T*)Ib, ldw,
                           W + j0 * ldw, ldw, n, T(-1));
        HIPCHK(hipGetLastError());
        if (up) HIPCHK(hipMemcpy2DAsync(up + j0 * ldu + j0, sizeof(T) * ldu, Ib, sizeof(T) * ldw, sizeof(T) * n, n, hipMemcpyDeviceToDevice, s));
    }
    return 0;
}

// The restricted-row recursion of trsm_upper_rec for `nb` equal blocks at once (block b: X + b·xs against L + b·ls): every launch of the recursion
// carries the whole batch (trsm64 leaves: grid.y; GEMMs: grid.z with independent operand strides), so the inverse blocks of a factor cost the ≈ 63
// launches of ONE 2 048-column block instead of N/2 048 times that (C4: 32 × 0.6 ms of serial chains -> ≈ 1 ms).
template <typename T>
static int32_t trsm_upper_rec_batched(gp_ctx* c, hipStream_t s, T* X, long ldx, const T* L, long ldl, long j0, long n, long nb, long xs, long ls) {
    if (n <= 64) {
        hipLaunchKernelGGL(trsm64_mfma_kernel<T>, dim3((unsigned)((j0 + 64 + 127) / 128), (unsigned)nb), dim3(256), 0, s, X + j0, ldx, (int)(j0 + 64),
                           L + j0 * ldl + j0, ldl, xs, ls);
        HIPCHK(hipGetLastError());
        return 0;
    }
    const long h = split_half(n);
    RC(trsm_upper_rec_batched<T>(c, s, X, ldx, L, ldl, j0, h, nb, xs, ls));
    GridMap g = plain_map(0, 0, 0);
    g.nbatch = (int)nb;
    g.cstride = xs;
    g.astride = xs;
    g.bstride = ls;
    RC(launch_gemm<T>(c, s, X + j0 + h, ldx, X + j0, ldx, L + (j0 + h) * ldl + j0, ldl, j0 + h, n - h, h, g));
    RC(trsm_upper_rec_batched<T>(c, s, X, ldx, L, ldl, j0 + h, n - h, nb, xs, ls));
    return 0;
}

// The forward solve X ← X L⁻ᵀ against a RESIDENT factor L (np × np, nvalid real rows) with the inverse diagonal blocks kept in `cache` beside it: built on
// first use (kept until the owner of the factor is freed), then the recursion runs with them.  X: M × np (+ 128 slack rows).
template <typename T>
static int32_t trsm_cached(gp_ctx* c, hipStream_t s, T* X, long ldx, long M, const T* A, long ld, long np, long nvalid, DibCache& cache, DevBufs& bufs) {
    if (M <= 0) return 0;
    if (c->dib_nb < 128 || np < c->dib_nb || cache.nbi < 0) return trsm_rec_v<T>(c, s, X, ldx, M, A, ld, np);
    const long nbi = round_up(c->dib_nb, 128), ldw = nbi + c->ldpad;
    // a handful of rows never repays the build (≈ np·nbi elements allocated, zeroed and inverted: a sequential-conditioning loop with tiny batches would pay
    // it for every new handle): blocks that exist are used, new ones are built from 128 rows on
    if (!cache.w && M < 128) return trsm_rec_v<T>(c, s, X, ldx, M, A, ld, np);
    if (!cache.w) {
        // guard: a product with an explicit inverse carries an error of order cond(L_bb)·ε where substitution is backward stable.  max / min of the
        // factor's diagonal bounds cond(L) from below; beyond the limit — 1e5 in fp64 (cond(K + Σy) >= 1e10: interpolation-style fits with vanishing
        // noise), 300 ≈ √(1/ε)/10 in fp32 — this factor keeps the recursion with substitution leaves for good (nbi = −1)
        const double limit = sizeof(T) == 8 ? 1e5 : 300.0;
        RC(ctx_scal(c, 16));
        hipLaunchKernelGGL(diag_minmax_kernel<T>, dim3(1), dim3(1024), 0, s, A, ld, nvalid, c->scal_dev + 6);
        HIPCHK(hipGetLastError());
        double mm[2] = {1.0, 1.0};
        HIPCHK(hipMemcpyAsync(mm, c->scal_dev + 6, sizeof(mm), hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        if (!(mm[0] > 0.0) || mm[1] / mm[0] > limit) {
            cache.nbi = -1;
            return trsm_rec_v<T>(c, s, X, ldx, M, A, ld, np);
        }
    }
    const size_t wb = sizeof(T) * (size_t)(np + 128) * ldw;
    if (!cache.w || cache.nbi != nbi) {
        if (cache.w) ctx_release(c, cache.w, cache.bytes);
        cache.w = nullptr;
        void* w = nullptr;
        void* iw = nullptr;
        // no memory for the blocks (W) or the build's workspace (Iw): the solve itself needs neither — substitution leaves, this call only
        if (ctx_alloc(c, wb, &w) != 0) return trsm_rec_v<T>(c, s, X, ldx, M, A, ld, np);
        if (bufs.get(wb, &iw) != 0) {
            ctx_release(c, w, wb);
            return trsm_rec_v<T>(c, s, X, ldx, M, A, ld, np);
        }
        int32_t rc = dib_build<T>(c, s, A, ld, np, nbi, (T*)w, ldw, (T*)iw, (T*)nullptr, 0);
        if (rc != 0) {
            (void)hipStreamSynchronize(s);
            ctx_release(c, w, wb);
            return rc;
        }
        cache.w = w; cache.bytes = wb; cache.nbi = nbi; cache.ldw = ldw;
    }
    // S: ONE panel per API call (DevBufs::scratch), shared by the chunks of a large prediction — allocated per call of this function it grew with the
    // number of test points (70 MB per 4 096-row chunk: 17 GB beside the factor for 10⁶ points)
    void* S_v = nullptr;
    const long lds = nbi + c->ldpad;
    if (bufs.scratch(sizeof(T) * (size_t)(M + 128) * lds, &S_v) != 0) return trsm_rec_v<T>(c, s, X, ldx, M, A, ld, np);
    DibArgs<T> dib;
    dib.W = (const T*)cache.w; dib.ldw = ldw; dib.nbi = nbi; dib.S = (T*)S_v; dib.lds = lds;
    return trsm_rec_v<T>(c, s, X, ldx, M, A, ld, np, dib);
}
template <typename T>
static int32_t trsm_post(gp_post* post, hipStream_t s, T* X, long ldx, long M, DevBufs& bufs) {
    return trsm_cached<T>(post->ctx, s, X, ldx, M, (const T*)post->A, post->ld, post->np, post->n, post->dibc, bufs);
}

// Full factorisation of the np×np matrix (rows [np, mtot) are carried RHS rows): right-looking over panels of width nb with a
// one-panel look-ahead (potrf_full_la below); nb = 0: the plain recursion on one stream.
template <typename T>
static int32_t potrf_full_la(gp_ctx* c, T* A, long lda, long np, long mtot, int* info_dev, long n_valid, double* logdet_dev);

template <typename T>
static int32_t potrf_full(gp_ctx* c, T* A, long lda, long np, long mtot, int* info_dev, long n_valid,
                          double* logdet_dev) {
    if (c->nb == 0) return potrf_rec<T>(c, c->sm, A, lda, 0, np, mtot, info_dev, 0, n_valid, logdet_dev);
    return potrf_full_la<T>(c, A, lda, np, mtot, info_dev, n_valid, logdet_dev);
}

// sched 0: right-looking over panels of width nb with a one-panel look-ahead: the whole next panel (recursive
// Cholesky of its columns including every row below) is factored on the panel stream while the rest of the
// trailing update still runs on the main stream.  (A variant with the two streams on disjoint CU sets was built in round 3
// and measured slower at every size in rounds 3 and 4 — profiles/r3/sweep_cusplit.jsonl, profiles/r4/nb_sweep.jsonl; it lives
// in the history at f1ed70e.)
template <typename T>
static int32_t potrf_full_la(gp_ctx* c, T* A, long lda, long np, long mtot, int* info_dev, long n_valid,
                             double* logdet_dev) {
    long nb = c->nb;
    // below the look-ahead threshold the schedule is one stream anyway: panels of "nb_small" (4 096) columns halve the passes over the trailing matrix
    // (K = 4 096 updates) — C2 29.8-30.1 -> 29.0-29.2 ms on two boxes, N = 8 192 6.51 -> 6.43 (profiles/r5/nb_sweep.txt); from the threshold on the
    // widths measure within ± 0.5 % of each other ("nb_large" = 2 048).  An explicit "nb" >= 0 applies to every size.
    if (nb < 0) nb = np < c->lookahead_min_n ? c->nb_small : c->nb_large;  // "nb" = −1: automatic
    if (nb <= 0 || nb >= np) return potrf_rec<T>(c, c->sm, A, lda, 0, np, mtot, info_dev, 0, n_valid, logdet_dev);
    nb = round_up(nb, 128);
    const bool la = c->lookahead != 0 && np >= c->lookahead_min_n;
    hipStream_t sM = c->sm, sP = la ? c->sp : c->sm;
    hipEvent_t ev_u1 = nullptr, ev_panel = nullptr;
    if (la) {  // the panel stream starts after everything queued so far on the main stream (assembly)
        RC(ctx_event(c, &ev_u1, false));
        HIPCHK(hipEventRecord(ev_u1, c->sm));
        HIPCHK(hipStreamWaitEvent(sP, ev_u1, 0));
    }
    for (long k = 0; k < np; k += nb) {
        const long nbk = std::min(nb, np - k);
        RC(potrf_rec<T>(c, sP, A, lda, k, nbk, mtot, info_dev, 0, n_valid, logdet_dev));
        const long k1 = k + nbk;
        if (la) {
            RC(ctx_event(c, &ev_panel, false));
            HIPCHK(hipEventRecord(ev_panel, sP));
            HIPCHK(hipStreamWaitEvent(sM, ev_panel, 0));
        }
        if (k1 >= np) break;  // RHS rows were already solved inside potrf_rec
        const long nb1 = std::min(nb, np - k1);
        // U1: next panel's columns, all rows below
        RC(launch_gemm<T>(c, sM, A + k1 * lda + k1, lda, A + k1 * lda + k, lda, A + k1 * lda + k, lda, mtot - k1, nb1, nbk, plain_map(1, k1, k1)));
        if (la) {
            RC(ctx_event(c, &ev_u1, false));
            HIPCHK(hipEventRecord(ev_u1, sM));
            HIPCHK(hipStreamWaitEvent(sP, ev_u1, 0));
        }
        // U2: the rest of the trailing matrix
        const long k2 = k1 + nb1;
        if (k2 < np)
            RC(launch_gemm<T>(c, sM, A + k2 * lda + k2, lda, A + k2 * lda + k, lda, A + k2 * lda + k, lda,
                              mtot - k2, np - k2, nbk, plain_map(1, k2, k2)));
    }
    return 0;
}

// Vector solves with the resident factor: R rows hold nrhs right-hand sides of length np.
// Blocks of trsv_nb (256) columns: a one-workgroup diagonal solve (trsv_diag2: the memory round trips of every 64-wide step are in
// flight before they are needed) and a many-workgroup, HBM-bound update of the remaining vector.
template <typename T>
static int32_t trsv(gp_ctx* c, hipStream_t s, const T* L, long ldl, long np, T* R, long ldr, int nrhs, bool fwd) {
    const int NBV = (int)c->trsv_nb;
    const bool k2 = NBV <= 256;
    const size_t smem = k2 ? sizeof(T) * (256 + 64 * 65 + 16 * 64 + 4 * 256) : sizeof(T) * (NBV + 64 * 65 + 16 * 64);
    const long nblk = (np + NBV - 1) / NBV;
    T* W = nullptr;
    RC(trtri_tiles<T>(c, s, L, ldl, np, &W));  // I − inv(L_jj) for every 64×64 diagonal tile, one batched launch
    for (long bb = 0; bb < nblk; ++bb) {
        const long b = fwd ? bb : (nblk - 1 - bb);
        const long b0 = b * NBV;
        const int nbv = (int)std::min<long>(NBV, np - b0);  // multiple of 64 (np is a multiple of 128)
        if (fwd) {
            if (k2)
                hipLaunchKernelGGL((trsv_diag2_kernel<T, true>), dim3(1), dim3(1024), smem, s, L, ldl, b0, nbv, R, ldr, nrhs, (const T*)W);
            else
                hipLaunchKernelGGL((trsv_diag_kernel<T, true>), dim3(1), dim3(1024), smem, s, L, ldl, b0, nbv, R, ldr, nrhs, (const T*)W);
            HIPCHK(hipGetLastError());
            const long lo = b0 + nbv;
            if (lo < np) {
                hipLaunchKernelGGL(trsv_upd_fwd_kernel<T>, dim3((unsigned)((np - lo + 3) / 4)), dim3(256), 0, s, L,
                                   ldl, b0, nbv, lo, np, R, ldr, nrhs);
                HIPCHK(hipGetLastError());
            }
        } else {
            if (k2)
                hipLaunchKernelGGL((trsv_diag2_kernel<T, false>), dim3(1), dim3(1024), smem, s, L, ldl, b0, nbv, R, ldr, nrhs, (const T*)W);
            else
                hipLaunchKernelGGL((trsv_diag_kernel<T, false>), dim3(1), dim3(1024), smem, s, L, ldl, b0, nbv, R, ldr, nrhs, (const T*)W);
            HIPCHK(hipGetLastError());
            if (b0 > 0) {
                hipLaunchKernelGGL(trsv_upd_bwd_kernel<T>, dim3((unsigned)((b0 + 255) / 256), c->deterministic ? 1u : (unsigned)(nbv / 64)),
                                   dim3(256), 0, s, L, ldl, b0, nbv, R, ldr, nrhs);
                HIPCHK(hipGetLastError());
            }
        }
    }
    return 0;
}

// ------------------------------------------------------------------------------------------------
// engine entry points for multi.hip (declared in engine.hpp)
// ------------------------------------------------------------------------------------------------
int32_t gpmi::eng_assemble(gp_ctx* c, hipStream_t s, int kind, double variance, const double* x_dev, long n_valid, long n_pad, int d,
                           const double* noise_dev, GridMap g, double* a_loc, long lda, long m_loc, long n_loc) {
    (void)c;
    dim3 grid((unsigned)(n_loc / 128), (unsigned)(m_loc / 128));
    if (grid.x == 0 || grid.y == 0) return 0;
    launch_kmat<double>(grid, s, a_loc, lda, x_dev, n_pad, x_dev, n_pad, d, kind, variance,
                       noise_dev, n_valid, n_valid, 1, g, (const double*)nullptr, (const double*)nullptr);
    HIPCHK(hipGetLastError());
    return 0;
}
int32_t gpmi::eng_potrf(gp_ctx* c, hipStream_t s, double* a, long lda, long m, long n, int* info_dev, long col0, long n_valid,
                        double* logdet_dev) {
    return potrf_rec<double>(c, s, a, lda, 0, n, m, info_dev, col0, n_valid, logdet_dev);
}
int32_t gpmi::eng_trsm(gp_ctx* c, hipStream_t s, double* x, long ldx, long m, const double* l, long ldl, long n) {
    return trsm_rec_v<double>(c, s, x, ldx, m, l, ldl, n);
}
// W ← −inv(L) for ONE nb×nb lower block (the multi-device driver's diagonal block).  W, iw (and v): nb × ldw; the caller keeps 128 finite slack rows below each
// (the operand over-read of the GEMM) and W zero above its diagonal.
//   nb = 64·2^m and a second scratch v given — LEVEL-WISE (round 6): −inv and its transpose of every 64×64 diagonal tile in one launch (trtri_64_neg), then per level
//   s = 64, 128, …, nb/2 and for ALL pairs of the level at once (batched launches): with An = −inv(L11), Cn = −inv(L22) of the level below (lower, in W) and their
//   transposes (upper, in iw),  V = −AnT·L21ᵀ (= (L21·inv(L11))ᵀ),  W21 = −Cn·Vᵀ (= inv(L22)·L21·inv(L11) = −(−inv(L))21),  WT12 = −V·Cnᵀ (its transpose):
//   1 + 3·log2(nb/64) launches (13 at nb = 1 024) where the restricted-row recursion on the identity takes nb/64 substitution leaves and as many few-tile GEMMs (33).
//   iw's lower-left and W's upper-right triangles are never written (the caller zeroed them once).
//   otherwise: Iw = I, Iw ← Iw L⁻ᵀ (upper; the restricted-row recursion), W = −Iwᵀ.
int32_t gpmi::eng_inv_lower(gp_ctx* c, hipStream_t s, const double* l, long ldl, long nb, double* w, long ldw, double* iw, double* v) {
    const long t64 = nb / 64;
    if (v && nb >= 128 && nb % 64 == 0 && (t64 & (t64 - 1)) == 0) {
        hipLaunchKernelGGL(trtri_64_neg_kernel<double>, dim3((unsigned)t64), dim3(64), 0, s, l, ldl, w, iw, ldw);
        HIPCHK(hipGetLastError());
        for (long sz = 64; sz < nb; sz *= 2) {
            GridMap g = plain_map(0, 0, 0);
            g.beta0 = 1;
            g.nbatch = (int)(nb / (2 * sz));
            const long sw = 2 * sz * ldw + 2 * sz, sl = 2 * sz * ldl + 2 * sz;  // pair p: diagonal position 2·p·sz in W / iw / v, and in l
            g.cstride = sw;
            g.astride = sw;
            g.bstride = sl;
            RC(launch_gemm<double>(c, s, v, ldw, iw, ldw, l + sz * ldl, ldl, sz, sz, sz, g));                        // V = −AnT · L21ᵀ
            g.bstride = sw;
            RC(launch_gemm<double>(c, s, w + sz * ldw, ldw, w + sz * ldw + sz, ldw, v, ldw, sz, sz, sz, g));         // W21 = −Cn · Vᵀ
            RC(launch_gemm<double>(c, s, iw + sz, ldw, v, ldw, w + sz * ldw + sz, ldw, sz, sz, sz, g));              // WT12 = −V · Cnᵀ
        }
        return 0;
    }
    hipLaunchKernelGGL(identity_kernel<double>, dim3((unsigned)((nb + 255) / 256), (unsigned)nb), dim3(256), 0, s, iw, ldw, nb);
    HIPCHK(hipGetLastError());
    RC(trsm_upper_rec<double>(c, s, iw, ldw, l, ldl, 0, nb));
    hipLaunchKernelGGL(transpose_scale_kernel<double>, dim3((unsigned)((nb + 31) / 32), (unsigned)((nb + 31) / 32)), dim3(256), 0, s, (const double*)iw, ldw, w, ldw,
                       nb, -1.0);
    HIPCHK(hipGetLastError());
    return 0;
}
// X (m × nb) ← X L⁻ᵀ as ONE triangular-k MFMA GEMM with w = −inv(L): S = −X wᵀ into the scratch sc ((m + 128) × lds), copied back over X
int32_t gpmi::eng_trsm_inv(gp_ctx* c, hipStream_t s, double* x, long ldx, long m, const double* w, long ldw, long nb, double* sc, long lds) {
    DibArgs<double> dib;
    dib.W = w; dib.ldw = ldw; dib.nbi = nb; dib.S = sc; dib.lds = lds;
    return dib_apply<double>(c, s, x, ldx, m, nb, dib);
}
int32_t gpmi::eng_gemm_nt(gp_ctx* c, hipStream_t s, double* cm, long ldc, const double* a, long lda, const double* b, long ldb,
                          long m, long n, long k, GridMap g) {
    return launch_gemm<double>(c, s, cm, ldc, a, lda, b, ldb, m, n, k, g);
}
int32_t gpmi::eng_trsv(gp_ctx* c, hipStream_t s, const double* l, long ldl, long np, double* r, long ldr, int nrhs, bool forward) {
    return trsv<double>(c, s, l, ldl, np, r, ldr, nrhs, forward);
}
int32_t gpmi::eng_gemv_t(gp_ctx* c, hipStream_t s, const double* l, long ldl, long nrows, long ncols, const double* a, double* r) {
    (void)c;
    if (nrows <= 0 || ncols <= 0) return 0;
    hipLaunchKernelGGL(gemv_t_kernel<double>, dim3((unsigned)((ncols + 255) / 256), (unsigned)((nrows + 63) / 64)), dim3(256), 0, s, l,
                       ldl, nrows, ncols, a, r);
    HIPCHK(hipGetLastError());
    return 0;
}
int32_t gpmi::eng_rowsumsq(gp_ctx* c, hipStream_t s, const double* x, long ldx, long nrows, long ncols, double* out_dev) {
    (void)c;
    if (nrows <= 0) return 0;
    hipLaunchKernelGGL(rowsumsq_kernel<double>, dim3((unsigned)nrows), dim3(256), 0, s, x, ldx, ncols, out_dev);
    HIPCHK(hipGetLastError());
    return 0;
}
int32_t gpmi::eng_copy2d(gp_ctx* c, hipStream_t s, double* dst, long dld, const double* src, long sld, long rows, long cols) {
    (void)c;
    if (rows <= 0 || cols <= 0) return 0;
    hipLaunchKernelGGL(copy2d_kernel, dim3((unsigned)std::min<long>(rows, 1024)), dim3(256), 0, s, dst, dld, src, sld, rows, cols);
    HIPCHK(hipGetLastError());
    return 0;
}
int32_t gpmi::eng_kcross(gp_ctx* c, hipStream_t s, int kind, double variance, const double* xr, long ldxr, long nr_valid, long nr_pad,
                         const double* xc, long ldxc, long nc_valid, long nc_pad, int d, double* out, long ld) {
    (void)c;
    dim3 grid((unsigned)(nc_pad / 128), (unsigned)(nr_pad / 128));
    if (grid.x == 0 || grid.y == 0) return 0;
    launch_kmat<double>(grid, s, out, ld, xr, ldxr, xc, ldxc, d, kind, variance, (const double*)nullptr, nr_valid,
                       nc_valid, 0, plain_map(0, 0, 0), (const double*)nullptr, (const double*)nullptr);
    HIPCHK(hipGetLastError());
    return 0;
}
int32_t gpmi::eng_kvec(gp_ctx* c, hipStream_t s, const double* xs, long ldxs, const double* x, long ldx, int d, int kind, double variance,
                       long n, const double* alpha, double* out, long nrows) {
    (void)c;
    if (nrows <= 0) return 0;
    hipLaunchKernelGGL(kvec_kernel<double>, dim3((unsigned)nrows), dim3(256), 0, s, xs, ldxs, x, ldx, d, kind, variance, n, alpha, out);
    HIPCHK(hipGetLastError());
    return 0;
}
int32_t gpmi::eng_add_vec(gp_ctx* c, hipStream_t s, double* dst, const double* src, long n) {
    (void)c;
    if (n <= 0) return 0;
    hipLaunchKernelGGL(axpy_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, dst, src, n);
    HIPCHK(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------------
// host-side input marshalling
// ------------------------------------------------------------------------------------------------
template <typename T> static T pt_get(const gp_points* x, long i, int dd) {
    const T* p = (const T*)x->data;
    if (x->layout == 0) return p[i];
    if (x->layout == 1) return p[(long)dd + i * x->d];
    return p[i + (long)dd * x->n];
}
static int32_t check_kernel(const gp_kernel* k, int d, int argi) {
    if (!k) return set_arg_err(argi, "kernel is NULL");
    if (k->kind < 0 || k->kind > 3) return set_arg_err(argi, "kernel kind must be 0..3");
    if (k->dtype != 0 && k->dtype != 1) return set_arg_err(argi, "dtype must be 0 (f64) or 1 (f32)");
    if (!(k->variance > 0)) return set_arg_err(argi, "variance must be > 0");
    if (k->nscale != 0 && k->nscale != 1 && k->nscale != d) return set_arg_err(argi, "nscale must be 0, 1 or D");
    if (k->nscale != 0 && !k->scale) return set_arg_err(argi, "scale is NULL");
    return 0;
}
static int32_t check_points(const gp_points* x, int argi) {
    if (!x || !x->data) return set_arg_err(argi, "points NULL");
    if (x->n <= 0) return set_arg_err(argi, "n must be > 0");
    if (x->d <= 0) return set_arg_err(argi, "d must be > 0");
    if (x->layout < 0 || x->layout > 2) return set_arg_err(argi, "layout must be 0..2");
    if (x->layout == 0 && x->d != 1) return set_arg_err(argi, "layout 0 requires d == 1");
    return 0;
}
// scaled, dimension-major, zero-padded copy [d][ldx]
template <typename T>
static void scale_points_into(const gp_kernel* k, const gp_points* x, long ldx, T* out);
template <typename T>
static void scale_points(const gp_kernel* k, const gp_points* x, long ldx, std::vector<T>& out) {
    out.resize((size_t)x->d * ldx);
    scale_points_into<T>(k, x, ldx, out.data());
}
template <typename T>
static void scale_points_into(const gp_kernel* k, const gp_points* x, long ldx, T* out) {  // out: [d][ldx], every element written (padding zero)
    const int d = x->d;
    for (int dd = 0; dd < d; ++dd) {
        T s = T(1);
        if (k->nscale == 1) s = (T)k->scale[0];
        else if (k->nscale > 1) s = (T)k->scale[dd];
        T* o = out + (size_t)dd * ldx;
        const T* p = (const T*)x->data;
        const long n = x->n;
        std::fill(o + n, o + ldx, T(0));
        if (x->layout == 1) {  // point-contiguous: stride d
            for (long i = 0; i < n; ++i) o[i] = s * p[(long)dd + i * d];
        } else {               // a vector, or dimension-contiguous: unit stride
            const T* q = x->layout == 0 ? p : p + (long)dd * n;
            for (long i = 0; i < n; ++i) o[i] = s * q[i];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// posterior handle
// ------------------------------------------------------------------------------------------------

template <typename T> static int32_t assemble_sym(gp_ctx* c, const gp_kernel* k, const T* xs_dev, long ldx, int d,
                                                  const T* noise_dev, long n, long np, T* A, long ld) {
    GridMap g = plain_map(1, 0, 0);
    dim3 grid((unsigned)(np / 128), (unsigned)(np / 128));
    launch_kmat<T>(grid, c->sm, A, ld, xs_dev, ldx, xs_dev, ldx, d, k->kind,
                       (T)k->variance, noise_dev, n, n, 1, g, (const T*)nullptr, (const T*)nullptr);
    HIPCHK(hipGetLastError());
    return 0;
}

struct FitOut {
    std::vector<double> logpdf;  // per RHS column
    double logdet = 0;           // logdet(K + Σy)            (src/finite_gp_projection.jl:310)
    std::vector<double> sqmahal; // ‖U⁻ᵀ(y_s − m)‖² per column (src/finite_gp_projection.jl:325-326)
    int32_t info = 0;
};

// Shared by gp_logpdf and gp_posterior_fit.  Y: n×ncols column-major host.  If post != NULL the factor
// is kept and α is computed for column 0.
template <typename T>
static int32_t fit_impl(gp_ctx* c, const gp_kernel* k, const gp_points* x, const gp_noise* noise,
                        const void* mean_or_null, const void* Yv, long ldy, int ncols, FitOut& out, gp_post* post,
                        void* alpha_out) {
    const long n = x->n, np = round_up(n, 128);
    const int d = x->d;
    const long R = round_up(std::max(ncols, 1), 128);
    const long mtot = np + R;
    const long ld = np + c->ldpad;
    const T* Y = (const T*)Yv;
    const T* mean = (const T*)mean_or_null;

    c->ev_used = 0;
    c->gemm_recs.clear();
    for (auto& e : c->ev_phase)
        if (!e) HIPCHK(hipEventCreate(&e));
    if (!c->info_dev) HIPCHK(hipMalloc((void**)&c->info_dev, sizeof(int)));
    RC(ctx_scal(c, 8 + R));

    // ---- host marshalling
    std::vector<T> xs_h;
    scale_points<T>(k, x, np, xs_h);
    std::vector<T> noise_h((size_t)np, T(0));
    for (long i = 0; i < n; ++i) noise_h[i] = noise->kind == 0 ? (T)noise->s : ((const T*)noise->diag)[i];
    std::vector<T> rhs_h((size_t)ncols * np, T(0));  // rows: δ_sᵀ = (Y[:,s] - m)ᵀ, zero padded
    for (int s = 0; s < ncols; ++s)
        for (long i = 0; i < n; ++i) rhs_h[(size_t)s * np + i] = Y[(size_t)s * ldy + i] - (mean ? mean[i] : T(0));

    // ---- device buffers (RAII: every exit path returns what is not handed to the posterior handle)
    void *A_v = nullptr, *xs_v = nullptr, *noise_v = nullptr, *alpha_v = nullptr;
    const size_t A_bytes = sizeof(T) * (size_t)(mtot + 128) * ld;
    const size_t xs_bytes = sizeof(T) * (size_t)d * np, nz_bytes = sizeof(T) * (size_t)np;
    DevBufs bufs(c);
    RC(bufs.get(A_bytes, &A_v));
    RC(bufs.get(xs_bytes, &xs_v));
    RC(bufs.get(nz_bytes, &noise_v));
    RC(bufs.get(nz_bytes, &alpha_v));
    T* A = (T*)A_v;
    double logdet_half_out = 0;
    int32_t rc = [&]() -> int32_t {
        HIPCHK(hipEventRecord(c->ev_phase[0], c->sm));
        HIPCHK(hipMemcpyAsync(xs_v, xs_h.data(), xs_bytes, hipMemcpyHostToDevice, c->sm));
        HIPCHK(hipMemcpyAsync(noise_v, noise_h.data(), nz_bytes, hipMemcpyHostToDevice, c->sm));
        HIPCHK(hipMemsetAsync(c->info_dev, 0, sizeof(int), c->sm));
        HIPCHK(hipMemsetAsync(c->scal_dev, 0, sizeof(double) * (8 + R), c->sm));
        // RHS rows (+ the 128 slack rows) zeroed, then δ rows copied in
        HIPCHK(hipMemsetAsync(A + np * ld, 0, sizeof(T) * (size_t)(R + 128) * ld, c->sm));
        HIPCHK(hipMemcpy2DAsync(A + np * ld, sizeof(T) * ld, rhs_h.data(), sizeof(T) * np, sizeof(T) * np, ncols,
                                hipMemcpyHostToDevice, c->sm));
        RC(assemble_sym<T>(c, k, (const T*)xs_v, np, d, (const T*)noise_v, n, np, A, ld));
        HIPCHK(hipEventRecord(c->ev_phase[1], c->sm));
        RC(potrf_full<T>(c, A, ld, np, mtot, c->info_dev, n, c->scal_dev));
        HIPCHK(hipEventRecord(c->ev_phase[2], c->sm));
        // sqmahal per RHS row: ‖z_s‖², z_sᵀ = δ_sᵀ L⁻ᵀ sits in row np+s
        hipLaunchKernelGGL(rowsumsq_kernel<T>, dim3((unsigned)ncols), dim3(256), 0, c->sm, A + np * ld, ld, np,
                           c->scal_dev + 8);
        HIPCHK(hipGetLastError());
        if (post) {  // α = L⁻ᵀ z for column 0
            HIPCHK(hipMemcpyAsync(alpha_v, A + np * ld, sizeof(T) * np, hipMemcpyDeviceToDevice, c->sm));
            RC(trsv<T>(c, c->sm, A, ld, np, (T*)alpha_v, np, 1, false));
        }
        HIPCHK(hipEventRecord(c->ev_phase[3], c->sm));
        int info_h = 0;
        std::vector<double> scal_h(8 + ncols);
        HIPCHK(hipMemcpyAsync(&info_h, c->info_dev, sizeof(int), hipMemcpyDeviceToHost, c->sm));
        HIPCHK(hipMemcpyAsync(scal_h.data(), c->scal_dev, sizeof(double) * (8 + ncols), hipMemcpyDeviceToHost, c->sm));
        if (post && alpha_out)
            HIPCHK(hipMemcpyAsync(alpha_out, alpha_v, sizeof(T) * n, hipMemcpyDeviceToHost, c->sm));
        HIPCHK(hipStreamSynchronize(c->sm));
        out.info = info_h;
        out.logpdf.resize(ncols);
        out.sqmahal.resize(ncols);
        logdet_half_out = scal_h[0];
        const double logdet = 2.0 * scal_h[0];
        out.logdet = logdet;
        for (int s = 0; s < ncols; ++s) {
            out.sqmahal[s] = scal_h[8 + s];
            out.logpdf[s] = -0.5 * ((double)n * LOG2PI + logdet + scal_h[8 + s]);
        }
        // timings
        float ms;
        HIPCHK(hipEventElapsedTime(&ms, c->ev_phase[0], c->ev_phase[1]));
        c->tm.assemble_ms = ms;
        HIPCHK(hipEventElapsedTime(&ms, c->ev_phase[1], c->ev_phase[2]));
        c->tm.potrf_ms = ms;
        HIPCHK(hipEventElapsedTime(&ms, c->ev_phase[2], c->ev_phase[3]));
        c->tm.solve_ms = ms;
        HIPCHK(hipEventElapsedTime(&ms, c->ev_phase[0], c->ev_phase[3]));
        c->tm.total_ms = ms;
        c->tm.gemm_ms = 0;
        c->tm.gemm_flops = 0;
        c->tm.gemm_bytes = 0;
        c->tm.gemm_launches = (int64_t)c->gemm_recs.size();
        for (auto& r : c->gemm_recs) {
            HIPCHK(hipEventElapsedTime(&ms, r.a, r.b));
            c->tm.gemm_ms += ms;
            c->tm.gemm_flops += r.flops;
            c->tm.gemm_bytes += r.bytes;
            if (getenv("GPMI_DUMP_GEMM"))
                fprintf(stderr, "GEMM s%d M=%ld N=%ld K=%ld ms=%.4f tflops=%.2f\n", r.stream, r.M, r.N, r.K, ms, r.flops / ms / 1e9);
        }
        return 0;
    }();
    if (rc != 0) {
        (void)hipStreamSynchronize(c->sm);
        (void)hipStreamSynchronize(c->sp);
        return rc;
    }
    if (out.info != 0 || !post) return out.info;
    bufs.keep(A_v);
    bufs.keep(xs_v);
    bufs.keep(alpha_v);
    post->dtype = k->dtype;
    post->n = n; post->np = np; post->ld = ld; post->mtot = mtot; post->d = d;
    post->kind = k->kind; post->variance = k->variance; post->nscale = k->nscale;
    post->scale.clear();
    if (k->scale && k->nscale > 0) post->scale.assign(k->scale, k->scale + k->nscale);
    post->A = A_v; post->A_bytes = A_bytes;
    post->xs = xs_v; post->xs_bytes = xs_bytes;
    post->alpha = alpha_v; post->alpha_bytes = nz_bytes;
    post->logdet_half = logdet_half_out;
    return 0;
}

template <typename T>
static int32_t predict_impl(gp_post* post, const gp_points* xs, const void* pm, int what, void* mean_out,
                            void* var_out, void* cov_out) {
    gp_ctx* c = post->ctx;
    SkScope sk(c);
    const long n = post->n, np = post->np, ld = post->ld;
    const long ns = xs->n, nsp = round_up(ns, 128);
    const int d = post->d;
    gp_kernel k{};
    k.kind = post->kind; k.dtype = post->dtype; k.variance = post->variance; k.nscale = post->nscale;
    k.scale = post->scale.empty() ? nullptr : post->scale.data();
    std::vector<T> xs_h;
    scale_points<T>(&k, xs, nsp, xs_h);
    void* xs_v = nullptr;
    const size_t xs_bytes = sizeof(T) * (size_t)d * nsp;
    DevBufs bufs(c);
    RC(bufs.get(xs_bytes, &xs_v));
    void *m_v = nullptr, *X_v = nullptr, *C_v = nullptr;
    size_t m_bytes = sizeof(T) * (size_t)nsp, X_bytes = 0, C_bytes = 0;
    const T* prior_mean = (const T*)pm;
    const T* A = (const T*)post->A;
    c->ev_used = 0;
    c->gemm_recs.clear();
    int32_t rc = [&]() -> int32_t {
        HIPCHK(hipMemcpyAsync(xs_v, xs_h.data(), xs_bytes, hipMemcpyHostToDevice, c->sm));
        if (what & 1) {
            RC(bufs.get(m_bytes, &m_v));
            hipLaunchKernelGGL(kvec_kernel<T>, dim3((unsigned)ns), dim3(256), 0, c->sm, (const T*)xs_v, nsp,
                               (const T*)post->xs, np, d, post->kind, (T)post->variance, n, (const T*)post->alpha,
                               (T*)m_v);
            HIPCHK(hipGetLastError());
            std::vector<T> m_h(ns);
            HIPCHK(hipMemcpyAsync(m_h.data(), m_v, sizeof(T) * ns, hipMemcpyDeviceToHost, c->sm));
            HIPCHK(hipStreamSynchronize(c->sm));
            T* mo = (T*)mean_out;
            for (long i = 0; i < ns; ++i) mo[i] = (prior_mean ? prior_mean[i] : T(0)) + m_h[i];
        }
        if (what & 6) {
            const bool want_cov = (what & 4) != 0;
            // test points are processed in row chunks of the cross-covariance X = K_*x (chunk × np);
            // the full covariance needs all of X at once.
            const long chunk = want_cov ? nsp : std::min<long>(nsp, 4096);
            const long ldx = np + c->ldpad;
            X_bytes = sizeof(T) * (size_t)(chunk + 128) * ldx;
            RC(bufs.get(X_bytes, &X_v));
            RC(ctx_scal(c, 8 + chunk));
            T* X = (T*)X_v;
            std::vector<double> ss(chunk);
            T* vo = (T*)var_out;
            for (long r0 = 0; r0 < nsp; r0 += chunk) {
                const long rows = std::min(chunk, nsp - r0);
                GridMap g = plain_map(0, r0, 0);
                dim3 grid((unsigned)(np / 128), (unsigned)(rows / 128));
                launch_kmat<T>(grid, c->sm, X, ldx, (const T*)xs_v, nsp,
                                   (const T*)post->xs, np, d, post->kind, (T)post->variance, (const T*)nullptr, ns, n,
                                   0, g, (const T*)nullptr, (const T*)nullptr);
                HIPCHK(hipGetLastError());
                RC(trsm_post<T>(post, c->sm, X, ldx, rows, bufs));
                if (what & 2) {
                    hipLaunchKernelGGL(rowsumsq_kernel<T>, dim3((unsigned)rows), dim3(256), 0, c->sm, X, ldx, np,
                                       c->scal_dev + 8);
                    HIPCHK(hipGetLastError());
                    HIPCHK(hipMemcpyAsync(ss.data(), c->scal_dev + 8, sizeof(double) * rows, hipMemcpyDeviceToHost,
                                          c->sm));
                    HIPCHK(hipStreamSynchronize(c->sm));
                    for (long i = 0; i < rows && r0 + i < ns; ++i)
                        vo[r0 + i] = (T)((double)post->variance - ss[i]);
                }
            }
            if (want_cov) {
                const long ldc = nsp + c->ldpad;
                C_bytes = sizeof(T) * (size_t)(nsp + 128) * ldc;
                RC(bufs.get(C_bytes, &C_v));
                T* Cm = (T*)C_v;
                GridMap g = plain_map(0, 0, 0);
                dim3 grid((unsigned)(nsp / 128), (unsigned)(nsp / 128));
                launch_kmat<T>(grid, c->sm, Cm, ldc, (const T*)xs_v, nsp,
                                   (const T*)xs_v, nsp, d, post->kind, (T)post->variance, (const T*)nullptr, ns, ns,
                                   0, g, (const T*)nullptr, (const T*)nullptr);
                HIPCHK(hipGetLastError());
                RC(launch_gemm<T>(c, c->sm, Cm, ldc, X, ldx, X, ldx, nsp, nsp, np, plain_map(0, 0, 0)));
                // symmetric: row-major == column-major
                HIPCHK(hipMemcpy2DAsync(cov_out, sizeof(T) * ns, Cm, sizeof(T) * ldc, sizeof(T) * ns, ns,
                                        hipMemcpyDeviceToHost, c->sm));
                HIPCHK(hipStreamSynchronize(c->sm));
            }
        }
        return 0;
    }();
    if (rc != 0) (void)hipStreamSynchronize(c->sm);
    return rc;
}

// logpdf value + gradient (see include/gpmi355.h gp_logpdf_grad)
template <typename T>
static int32_t grad_impl(gp_ctx* c, const gp_kernel* k, const gp_points* x, const gp_noise* noise, const void* mean,
                         const void* y, void* logpdf_out, double* dvar, double* dscale, void* dnoise, void* dy, void* dx) {
    const long n = x->n;
    const int d = x->d;
    gp_post post{};
    post.ctx = c;
    FitOut fo;
    std::vector<T> alpha_h((size_t)n);
    RC(fit_impl<T>(c, k, x, noise, mean, y, n, 1, fo, &post, alpha_h.data()));
    SkScope sk(c);
    const long np = post.np, ld = post.ld;
    hipStream_t s = c->sm;
    void *W_v = 0, *Ci_v = 0, *g_v = 0, *dn_v = 0, *sc_v = 0, *gx_v = 0;
    const size_t gx_b = sizeof(double) * (size_t)d * np;
    std::vector<double> gx_h(dx ? (size_t)d * np : 0);
    const int nsc = std::max(k->nscale, 1);
    const size_t M_b = sizeof(T) * (size_t)(np + 128) * ld, g_b = sizeof(double) * (size_t)(2 + nsc), dn_b = sizeof(T) * (size_t)np;
    const size_t sc_b = sizeof(double) * (size_t)nsc;
    DevBufs bufs(c);
    bufs.v.push_back(post.A);  // the temporary posterior's blocks go back to the cache with everything else
    bufs.v.push_back(post.xs);
    bufs.v.push_back(post.alpha);
    std::vector<double> g_hv((size_t)(2 + nsc), 0.0);  // [0] ∂/∂variance, [1] Σ_i ∂/∂Σy_ii, [2 + p] ∂/∂scale_p
    double* g_h = g_hv.data();
    std::vector<T> dn_h((size_t)n);
    std::vector<double> sc_h((size_t)nsc, 1.0);
    for (int p = 0; p < k->nscale; ++p) sc_h[p] = k->scale[p];
    int32_t rc = [&]() -> int32_t {
        RC(bufs.get(M_b, &W_v));
        RC(bufs.get(M_b, &Ci_v));
        RC(bufs.get(g_b, &g_v));
        RC(bufs.get(dn_b, &dn_v));
        RC(bufs.get(sc_b, &sc_v));
        if (dx) RC(bufs.get(gx_b, &gx_v));
        T* W = (T*)W_v;
        T* Ci = (T*)Ci_v;
        HIPCHK(hipMemsetAsync(g_v, 0, g_b, s));
        HIPCHK(hipMemcpyAsync(sc_v, sc_h.data(), sc_b, hipMemcpyHostToDevice, s));
        // W is written in full by identity_kernel and Ci by the product below (beta0: C is overwritten, not read): only the 128 slack rows the
        // GEMM over-reads need defined values (until round 5 both N×N blocks were zeroed first: 2 × 34 GB of writes at C4)
        HIPCHK(hipMemsetAsync(W + np * ld, 0, sizeof(T) * (size_t)128 * ld, s));
        HIPCHK(hipMemsetAsync(Ci + np * ld, 0, sizeof(T) * (size_t)128 * ld, s));
        hipLaunchKernelGGL(identity_kernel<T>, dim3((unsigned)((np + 255) / 256), (unsigned)np), dim3(256), 0, s, W, ld, np);
        HIPCHK(hipGetLastError());
        if (c->dib_nb >= 128 && np >= 2 * c->dib_nb) {
            // L⁻ᵀ with the inverse diagonal blocks: dib_build (batched: one launch sequence for all blocks) writes L_bb⁻ᵀ onto W's diagonal and −L_bb⁻¹
            // into Wn; a leaf of the recursion is then ONE triangular-k GEMM over the rows above its block.  (With the blocks built one after the
            // other this did not pay — C2 93.0 -> 95.6 ms, C4 4.71 -> 4.72 s: the serial chains cost what the leaves they replaced cost.)
            const long nbi = round_up(c->dib_nb, 128), ldw = nbi + c->ldpad;
            const size_t wb = sizeof(T) * (size_t)(np + 128) * ldw;
            void *Wn_v = 0, *Iw_v = 0, *S_v = 0;
            RC(bufs.get(wb, &Wn_v));
            RC(bufs.get(wb, &Iw_v));
            RC(bufs.get(wb, &S_v));
            RC(dib_build<T>(c, s, (const T*)post.A, ld, np, nbi, (T*)Wn_v, ldw, (T*)Iw_v, W, ld));
            DibArgs<T> dib;
            dib.W = (const T*)Wn_v; dib.ldw = ldw; dib.nbi = nbi; dib.S = (T*)S_v; dib.lds = ldw;
            RC(trsm_upper_rec<T>(c, s, W, ld, (const T*)post.A, ld, 0, np, dib));
        } else {
            RC(trsm_upper_rec<T>(c, s, W, ld, (const T*)post.A, ld, 0, np));             // W = I L⁻ᵀ = L⁻ᵀ (upper, row-major)
        }
        {
            GridMap gw = plain_map(1, 0, 0);
            gw.ktri = 2;                                                                   // W upper: k starts at the row tile
            gw.beta0 = 1;                                                                  // Ci is overwritten (its upper tiles are never read)
            RC(launch_gemm<T>(c, s, Ci, ld, W, ld, W, ld, np, np, np, gw));                // Ci = −W Wᵀ = −C⁻¹ (lower), read as it is
        }
        // (no sign fold: the kernels below add the −C⁻¹ the product left — kernels.hpp)
        dim3 grid((unsigned)(np / 128), (unsigned)(np / 128));
        // one launch per chunk of 16 ARD scales (a single launch for scalar / no transform, whatever D is)
#define GPMI_KGRAD_FAST(ND_)                                                                                                                  \
    hipLaunchKernelGGL((kgrad_fast_kernel<T, ND_>), grid, dim3(256), 0, s, (const T*)Ci, ld, (const T*)post.xs, np, d, post.kind, (T)post.variance, \
                       post.nscale, (const double*)sc_v, (const T*)post.alpha, n, (double*)g_v)
        if (d <= 16 && (ld * (long)sizeof(T)) % 16 == 0) {  // the scratch-free form (16-byte loads of the weights)
            if (d <= 4) GPMI_KGRAD_FAST(4);
            else if (d <= 8) GPMI_KGRAD_FAST(8);
            else GPMI_KGRAD_FAST(16);
            HIPCHK(hipGetLastError());
        } else {
            for (int p0 = 0; p0 < (post.nscale > 1 ? post.nscale : 1); p0 += 16) {
                hipLaunchKernelGGL(kgrad_kernel<T>, grid, dim3(256), 0, s, (const T*)Ci, ld, (const T*)post.xs, np, d, post.kind,
                                   (T)post.variance, post.nscale, (const double*)sc_v, (const T*)post.alpha, n, (double*)g_v, p0);
                HIPCHK(hipGetLastError());
            }
        }
#undef GPMI_KGRAD_FAST
        hipLaunchKernelGGL(noise_grad_kernel<T>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const T*)Ci, ld,
                           (const T*)post.alpha, n, (T*)dn_v, (double*)g_v + 1);
        HIPCHK(hipGetLastError());
        if (dx) {  // ∂/∂x: full-square pass (the mirrored C⁻¹ entry serves the tiles above the diagonal), 16 dimensions per launch
            HIPCHK(hipMemsetAsync(gx_v, 0, gx_b, s));
            for (int p0 = 0; p0 < d; p0 += 16) {
                hipLaunchKernelGGL(kgradx_kernel<T>, grid, dim3(256), 0, s, (const T*)Ci, ld, (const T*)post.xs, np, d, post.kind,
                                   (T)post.variance, post.nscale, (const double*)sc_v, (const T*)post.alpha, n, (double*)gx_v, np, p0);
                HIPCHK(hipGetLastError());
            }
            HIPCHK(hipMemcpyAsync(gx_h.data(), gx_v, gx_b, hipMemcpyDeviceToHost, s));
        }
        HIPCHK(hipMemcpyAsync(g_h, g_v, g_b, hipMemcpyDeviceToHost, s));
        HIPCHK(hipMemcpyAsync(dn_h.data(), dn_v, sizeof(T) * (size_t)n, hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        return 0;
    }();
    if (rc != 0) {
        (void)hipStreamSynchronize(c->sm);
        (void)hipStreamSynchronize(c->sp);
    }
    if (rc != 0) return rc;
    *(T*)logpdf_out = (T)fo.logpdf[0];
    if (dvar) *dvar = g_h[0];
    if (dscale)
        for (int p = 0; p < k->nscale; ++p) dscale[p] = g_h[2 + p];
    if (dnoise) {
        if (noise->kind == 0) *(T*)dnoise = (T)g_h[1];
        else memcpy(dnoise, dn_h.data(), sizeof(T) * (size_t)n);
    }
    if (dy)
        for (long i = 0; i < n; ++i) ((T*)dy)[i] = -alpha_h[i];
    if (dx) {  // same container layout as the inputs (src/finite_gp_projection.jl:32-37)
        T* o = (T*)dx;
        for (int dd = 0; dd < d; ++dd)
            for (long i = 0; i < n; ++i) {
                const T v = (T)gx_h[(size_t)dd * np + i];
                if (x->layout == 0) o[i] = v;
                else if (x->layout == 1) o[(long)dd + i * d] = v;
                else o[i + (long)dd * n] = v;
            }
    }
    return 0;
}

// Sequential conditioning: bordered Cholesky on the device (reference src/exact_gpr_posterior.jl:46-56,
// src/util/common_covmat_ops.jl:38-42).
template <typename T>
static int32_t update_impl(gp_post* old, const gp_points* x2, const gp_noise* noise2, const void* delta_all, gp_post* post,
                           void* alpha_out, double* logpdf_out) {
    gp_ctx* c = old->ctx;
    const long n1 = old->n, np1 = old->np, ld1 = old->ld;
    const long n2 = x2->n, n2p = round_up(n2, 128);
    const long n = n1 + n2, np = round_up(n, 128), ld = np + c->ldpad, mtot = np + 128;
    const int d = old->d;
    gp_kernel k{};
    k.kind = old->kind; k.dtype = old->dtype; k.variance = old->variance; k.nscale = old->nscale;
    k.scale = old->scale.empty() ? nullptr : old->scale.data();
    SkScope sk(c);
    hipStream_t s = c->sm;
    c->ev_used = 0;
    c->gemm_recs.clear();
    if (!c->info_dev) HIPCHK(hipMalloc((void**)&c->info_dev, sizeof(int)));
    RC(ctx_scal(c, 16));

    std::vector<T> x2s_h;
    scale_points<T>(&k, x2, n2p, x2s_h);  // [d][n2p]
    std::vector<T> noise_h((size_t)n2p, T(0));
    for (long i = 0; i < n2; ++i) noise_h[i] = noise2->kind == 0 ? (T)noise2->s : ((const T*)noise2->diag)[i];
    std::vector<T> delta_h((size_t)np, T(0));
    memcpy(delta_h.data(), delta_all, sizeof(T) * (size_t)n);

    void *A_v = 0, *xs_v = 0, *alpha_v = 0, *X_v = 0, *S_v = 0, *x2_v = 0, *nz_v = 0;
    const long ldx = np1 + c->ldpad, lds = n2p + c->ldpad;
    const size_t A_b = sizeof(T) * (size_t)(mtot + 128) * ld, xs_b = sizeof(T) * (size_t)d * np, v_b = sizeof(T) * (size_t)np;
    const size_t X_b = sizeof(T) * (size_t)(n2p + 128) * ldx, S_b = sizeof(T) * (size_t)(n2p + 128) * lds;
    const size_t x2_b = sizeof(T) * (size_t)d * n2p, nz_b = sizeof(T) * (size_t)n2p;
    DevBufs bufs(c);
    RC(bufs.get(A_b, &A_v));
    RC(bufs.get(xs_b, &xs_v));
    RC(bufs.get(v_b, &alpha_v));
    RC(bufs.get(X_b, &X_v));
    RC(bufs.get(S_b, &S_v));
    RC(bufs.get(x2_b, &x2_v));
    RC(bufs.get(nz_b, &nz_v));
    T* A = (T*)A_v;
    T* X = (T*)X_v;
    T* S = (T*)S_v;
    const T* A1 = (const T*)old->A;
    int info_h = 0;
    double scal_h[16] = {0};
    int32_t rc = [&]() -> int32_t {
        HIPCHK(hipMemcpyAsync(x2_v, x2s_h.data(), x2_b, hipMemcpyHostToDevice, s));
        HIPCHK(hipMemcpyAsync(nz_v, noise_h.data(), nz_b, hipMemcpyHostToDevice, s));
        HIPCHK(hipMemsetAsync(c->info_dev, 0, sizeof(int), s));
        HIPCHK(hipMemsetAsync(c->scal_dev, 0, sizeof(double) * 16, s));
        // combined scaled inputs [d][np]: old block, then the new points
        HIPCHK(hipMemsetAsync(xs_v, 0, xs_b, s));
        HIPCHK(hipMemcpy2DAsync(xs_v, sizeof(T) * np, old->xs, sizeof(T) * np1, sizeof(T) * n1, d, hipMemcpyDeviceToDevice, s));
        HIPCHK(hipMemcpy2DAsync((T*)xs_v + n1, sizeof(T) * np, x2_v, sizeof(T) * n2p, sizeof(T) * n2, d, hipMemcpyDeviceToDevice, s));
        // X = K(x2, x1) (n2p × np1; padding rows / columns zero), then X ← X L11⁻ᵀ = U12ᵀ                  :39
        {
            GridMap g = plain_map(0, 0, 0);
            dim3 grid((unsigned)(np1 / 128), (unsigned)(n2p / 128));
            launch_kmat<T>(grid, s, X, ldx, (const T*)x2_v, n2p, (const T*)old->xs, np1, d,
                               k.kind, (T)k.variance, (const T*)nullptr, n2, n1, 0, g, (const T*)nullptr, (const T*)nullptr);
            HIPCHK(hipGetLastError());
        }
        RC(trsm_post<T>(old, s, X, ldx, n2p, bufs));
        // S = C22 − U12ᵀ U12 (lower), chol(S) = U22ᵀ                                                       :40
        {
            GridMap g = plain_map(1, 0, 0);
            dim3 grid((unsigned)(n2p / 128), (unsigned)(n2p / 128));
            launch_kmat<T>(grid, s, S, lds, (const T*)x2_v, n2p, (const T*)x2_v, n2p, d,
                               k.kind, (T)k.variance, (const T*)nz_v, n2, n2, 1, g, (const T*)nullptr, (const T*)nullptr);
            HIPCHK(hipGetLastError());
        }
        RC(launch_gemm<T>(c, s, S, lds, X, ldx, X, ldx, n2p, n2p, np1, plain_map(1, 0, 0)));
        RC(potrf_full<T>(c, S, lds, n2p, n2p, c->info_dev, n2, c->scal_dev));
        // new factor [L11 0; U12ᵀ U22ᵀ], identity padding, δ as the right-hand side                          :41
        HIPCHK(hipMemsetAsync(A + np * ld, 0, sizeof(T) * (size_t)(128 + 128) * ld, s));
        HIPCHK(hipMemcpy2DAsync(A, sizeof(T) * ld, A1, sizeof(T) * ld1, sizeof(T) * n1, n1, hipMemcpyDeviceToDevice, s));
        HIPCHK(hipMemcpy2DAsync(A + n1 * ld, sizeof(T) * ld, X, sizeof(T) * ldx, sizeof(T) * n1, n2, hipMemcpyDeviceToDevice, s));
        HIPCHK(hipMemcpy2DAsync(A + n1 * ld + n1, sizeof(T) * ld, S, sizeof(T) * lds, sizeof(T) * n2, n2, hipMemcpyDeviceToDevice, s));
        if (np > n) {
            hipLaunchKernelGGL(pad_identity_kernel<T>, dim3((unsigned)((np + 255) / 256), (unsigned)(np - n)), dim3(256), 0, s, A,
                               ld, n, np);
            HIPCHK(hipGetLastError());
        }
        HIPCHK(hipMemcpyAsync(alpha_v, delta_h.data(), v_b, hipMemcpyHostToDevice, s));
        RC(trsv<T>(c, s, A, ld, np, (T*)alpha_v, np, 1, true));                                     // z = L⁻¹ δ
        hipLaunchKernelGGL(rowsumsq_kernel<T>, dim3(1), dim3(256), 0, s, (const T*)alpha_v, np, np, c->scal_dev + 8);
        HIPCHK(hipGetLastError());
        RC(trsv<T>(c, s, A, ld, np, (T*)alpha_v, np, 1, false));                                    // α = L⁻ᵀ z        :53
        HIPCHK(hipMemcpyAsync(&info_h, c->info_dev, sizeof(int), hipMemcpyDeviceToHost, s));
        HIPCHK(hipMemcpyAsync(scal_h, c->scal_dev, sizeof(double) * 16, hipMemcpyDeviceToHost, s));
        if (alpha_out) HIPCHK(hipMemcpyAsync(alpha_out, alpha_v, sizeof(T) * n, hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        return 0;
    }();
    if (rc != 0) {
        (void)hipStreamSynchronize(c->sm);
        (void)hipStreamSynchronize(c->sp);
    }
    if (rc == 0 && info_h != 0) rc = (int32_t)n1 + info_h;  // order of the failing leading minor of the bordered matrix
    if (rc != 0) return rc;
    bufs.keep(A_v);
    bufs.keep(xs_v);
    bufs.keep(alpha_v);
    post->ctx = c;
    post->dtype = old->dtype;
    post->n = n; post->np = np; post->ld = ld; post->mtot = mtot; post->d = d;
    post->kind = old->kind; post->variance = old->variance; post->nscale = old->nscale;
    post->scale = old->scale;
    post->A = A_v; post->A_bytes = A_b;
    post->xs = xs_v; post->xs_bytes = xs_b;
    post->alpha = alpha_v; post->alpha_bytes = v_b;
    post->logdet_half = old->logdet_half + scal_h[0];
    if (logpdf_out) *logpdf_out = -0.5 * ((double)n * LOG2PI + 2.0 * post->logdet_half + scal_h[8]);
    return 0;
}

template <typename T> static int32_t factor_mul_impl(gp_post* post, const void* xi, int ncols, void* out) {
    gp_ctx* c = post->ctx;
    const long n = post->n, np = post->np;
    void *in_v = 0, *out_v = 0;
    const size_t b = sizeof(T) * (size_t)np * ncols;
    DevBufs bufs(c);
    RC(bufs.get(b, &in_v));
    RC(bufs.get(b, &out_v));
    hipStream_t s = c->sm;
    int32_t rc = [&]() -> int32_t {
        HIPCHK(hipMemsetAsync(in_v, 0, b, s));
        HIPCHK(hipMemcpy2DAsync(in_v, sizeof(T) * np, xi, sizeof(T) * n, sizeof(T) * n, ncols, hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(trmv_lower_kernel<T>, dim3((unsigned)n), dim3(256), 0, s, (const T*)post->A, post->ld, (const T*)in_v, np,
                           ncols, (T*)out_v);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpy2DAsync(out, sizeof(T) * n, out_v, sizeof(T) * np, sizeof(T) * n, ncols, hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        return 0;
    }();
    if (rc != 0) (void)hipStreamSynchronize(s);
    return rc;
}

// out[:, s] = C \ B[:, s] with the resident factor (forward + backward vector sweeps, all columns per sweep)
template <typename T> static int32_t solve_impl(gp_post* post, const void* B, int ncols, void* out) {
    gp_ctx* c = post->ctx;
    const long n = post->n, np = post->np;
    void* r_v = 0;
    const size_t b = sizeof(T) * (size_t)np * ncols;
    DevBufs bufs(c);
    RC(bufs.get(b, &r_v));
    hipStream_t s = c->sm;
    int32_t rc = [&]() -> int32_t {
        HIPCHK(hipMemsetAsync(r_v, 0, b, s));
        HIPCHK(hipMemcpy2DAsync(r_v, sizeof(T) * np, B, sizeof(T) * n, sizeof(T) * n, ncols, hipMemcpyHostToDevice, s));
        RC(trsv<T>(c, s, (const T*)post->A, post->ld, np, (T*)r_v, np, ncols, true));
        RC(trsv<T>(c, s, (const T*)post->A, post->ld, np, (T*)r_v, np, ncols, false));
        HIPCHK(hipMemcpy2DAsync(out, sizeof(T) * n, r_v, sizeof(T) * np, sizeof(T) * n, ncols, hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        return 0;
    }();
    if (rc != 0) (void)hipStreamSynchronize(s);
    return rc;
}

// ------------------------------------------------------------------------------------------------
// Joint predictive distribution on the device: logpdf(post(x*, Σy*), y*) and rand(post(x*, Σy*)) without leaving HBM.
// Reference: a FiniteGP over a PosteriorGP goes through the generic path — mean_and_cov(fx) (src/finite_gp_projection.jl:133-136
// on top of src/exact_gpr_posterior.jl:78-83), cholesky (:308 / :235), logdet + _sqmahal (:310, :325-326) or m + C.U'ξ (:236).
// TC = compute type of the N*×N* side (the posterior's dtype; always fp64 for VFE), TIO = host array type.
// ------------------------------------------------------------------------------------------------
template <typename TC> struct Joint {
    void* C = nullptr;  // (nsp + R + 128) × ld, row-major lower triangle of cov(f_post, x*) + Σy*, identity padding; R RHS rows
    long ns = 0, nsp = 0, ld = 0, R = 0;
    std::vector<double> mean;  // m(x*) + K_*· α  (host)
};

template <typename TC, typename TIO>
static void noise_to(const gp_noise* noise, long ns, long nsp, std::vector<TC>& out) {
    out.assign((size_t)nsp, TC(0));
    if (!noise) return;
    for (long i = 0; i < ns; ++i) out[i] = noise->kind == 0 ? (TC)noise->s : (TC)((const TIO*)noise->diag)[i];
}

template <typename T>
static int32_t post_joint(gp_post* post, const gp_points* xs, const void* pm, const gp_noise* noise, long R, DevBufs& bufs,
                          Joint<T>& J) {
    gp_ctx* c = post->ctx;
    SkScope sk(c);
    const long n = post->n, np = post->np, ld = post->ld;
    const long ns = xs->n, nsp = round_up(ns, 128);
    const int d = post->d;
    gp_kernel k{};
    k.kind = post->kind; k.dtype = post->dtype; k.variance = post->variance; k.nscale = post->nscale;
    k.scale = post->scale.empty() ? nullptr : post->scale.data();
    std::vector<T> xs_h, nz_h;
    scale_points<T>(&k, xs, nsp, xs_h);
    noise_to<T, T>(noise, ns, nsp, nz_h);
    const long ldx = np + c->ldpad, ldc = nsp + c->ldpad;
    void *xs_v = 0, *m_v = 0, *X_v = 0, *nz_v = 0;
    RC(bufs.get(sizeof(T) * (size_t)d * nsp, &xs_v));
    RC(bufs.get(sizeof(T) * (size_t)nsp, &m_v));
    RC(bufs.get(sizeof(T) * (size_t)nsp, &nz_v));
    RC(bufs.get(sizeof(T) * (size_t)(nsp + 128) * ldx, &X_v));
    RC(bufs.get(sizeof(T) * (size_t)(nsp + R + 128) * ldc, &J.C));
    J.ns = ns; J.nsp = nsp; J.ld = ldc; J.R = R;
    T* X = (T*)X_v;
    T* Cm = (T*)J.C;
    hipStream_t s = c->sm;
    c->ev_used = 0;
    c->gemm_recs.clear();
    std::vector<T> m_h((size_t)ns);
    HIPCHK(hipMemcpyAsync(xs_v, xs_h.data(), sizeof(T) * (size_t)d * nsp, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(nz_v, nz_h.data(), sizeof(T) * (size_t)nsp, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(kvec_kernel<T>, dim3((unsigned)ns), dim3(256), 0, s, (const T*)xs_v, nsp, (const T*)post->xs, np, d,
                       post->kind, (T)post->variance, n, (const T*)post->alpha, (T*)m_v);                       // K_*x α
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(m_h.data(), m_v, sizeof(T) * (size_t)ns, hipMemcpyDeviceToHost, s));
    {
        GridMap g = plain_map(0, 0, 0);
        dim3 grid((unsigned)(np / 128), (unsigned)(nsp / 128));
        launch_kmat<T>(grid, s, X, ldx, (const T*)xs_v, nsp, (const T*)post->xs, np, d,
                           post->kind, (T)post->variance, (const T*)nullptr, ns, n, 0, g, (const T*)nullptr, (const T*)nullptr);
        HIPCHK(hipGetLastError());
    }
    RC(trsm_post<T>(post, s, X, ldx, nsp, bufs));                                                              // V ᵀ = K_*x L⁻ᵀ
    {
        GridMap g = plain_map(1, 0, 0);
        dim3 grid((unsigned)(nsp / 128), (unsigned)(nsp / 128));
        launch_kmat<T>(grid, s, Cm, ldc, (const T*)xs_v, nsp, (const T*)xs_v, nsp, d,
                           post->kind, (T)post->variance, (const T*)nz_v, ns, ns, 1, g, (const T*)nullptr, (const T*)nullptr);
        HIPCHK(hipGetLastError());
    }
    HIPCHK(hipMemsetAsync(Cm + nsp * ldc, 0, sizeof(T) * (size_t)(R + 128) * ldc, s));
    RC(launch_gemm<T>(c, s, Cm, ldc, X, ldx, X, ldx, nsp, nsp, np, plain_map(1, 0, 0)));                       // K** + Σy* − VᵀV
    HIPCHK(hipStreamSynchronize(s));
    const T* prior_mean = (const T*)pm;
    J.mean.resize((size_t)ns);
    for (long i = 0; i < ns; ++i) J.mean[i] = (double)(prior_mean ? prior_mean[i] : T(0)) + (double)m_h[i];
    return 0;
}

// The same joint for a multi-device posterior whose factor still lives as block-cyclic pieces (fp64): mean from α, V ᵀV from the
// forward solve ON the pieces (multi.hip: multi_predict_var) — the factor is not gathered.
static int32_t post_joint_dist(gp_post* post, const gp_points* xs, const void* pm, const gp_noise* noise, long R, DevBufs& bufs,
                               Joint<double>& J) {
    gp_ctx* c = post->ctx;
    const long ns = xs->n, nsp = round_up(ns, 128);
    const int d = post->d;
    gp_kernel k{};
    k.kind = post->kind; k.dtype = 0; k.variance = post->variance; k.nscale = post->nscale;
    k.scale = post->scale.empty() ? nullptr : post->scale.data();
    std::vector<double> m_h((size_t)ns), xs_n, xs_p, nz_h;
    RC(predict_impl<double>(post, xs, pm, 1, m_h.data(), nullptr, nullptr));   // m(x*) + K_*x α (no factor involved)
    scale_points<double>(&k, xs, ns, xs_n);
    scale_points<double>(&k, xs, nsp, xs_p);
    noise_to<double, double>(noise, ns, nsp, nz_h);
    std::vector<double> sub((size_t)ns), csub((size_t)ns * ns);
    RC(multi_predict_var(post, xs_n.data(), ns, ns, sub.data(), csub.data()));
    const long ldc = nsp + c->ldpad;
    void *xs_v = 0, *nz_v = 0, *W_v = 0;
    RC(bufs.get(sizeof(double) * (size_t)d * nsp, &xs_v));
    RC(bufs.get(sizeof(double) * (size_t)nsp, &nz_v));
    RC(bufs.get(sizeof(double) * (size_t)nsp * ldc, &W_v));
    RC(bufs.get(sizeof(double) * (size_t)(nsp + R + 128) * ldc, &J.C));
    J.ns = ns; J.nsp = nsp; J.ld = ldc; J.R = R;
    double* Cm = (double*)J.C;
    std::vector<double> W((size_t)nsp * ldc, 0.0);
    for (long i = 0; i < ns; ++i)
        for (long j = 0; j < ns; ++j) W[(size_t)i * ldc + j] = -csub[(size_t)i * ns + j];
    hipStream_t s = c->sm;
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipMemcpyAsync(xs_v, xs_p.data(), sizeof(double) * xs_p.size(), hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(nz_v, nz_h.data(), sizeof(double) * (size_t)nsp, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(W_v, W.data(), sizeof(double) * W.size(), hipMemcpyHostToDevice, s));
    {
        GridMap g = plain_map(1, 0, 0);
        dim3 grid((unsigned)(nsp / 128), (unsigned)(nsp / 128));
        launch_kmat<double>(grid, s, Cm, ldc, (const double*)xs_v, nsp, (const double*)xs_v, nsp, d, post->kind,
                           post->variance, (const double*)nz_v, ns, ns, 1, g, (const double*)nullptr, (const double*)nullptr);
        HIPCHK(hipGetLastError());
    }
    HIPCHK(hipMemsetAsync(Cm + nsp * ldc, 0, sizeof(double) * (size_t)(R + 128) * ldc, s));
    RC(gpmi::eng_add_vec(c, s, Cm, (const double*)W_v, nsp * ldc));                                           // K** + Σy* − VᵀV
    HIPCHK(hipStreamSynchronize(s));
    J.mean.assign(m_h.begin(), m_h.end());
    return 0;
}

// logpdf of Y (ns × ncols column-major host, leading dimension ldy) under N(J.mean, J.C): factor J.C in place with δ rows riding along
template <typename TC, typename TIO>
static int32_t joint_logpdf(gp_ctx* c, Joint<TC>& J, const void* Yv, long ldy, int ncols, void* outv) {
    const long ns = J.ns, nsp = J.nsp, ld = J.ld;
    TC* Cm = (TC*)J.C;
    hipStream_t s = c->sm;
    const TIO* Y = (const TIO*)Yv;
    if (!c->info_dev) HIPCHK(hipMalloc((void**)&c->info_dev, sizeof(int)));
    RC(ctx_scal(c, 8 + J.R));
    std::vector<TC> rhs((size_t)ncols * nsp, TC(0));
    for (int q = 0; q < ncols; ++q)
        for (long i = 0; i < ns; ++i) rhs[(size_t)q * nsp + i] = (TC)((double)Y[(size_t)q * ldy + i] - J.mean[i]);
    int info_h = 0;
    std::vector<double> scal_h(8 + ncols);
    int32_t rc = [&]() -> int32_t {
        HIPCHK(hipMemsetAsync(c->info_dev, 0, sizeof(int), s));
        HIPCHK(hipMemsetAsync(c->scal_dev, 0, sizeof(double) * (8 + J.R), s));
        HIPCHK(hipMemcpy2DAsync(Cm + nsp * ld, sizeof(TC) * ld, rhs.data(), sizeof(TC) * nsp, sizeof(TC) * nsp, ncols,
                                hipMemcpyHostToDevice, s));
        RC(potrf_full<TC>(c, Cm, ld, nsp, nsp + J.R, c->info_dev, ns, c->scal_dev));
        hipLaunchKernelGGL(rowsumsq_kernel<TC>, dim3((unsigned)ncols), dim3(256), 0, s, Cm + nsp * ld, ld, nsp, c->scal_dev + 8);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(&info_h, c->info_dev, sizeof(int), hipMemcpyDeviceToHost, s));
        HIPCHK(hipMemcpyAsync(scal_h.data(), c->scal_dev, sizeof(double) * (8 + ncols), hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        return 0;
    }();
    if (rc != 0) {
        (void)hipStreamSynchronize(c->sm);
        (void)hipStreamSynchronize(c->sp);
        return rc;
    }
    if (info_h != 0) return info_h;
    for (int q = 0; q < ncols; ++q)
        ((TIO*)outv)[q] = (TIO)(-0.5 * ((double)ns * LOG2PI + 2.0 * scal_h[0] + scal_h[8 + q]));
    return 0;
}

// out[:, q] = J.mean + chol(J.C)ᵀ-factor product with xi[:, q]  (ns × ncols column-major host arrays, leading dimension ns)
template <typename TC, typename TIO>
static int32_t joint_rand(gp_ctx* c, Joint<TC>& J, DevBufs& bufs, const void* xiv, int ncols, void* outv) {
    const long ns = J.ns, nsp = J.nsp, ld = J.ld;
    TC* Cm = (TC*)J.C;
    hipStream_t s = c->sm;
    const TIO* xi = (const TIO*)xiv;
    if (!c->info_dev) HIPCHK(hipMalloc((void**)&c->info_dev, sizeof(int)));
    RC(ctx_scal(c, 16));
    std::vector<TC> in_h((size_t)ncols * nsp, TC(0)), out_h((size_t)ncols * nsp);
    for (int q = 0; q < ncols; ++q)
        for (long i = 0; i < ns; ++i) in_h[(size_t)q * nsp + i] = (TC)xi[(size_t)q * ns + i];
    void *in_v = 0, *out_v = 0;
    const size_t b = sizeof(TC) * (size_t)nsp * ncols;
    RC(bufs.get(b, &in_v));
    RC(bufs.get(b, &out_v));
    int info_h = 0;
    int32_t rc = [&]() -> int32_t {
        HIPCHK(hipMemsetAsync(c->info_dev, 0, sizeof(int), s));
        HIPCHK(hipMemsetAsync(c->scal_dev, 0, sizeof(double) * 16, s));
        HIPCHK(hipMemcpyAsync(in_v, in_h.data(), b, hipMemcpyHostToDevice, s));
        RC(potrf_full<TC>(c, Cm, ld, nsp, nsp, c->info_dev, ns, c->scal_dev));
        hipLaunchKernelGGL(trmv_lower_kernel<TC>, dim3((unsigned)ns), dim3(256), 0, s, (const TC*)Cm, ld, (const TC*)in_v, nsp, ncols,
                           (TC*)out_v);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(&info_h, c->info_dev, sizeof(int), hipMemcpyDeviceToHost, s));
        HIPCHK(hipMemcpyAsync(out_h.data(), out_v, b, hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        return 0;
    }();
    if (rc != 0) {
        (void)hipStreamSynchronize(c->sm);
        (void)hipStreamSynchronize(c->sp);
        return rc;
    }
    if (info_h != 0) return info_h;
    for (int q = 0; q < ncols; ++q)
        for (long i = 0; i < ns; ++i) ((TIO*)outv)[(size_t)q * ns + i] = (TIO)(J.mean[i] + (double)out_h[(size_t)q * nsp + i]);
    return 0;
}

#include "vfe.hpp"

// ------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------
extern "C" {

int32_t gp_abi_version(void) { return GPMI355_ABI_VERSION; }
const char* gp_last_error(void) { return g_err.c_str(); }

int32_t gp_ctx_create(gp_ctx** out, int32_t device, void* stream_or_null) {
    if (!out) return set_arg_err(1, "out is NULL");
    *out = nullptr;
    int ndev = 0;
    HIPCHK(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return set_arg_err(2, "no such device");
    HIPCHK(hipSetDevice(device));
    gp_ctx* c = new gp_ctx();
    c->device = device;
    if (stream_or_null) {
        c->sm = (hipStream_t)stream_or_null;
    } else {
        hipError_t e = hipStreamCreateWithFlags(&c->sm, hipStreamNonBlocking);
        if (e != hipSuccess) {
            delete c;
            return set_hip_err(e, "hipStreamCreate", __LINE__);
        }
        c->own_sm = true;
    }
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) c->num_cus = prop.multiProcessorCount;
    }
    int lo = 0, hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
    hipError_t e = hipStreamCreateWithPriority(&c->sp, hipStreamNonBlocking, hi);
    if (e != hipSuccess) {
        if (c->own_sm) (void)hipStreamDestroy(c->sm);
        delete c;
        return set_hip_err(e, "hipStreamCreateWithPriority", __LINE__);
    }
    reg_add(c);
    *out = c;
    // GPMI_PARAMS="name=value,name=value": tuning overrides for experiments (same names as gp_ctx_set_param)
    if (const char* s = getenv("GPMI_PARAMS")) {
        std::string all(s);
        size_t pos = 0;
        while (pos < all.size()) {
            size_t e = all.find(',', pos);
            if (e == std::string::npos) e = all.size();
            const std::string kv = all.substr(pos, e - pos);
            const size_t q = kv.find('=');
            if (q != std::string::npos) (void)gp_ctx_set_param(c, kv.substr(0, q).c_str(), atoll(kv.c_str() + q + 1));
            pos = e + 1;
        }
    }
    return 0;
}

int32_t gp_ctx_destroy(gp_ctx* c) {
    if (!c || !reg_take(c)) return set_arg_err(1, "not a live gp_ctx");
    gp_multi* multi = nullptr;
    {
        std::lock_guard<std::mutex> l(c->mu);
        (void)hipSetDevice(c->device);
        (void)hipStreamSynchronize(c->sm);
        (void)hipStreamSynchronize(c->sp);
        c->dead = true;
        multi = c->multi;
        c->multi = nullptr;
    }
    if (multi) multi_destroy(multi);  // rank contexts, comm streams, RCCL communicators (factor pieces still alive keep their rank ctx)
    (void)hipSetDevice(c->device);
    ctx_unref(c);
    return 0;
}

int32_t gp_ctx_set_param(gp_ctx* c, const char* name, int64_t v) {
    if (!c || !reg_has(c)) return set_arg_err(1, "not a live gp_ctx");
    if (!name) return set_arg_err(2, "name is NULL");
    std::lock_guard<std::mutex> l(c->mu);
    if (c->multi && multi_set_param(c, name, v) == 0) return 0;  // (generic names are forwarded to the rank contexts too)
    if (!strcmp(name, "nb")) c->nb = v < 0 ? -1 : (v == 0 ? 0 : round_up(v, 128));
    else if (!strcmp(name, "nb_small")) c->nb_small = (v <= 0) ? 0 : round_up(v, 128);
    else if (!strcmp(name, "nb_large")) c->nb_large = (v <= 0) ? 0 : round_up(v, 128);
    else if (!strcmp(name, "lookahead")) c->lookahead = v != 0;
    else if (!strcmp(name, "lookahead_min_n")) c->lookahead_min_n = std::max<int64_t>(0, v);
    else if (!strcmp(name, "time_kernels")) c->time_kernels = v != 0;
    else if (!strcmp(name, "xcd_swizzle")) c->xcd_swizzle = v != 0;
    else if (!strcmp(name, "gemm_streamk")) c->gemm_streamk = v != 0;
    else if (!strcmp(name, "sk_max_tiles")) c->sk_max_tiles = v;
    else if (!strcmp(name, "sk_min_k")) c->sk_min_k = v;
    else if (!strcmp(name, "gemm_pipe")) c->gemm_pipe = v != 0;
    else if (!strcmp(name, "kmat_rows")) g_kmat_rows = v != 0;
    else if (!strcmp(name, "dib_nb")) c->dib_nb = v <= 0 ? 0 : std::min<int64_t>(round_up(std::max<int64_t>(v, 128), 128), 8192);
    else if (!strcmp(name, "gemm_pad_f32")) c->gemm_pad_f32 = std::min<int64_t>(std::max<int64_t>(0, v), 32768);
    else if (!strcmp(name, "gemm_pad_lds")) {
        c->gemm_pad_lds = std::min<int64_t>(std::max<int64_t>(0, v), 32768);
        c->gemm_pad_user = true;
    }
    else if (!strcmp(name, "trsv_nb")) c->trsv_nb = v >= 1024 ? 1024 : (v >= 512 ? 512 : (v >= 256 ? 256 : 128));
    else if (!strcmp(name, "deterministic")) c->deterministic = v != 0;
    else if (!strcmp(name, "leaf_v2")) c->leaf_v2 = v != 0;
    else if (!strcmp(name, "leaf_xr")) c->leaf_xr = v == 64 ? 64 : (v == 128 ? 128 : 0);
    else if (!strcmp(name, "leaf_cols")) c->leaf_cols = v == 64 ? 64 : 128;
    else if (!strcmp(name, "updk_max_k")) c->updk_max_k = std::max<int64_t>(0, v);
    else if (!strcmp(name, "updk_rt")) c->updk_rt = (int)v;
    else if (!strcmp(name, "updk_tall_k")) c->updk_tall_k = std::max<int64_t>(0, v);
    else if (!strcmp(name, "updk_tall_m")) c->updk_tall_m = std::max<int64_t>(0, v);
    else if (!strcmp(name, "upd128")) c->upd128 = v != 0;
    else if (!strcmp(name, "leaf_group")) c->leaf_group = v < 128 ? 64 : (v >= 512 ? 512 : (v >= 256 ? 256 : 128));
    else if (!strcmp(name, "xcd_min_tiles")) c->xcd_min_tiles = v;
    else if (!strcmp(name, "ldpad")) c->ldpad = round_up(std::max<int64_t>(0, v), 16);
    else if (!strcmp(name, "vfe_ks")) c->vfe_ks = std::max<int64_t>(512, round_up(v, 512));
    else if (!strcmp(name, "vfe_sk")) c->vfe_sk = v != 0;
    else if (!strcmp(name, "vfe_dual")) c->vfe_dual = v != 0;
    else if (!strcmp(name, "vfe_inv_nb")) c->vfe_inv_nb = v <= 0 ? 0 : round_up(v, 128);
    else if (!strcmp(name, "vfe_overlap")) c->vfe_overlap = v != 0;
    else if (!strcmp(name, "vfe_chunk")) c->vfe_chunk = v <= 0 ? 0 : std::max<int64_t>(2048, round_up(v, 2048));
    else if (!strcmp(name, "pool_cap_mb")) c->pool_cap = (size_t)std::max<int64_t>(0, v) << 20;
    else if (!strcmp(name, "lookahead_depth") || !strcmp(name, "dist_nb") || !strcmp(name, "copy_kernel") || !strcmp(name, "multi_debug_sync") ||
             !strcmp(name, "multi_check") || !strcmp(name, "multi_verify") || !strcmp(name, "multi_inject_fault") || !strcmp(name, "multi_dist_predict") || !strcmp(name, "multi_window") || !strcmp(name, "multi_timeout_s") || !strcmp(name, "multi_gemm_streamk") || !strcmp(name, "multi_leaf_cols")) return c->multi ? 0 : set_arg_err(2, "multi-device parameter on a single-device ctx");
    else return set_arg_err(2, "unknown parameter");
    return 0;
}

int32_t gp_ctx_get_param(gp_ctx* c, const char* name, int64_t* out) {
    if (!c || !reg_has(c)) return set_arg_err(1, "not a live gp_ctx");
    if (!name) return set_arg_err(2, "name is NULL");
    if (!out) return set_arg_err(3, "out is NULL");
    std::lock_guard<std::mutex> l(c->mu);
    if (c->multi && multi_get_param(c, name, out) == 0) return 0;
    const struct { const char* n; int64_t v; } tab[] = {
        {"nb", c->nb}, {"nb_small", c->nb_small}, {"nb_large", c->nb_large}, {"lookahead", c->lookahead}, {"lookahead_min_n", c->lookahead_min_n}, {"time_kernels", c->time_kernels},
        {"xcd_swizzle", c->xcd_swizzle}, {"xcd_min_tiles", c->xcd_min_tiles}, {"gemm_streamk", c->gemm_streamk},
        {"sk_max_tiles", c->sk_max_tiles}, {"sk_min_k", c->sk_min_k}, {"gemm_pipe", c->gemm_pipe}, {"gemm_pad_f32", c->gemm_pad_f32},
        {"gemm_pad_lds", c->gemm_pad_user ? c->gemm_pad_lds : 0}, {"trsv_nb", c->trsv_nb},  {"deterministic", c->deterministic},
        {"leaf_v2", c->leaf_v2}, {"leaf_xr", c->leaf_xr}, {"leaf_cols", c->leaf_cols}, {"updk_max_k", c->updk_max_k}, {"updk_rt", c->updk_rt},
        {"updk_tall_k", c->updk_tall_k}, {"updk_tall_m", c->updk_tall_m}, {"upd128", c->upd128}, {"leaf_group", c->leaf_group},
        {"ldpad", c->ldpad}, {"vfe_ks", c->vfe_ks}, {"vfe_sk", c->vfe_sk}, {"vfe_dual", c->vfe_dual}, {"vfe_inv_nb", c->vfe_inv_nb}, {"vfe_overlap", c->vfe_overlap}, {"vfe_chunk", c->vfe_chunk},
        {"kmat_rows", g_kmat_rows.load()}, {"dib_nb", c->dib_nb}, {"pool_cap_mb", (int64_t)(c->pool_cap >> 20)},
        {"pool_cached_mb", (int64_t)(c->pool_bytes >> 20)}, {"pool_blocks", (int64_t)c->pool.size()}};  // the last two are read-only
    for (const auto& e : tab)
        if (!strcmp(name, e.n)) {
            *out = e.v;
            return 0;
        }
    return set_arg_err(2, "unknown parameter");
}

int32_t gp_ctx_trim(gp_ctx* c) {
    Guard gd(c);
    if (!gd.ok) return set_arg_err(1, "not a live gp_ctx");
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->sm);
    (void)hipStreamSynchronize(c->sp);
    while (!c->pool.empty()) pool_drop(c, c->pool.size() - 1);
    if (c->multi) multi_trim(c);  // the rank contexts cache their own blocks (matrix pieces, operand buffers)
    return 0;
}

int32_t gp_get_timings(gp_ctx* c, gp_timings* out) {
    if (!c || !reg_has(c)) return set_arg_err(1, "not a live gp_ctx");
    if (!out) return set_arg_err(2, "out is NULL");
    std::lock_guard<std::mutex> l(c->mu);
    *out = c->tm;
    return 0;
}

int32_t gp_kernelmatrix(gp_ctx* c, const gp_kernel* k, const gp_points* x, const gp_points* y, void* out) {
    if (!c || !reg_has(c)) return set_arg_err(1, "not a live gp_ctx");
    RC(check_points(x, 3));
    RC(check_kernel(k, x->d, 2));
    if (y) {
        RC(check_points(y, 4));
        if (y->d != x->d) return set_arg_err(4, "x and y have different D");
    }
    if (!out) return set_arg_err(5, "out is NULL");
    std::lock_guard<std::mutex> l(c->mu);
    HIPCHK(hipSetDevice(c->device));
    auto run = [&](auto tag) -> int32_t {
        using T = decltype(tag);
        // device rows = y points (or x), device cols = x points: row-major (m×n) == column-major (n×m)
        const gp_points* rp = y ? y : x;
        const long n = x->n, m = rp->n, np = round_up(n, 128), mp = round_up(m, 128);
        const long ld = np + c->ldpad;
        std::vector<T> xc_h, xr_h;
        scale_points<T>(k, x, np, xc_h);
        if (y) scale_points<T>(k, y, mp, xr_h);
        void *xc_v = nullptr, *xr_v = nullptr, *K_v = nullptr;
        const size_t xcb = sizeof(T) * xc_h.size(), xrb = sizeof(T) * xr_h.size(), Kb = sizeof(T) * (size_t)mp * ld;
        DevBufs bufs(c);
        RC(bufs.get(xcb, &xc_v));
        if (y) RC(bufs.get(xrb, &xr_v));
        RC(bufs.get(Kb, &K_v));
        int32_t rc = [&]() -> int32_t {
            HIPCHK(hipMemcpyAsync(xc_v, xc_h.data(), xcb, hipMemcpyHostToDevice, c->sm));
            if (y) HIPCHK(hipMemcpyAsync(xr_v, xr_h.data(), xrb, hipMemcpyHostToDevice, c->sm));
            GridMap g = plain_map(0, 0, 0);
            dim3 grid((unsigned)(np / 128), (unsigned)(mp / 128));
            launch_kmat<T>(grid, c->sm, (T*)K_v, ld, (const T*)(y ? xr_v : xc_v),
                               y ? mp : np, (const T*)xc_v, np, x->d, k->kind, (T)k->variance, (const T*)nullptr, m, n,
                               0, g, (const T*)nullptr, (const T*)nullptr);
            HIPCHK(hipGetLastError());
            HIPCHK(hipMemcpy2DAsync(out, sizeof(T) * n, K_v, sizeof(T) * ld, sizeof(T) * n, m, hipMemcpyDeviceToHost,
                                    c->sm));
            HIPCHK(hipStreamSynchronize(c->sm));
            return 0;
        }();
        if (rc != 0) (void)hipStreamSynchronize(c->sm);
        return rc;
    };
    return k->dtype == 0 ? run(double()) : run(float());
}

static int32_t check_fit_args(gp_ctx* c, const gp_kernel* k, const gp_points* x, const gp_noise* noise) {
    if (!c || !reg_has(c)) return set_arg_err(1, "not a live gp_ctx");
    RC(check_points(x, 3));
    RC(check_kernel(k, x->d, 2));
    if (!noise) return set_arg_err(4, "noise is NULL");
    if (noise->kind != 0 && noise->kind != 1) return set_arg_err(4, "noise kind must be 0 or 1");
    if (noise->kind == 1 && !noise->diag) return set_arg_err(4, "noise diag is NULL");
    return 0;
}

// One (logpdf, posterior) pair on whatever the ctx drives: the 2D block-cyclic driver (multi.hip) for fp64 fits with at most
// 128 right-hand sides on a multi-device ctx, the single-device engine (devices[0] of a multi-device ctx) for everything else.
static int32_t fit_any(gp_ctx* c, const gp_kernel* k, const gp_points* x, const gp_noise* noise, const void* mean, const void* Y,
                       long ldy, int ncols, FitOut& fo, gp_post* p, void* alpha_out) {
    if (c->multi && k->dtype == 0 && ncols <= 128) {
        fo.logpdf.assign((size_t)ncols, 0.0);
        std::vector<double> terms((size_t)ncols + 1, 0.0);
        RC(multi_fit(c, k, x, noise, mean, Y, ldy, ncols, fo.logpdf.data(), terms.data(), p, alpha_out));
        fo.logdet = terms[0];
        fo.sqmahal.assign(terms.begin() + 1, terms.end());
        return 0;
    }
    return k->dtype == 0 ? fit_impl<double>(c, k, x, noise, mean, Y, ldy, ncols, fo, p, alpha_out)
                         : fit_impl<float>(c, k, x, noise, mean, Y, ldy, ncols, fo, p, alpha_out);
}

int32_t gp_logpdf(gp_ctx* c, const gp_kernel* k, const gp_points* x, const gp_noise* noise, const void* mean,
                  const void* Y, int64_t ldy, int32_t ncols, void* out) {
    RC(check_fit_args(c, k, x, noise));
    if (!Y) return set_arg_err(6, "Y is NULL");
    if (ncols < 1) return set_arg_err(8, "ncols must be >= 1");
    if (ldy < x->n) return set_arg_err(7, "ldy < n");
    if (!out) return set_arg_err(9, "out is NULL");
    std::lock_guard<std::mutex> l(c->mu);
    HIPCHK(hipSetDevice(c->device));
    FitOut fo;
    const int32_t rc = fit_any(c, k, x, noise, mean, Y, ldy, ncols, fo, nullptr, nullptr);
    if (rc != 0) return rc;
    for (int s = 0; s < ncols; ++s) {
        if (k->dtype == 0) ((double*)out)[s] = fo.logpdf[s];
        else ((float*)out)[s] = (float)fo.logpdf[s];
    }
    return 0;
}

// logdet(cov(fx)) and sqmahal(fx, Y) — the two terms logpdf adds up (src/finite_gp_projection.jl:306-311): `logdetcov` is not in
// the reference's API by that name but `sqmahal` is (src/finite_gp_projection.jl:313-326), and `gradlogpdf` (:328-337) is −α of
// the posterior fit.  Y may be NULL (ncols ignored): logdet only.  Outputs in the kernel's dtype; either may be NULL.
int32_t gp_logpdf_terms(gp_ctx* c, const gp_kernel* k, const gp_points* x, const gp_noise* noise, const void* mean,
                        const void* Y, int64_t ldy, int32_t ncols, void* logdet_out, void* sqmahal_out) {
    RC(check_fit_args(c, k, x, noise));
    if (Y) {
        if (ncols < 1) return set_arg_err(8, "ncols must be >= 1");
        if (ldy < x->n) return set_arg_err(7, "ldy < n");
    } else if (sqmahal_out) {
        return set_arg_err(6, "sqmahal needs Y");
    }
    std::lock_guard<std::mutex> l(c->mu);
    HIPCHK(hipSetDevice(c->device));
    FitOut fo;
    std::vector<char> zero;
    if (!Y) {
        // logdet only: the factorisation is all that is needed, but a multi-device fit verifies its result through the right-hand
        // side that rides along ((K + Σy) α = δ on every row) — an all-zero δ would make that check vacuous (α = 0 whatever the
        // factor holds).  A fixed pseudo-random probe in [−½, ½) rides instead; its sqmahal is discarded.
        zero.assign((size_t)x->n * (k->dtype == 0 ? 8 : 4), 0);
        for (int64_t i = 0; i < x->n; ++i) {
            const double v = (double)(((uint32_t)i * 2654435761u >> 8) & 0xffffu) / 65536.0 - 0.5 + 1.0 / 131072.0;
            if (k->dtype == 0) ((double*)zero.data())[i] = v;
            else ((float*)zero.data())[i] = (float)v;
        }
        ncols = 1;
        ldy = x->n;
    }
    const int32_t rc = fit_any(c, k, x, noise, Y ? mean : nullptr, Y ? Y : (const void*)zero.data(), ldy, ncols, fo, nullptr, nullptr);
    if (rc != 0) return rc;
    if (logdet_out) {
        if (k->dtype == 0) *(double*)logdet_out = fo.logdet;
        else *(float*)logdet_out = (float)fo.logdet;
    }
    if (sqmahal_out)
        for (int s = 0; s < ncols; ++s) {
            if (k->dtype == 0) ((double*)sqmahal_out)[s] = fo.sqmahal[s];
            else ((float*)sqmahal_out)[s] = (float)fo.sqmahal[s];
        }
    return 0;
}

// logdet(C) of a posterior's factor (C = cholesky(K + Σy), `post.data.C`): 2 Σ log L_ii kept from the fit.
int32_t gp_posterior_logdet(gp_post* post, double* out) {
    Guard gd(post);
    if (!gd.ok) return set_arg_err(1, "not a live gp_post");
    if (!out) return set_arg_err(2, "out is NULL");
    *out = 2.0 * post->logdet_half;
    return 0;
}

int32_t gp_posterior_fit(gp_ctx* c, const gp_kernel* k, const gp_points* x, const gp_noise* noise, const void* mean,
                         const void* y, gp_post** out, void* alpha_out, void* logpdf_out) {
    RC(check_fit_args(c, k, x, noise));
    if (!y) return set_arg_err(6, "y is NULL");
    if (!out) return set_arg_err(7, "out is NULL");
    *out = nullptr;
    std::lock_guard<std::mutex> l(c->mu);
    HIPCHK(hipSetDevice(c->device));
    gp_post* p = new gp_post();
    p->ctx = c;
    FitOut fo;
    const int32_t rc = fit_any(c, k, x, noise, mean, y, x->n, 1, fo, p, alpha_out);
    if (rc != 0) {
        delete p;
        return rc;
    }
    if (logpdf_out) {
        if (k->dtype == 0) *(double*)logpdf_out = fo.logpdf[0];
        else *(float*)logpdf_out = (float)fo.logpdf[0];
    }
    c->refs++;
    reg_add(p);
    *out = p;
    return 0;
}

int32_t gp_posterior_predict(gp_post* post, const gp_points* xs, const void* pm, int32_t what, void* mean_out,
                             void* var_out, void* cov_out) {
    Guard gd(post);
    if (!gd.ok) return set_arg_err(1, "not a live gp_post");
    RC(check_points(xs, 2));
    if (xs->d != post->d) return set_arg_err(2, "xs has a different D than the training inputs");
    if (what <= 0 || what > 7) return set_arg_err(4, "what must be a combination of 1|2|4");
    if ((what & 1) && !mean_out) return set_arg_err(5, "mean_out is NULL");
    if ((what & 2) && !var_out) return set_arg_err(6, "var_out is NULL");
    if ((what & 4) && !cov_out) return set_arg_err(7, "cov_out is NULL");
    gp_ctx* c = gd.c;
    HIPCHK(hipSetDevice(c->device));
    if ((what & 6) && multi_can_solve(post) && (!(what & 4) || xs->n <= 4096)) {
        // multi-device fit whose factor still lives as block-cyclic pieces: variances and (up to 4 096 test points) the full
        // covariance come from a forward solve ON the pieces (multi.hip: multi_predict_var) — no gather; the mean needs α only
        if (what & 1) RC(predict_impl<double>(post, xs, pm, 1, mean_out, nullptr, nullptr));
        gp_kernel k{};
        k.kind = post->kind; k.dtype = 0; k.variance = post->variance; k.nscale = post->nscale;
        k.scale = post->scale.empty() ? nullptr : post->scale.data();
        const long ns = xs->n;
        std::vector<double> xs_h;
        scale_points<double>(&k, xs, ns, xs_h);
        std::vector<double> sub((size_t)ns, 0.0), csub((what & 4) ? (size_t)ns * ns : 0, 0.0);
        RC(multi_predict_var(post, xs_h.data(), ns, ns, sub.data(), (what & 4) ? csub.data() : nullptr));
        if (what & 2)
            for (long i = 0; i < ns; ++i) ((double*)var_out)[i] = post->variance - sub[i];  // k** = σ² for the stationary kernels of the path
        if (what & 4) {  // K** on the device (the ctx's own stream), minus X Xᵀ
            const long nsp = round_up(ns, 128), ldc = nsp + c->ldpad;
            std::vector<double> xsp((size_t)post->d * nsp, 0.0);
            for (int dd = 0; dd < post->d; ++dd) memcpy(&xsp[(size_t)dd * nsp], &xs_h[(size_t)dd * ns], sizeof(double) * (size_t)ns);
            DevBufs bufs(c);
            void *x_v = nullptr, *C_v = nullptr;
            RC(bufs.get(sizeof(double) * xsp.size(), &x_v));
            RC(bufs.get(sizeof(double) * (size_t)nsp * ldc, &C_v));
            HIPCHK(hipMemcpyAsync(x_v, xsp.data(), sizeof(double) * xsp.size(), hipMemcpyHostToDevice, c->sm));
            RC(gpmi::eng_kcross(c, c->sm, post->kind, post->variance, (const double*)x_v, nsp, ns, nsp, (const double*)x_v, nsp, ns, nsp, post->d,
                                (double*)C_v, ldc));
            double* co = (double*)cov_out;
            HIPCHK(hipMemcpy2DAsync(co, sizeof(double) * ns, C_v, sizeof(double) * ldc, sizeof(double) * ns, ns, hipMemcpyDeviceToHost, c->sm));
            HIPCHK(hipStreamSynchronize(c->sm));
            for (size_t i = 0; i < (size_t)ns * ns; ++i) co[i] -= csub[i];
        }
        return 0;
    }
    if (what & 6) RC(multi_gather(post));  // multi-device fit: the factor is assembled on this device on first need
    return post->dtype == 0 ? predict_impl<double>(post, xs, pm, what, mean_out, var_out, cov_out)
                            : predict_impl<float>(post, xs, pm, what, mean_out, var_out, cov_out);
}

int32_t gp_logpdf_grad(gp_ctx* c, const gp_kernel* k, const gp_points* x, const gp_noise* noise, const void* mean, const void* y,
                       void* logpdf_out, double* dvar, double* dscale, void* dnoise, void* dy, void* dx) {
    RC(check_fit_args(c, k, x, noise));
    if (!y) return set_arg_err(6, "y is NULL");
    if (!logpdf_out) return set_arg_err(7, "logpdf_out is NULL");
    std::lock_guard<std::mutex> l(c->mu);
    HIPCHK(hipSetDevice(c->device));
    return k->dtype == 0 ? grad_impl<double>(c, k, x, noise, mean, y, logpdf_out, dvar, dscale, dnoise, dy, dx)
                         : grad_impl<float>(c, k, x, noise, mean, y, logpdf_out, dvar, dscale, dnoise, dy, dx);
}

int32_t gp_posterior_update(gp_post* old, const gp_points* x2, const gp_noise* noise2, const void* delta_all, gp_post** out,
                            void* alpha_out, void* logpdf_out) {
    Guard gd(old);
    if (!gd.ok) return set_arg_err(1, "not a live gp_post");
    RC(check_points(x2, 2));
    if (x2->d != old->d) return set_arg_err(2, "x2 has a different D than the training inputs");
    if (!noise2 || (noise2->kind != 0 && noise2->kind != 1) || (noise2->kind == 1 && !noise2->diag))
        return set_arg_err(3, "bad noise");
    if (!delta_all) return set_arg_err(4, "delta_all is NULL");
    if (!out) return set_arg_err(5, "out is NULL");
    *out = nullptr;
    gp_ctx* c = gd.c;
    HIPCHK(hipSetDevice(c->device));
    if (old->dtype == 0 && multi_can_solve(old) && x2->n <= 4096) {
        // multi-device posterior whose factor still lives as block-cyclic pieces: the factor is extended where it lives (multi.hip:
        // multi_update) — no gather; a failed forward / backward consistency check (-1991) falls back to the gathered path below
        gp_post* p = new gp_post();
        double lp = 0;
        const int32_t rc = multi_update(old, x2, noise2, delta_all, p, alpha_out, &lp);
        if (rc == 0) {
            if (logpdf_out) *(double*)logpdf_out = lp;
            c->refs++;
            reg_add(p);
            *out = p;
            return 0;
        }
        delete p;
        (void)hipSetDevice(c->device);
        if (rc != -1991) return rc;
    }
    RC(multi_gather(old));
    gp_post* p = new gp_post();
    double lp = 0;
    int32_t rc = old->dtype == 0 ? update_impl<double>(old, x2, noise2, delta_all, p, alpha_out, &lp)
                                 : update_impl<float>(old, x2, noise2, delta_all, p, alpha_out, &lp);
    if (rc != 0) {
        delete p;
        return rc;
    }
    if (logpdf_out) {
        if (old->dtype == 0) *(double*)logpdf_out = lp;
        else *(float*)logpdf_out = (float)lp;
    }
    c->refs++;
    reg_add(p);
    *out = p;
    return 0;
}

int32_t gp_posterior_factor_mul(gp_post* post, const void* xi, int32_t ncols, void* out) {
    Guard gd(post);
    if (!gd.ok) return set_arg_err(1, "not a live gp_post");
    if (!xi) return set_arg_err(2, "xi is NULL");
    if (ncols < 1) return set_arg_err(3, "ncols must be >= 1");
    if (!out) return set_arg_err(4, "out is NULL");
    gp_ctx* c = gd.c;
    HIPCHK(hipSetDevice(c->device));
    if (post->dtype == 0 && multi_can_solve(post) && ncols <= 1024) return multi_factor_mul(post, (const double*)xi, ncols, (double*)out);  // on the pieces
    RC(multi_gather(post));
    return post->dtype == 0 ? factor_mul_impl<double>(post, xi, ncols, out) : factor_mul_impl<float>(post, xi, ncols, out);
}

int32_t gp_posterior_solve(gp_post* post, const void* B, int32_t ncols, void* out) {
    Guard gd(post);
    if (!gd.ok) return set_arg_err(1, "not a live gp_post");
    if (!B) return set_arg_err(2, "B is NULL");
    if (ncols < 1) return set_arg_err(3, "ncols must be >= 1");
    if (!out) return set_arg_err(4, "out is NULL");
    gp_ctx* c = gd.c;
    HIPCHK(hipSetDevice(c->device));
    if (post->dtype == 0 && multi_can_solve(post) && ncols <= 128) return multi_solve(post, (const double*)B, ncols, (double*)out);  // on the pieces
    RC(multi_gather(post));
    return post->dtype == 0 ? solve_impl<double>(post, B, ncols, out) : solve_impl<float>(post, B, ncols, out);
}

int64_t gp_posterior_n(gp_post* post) {
    Guard gd(post);
    if (!gd.ok) return -1;
    return post->n;
}

int32_t gp_posterior_get_factor(gp_post* post, void* U_out) {
    Guard gd(post);
    if (!gd.ok) return set_arg_err(1, "not a live gp_post");
    if (!U_out) return set_arg_err(2, "U_out is NULL");
    gp_ctx* c = gd.c;
    HIPCHK(hipSetDevice(c->device));
    RC(multi_gather(post));
    const size_t es = post->dtype == 0 ? 8 : 4;
    const long n = post->n;
    // device row j (row-major lower L[j][0..j]) == host column j of the column-major upper U
    HIPCHK(hipMemcpy2DAsync(U_out, es * n, post->A, es * post->ld, es * n, n, hipMemcpyDeviceToHost, c->sm));
    HIPCHK(hipStreamSynchronize(c->sm));
    for (long j = 0; j < n; ++j) {  // strictly-lower part of U (i > j) is not part of the factor
        char* col = (char*)U_out + (size_t)j * n * es;
        if (j + 1 < n) memset(col + (size_t)(j + 1) * es, 0, (size_t)(n - j - 1) * es);
    }
    return 0;
}

int32_t gp_posterior_free(gp_post* post) {
    if (!post || !reg_take(post)) return set_arg_err(1, "not a live gp_post");
    gp_ctx* c = post->ctx;  // the handle holds a reference on its ctx: c is alive
    {
        std::lock_guard<std::mutex> l(c->mu);  // waits for any call still using the handle (Guard re-checks the registry under this lock)
        (void)hipSetDevice(c->device);
        multi_post_release(post);
        ctx_release(c, post->A, post->A_bytes);
        ctx_release(c, post->xs, post->xs_bytes);
        ctx_release(c, post->alpha, post->alpha_bytes);
        if (post->dibc.w) ctx_release(c, post->dibc.w, post->dibc.bytes);
        delete post;
    }
    ctx_unref(c);
    return 0;
}

// ---- VFE / DTC ---------------------------------------------------------------------------------------
int32_t gp_vfe_fit(gp_ctx* c, const gp_kernel* k, const gp_points* x, const gp_points* z, const gp_noise* noise,
                   double jitter, const void* mean, const void* y, int32_t approx, gp_vfe** out, void* objective_out) {
    RC(check_fit_args(c, k, x, noise));
    RC(check_points(z, 4));
    if (z->d != x->d) return set_arg_err(4, "z has a different D than x");
    if (!(jitter >= 0)) return set_arg_err(6, "jitter must be >= 0");
    if (!y) return set_arg_err(8, "y is NULL");
    if (approx != 0 && approx != 1) return set_arg_err(9, "approx must be 0 (VFE) or 1 (DTC)");
    if (out) *out = nullptr;
    std::lock_guard<std::mutex> l(c->mu);
    HIPCHK(hipSetDevice(c->device));
    gp_vfe* p = out ? new gp_vfe() : nullptr;
    if (p) p->ctx = c;
    double obj = 0;
    int32_t rc = k->dtype == 0 ? vfe_fit_impl<double>(c, k, x, z, noise, jitter, mean, y, approx, p, &obj)
                               : vfe_fit_impl<float>(c, k, x, z, noise, jitter, mean, y, approx, p, &obj);
    if (rc != 0) {
        delete p;
        return rc;
    }
    if (objective_out) {
        if (k->dtype == 0) *(double*)objective_out = obj;
        else *(float*)objective_out = (float)obj;
    }
    if (p) {
        c->refs++;
        reg_add(p);
        *out = p;
    }
    return 0;
}

int32_t gp_vfe_update(gp_vfe* old, const gp_points* x2, const gp_noise* noise2, const void* mean2, const void* y2, gp_vfe** out,
                      void* objective_out) {
    Guard gd(old);
    if (!gd.ok) return set_arg_err(1, "not a live gp_vfe");
    RC(check_points(x2, 2));
    if (x2->d != old->d) return set_arg_err(2, "x2 has a different D than the training inputs");
    if (!noise2 || (noise2->kind != 0 && noise2->kind != 1) || (noise2->kind == 1 && !noise2->diag))
        return set_arg_err(3, "bad noise");
    if (!y2) return set_arg_err(5, "y2 is NULL");
    if (!out) return set_arg_err(6, "out is NULL");
    *out = nullptr;
    gp_ctx* c = gd.c;
    HIPCHK(hipSetDevice(c->device));
    gp_kernel k{};
    k.kind = old->kind; k.dtype = old->dtype; k.variance = old->variance; k.nscale = old->nscale;
    k.scale = old->scale.empty() ? nullptr : old->scale.data();
    gp_vfe* p = new gp_vfe();
    p->ctx = c;
    double obj = 0;
    int32_t rc = old->dtype == 0
                     ? vfe_fit_impl<double>(c, &k, x2, nullptr, noise2, 0.0, mean2, y2, old->approx, p, &obj, old, VFE_UPDATE)
                     : vfe_fit_impl<float>(c, &k, x2, nullptr, noise2, 0.0, mean2, y2, old->approx, p, &obj, old, VFE_UPDATE);
    if (rc != 0) {
        delete p;
        return rc;
    }
    if (objective_out) {
        if (old->dtype == 0) *(double*)objective_out = obj;
        else *(float*)objective_out = (float)obj;
    }
    c->refs++;
    reg_add(p);
    *out = p;
    return 0;
}

int32_t gp_vfe_append(gp_vfe* old, const gp_points* z2, gp_vfe** out, void* objective_out) {
    Guard gd(old);
    if (!gd.ok) return set_arg_err(1, "not a live gp_vfe");
    RC(check_points(z2, 2));
    if (z2->d != old->d) return set_arg_err(2, "z2 has a different D than the pseudo-points");
    if (!out) return set_arg_err(3, "out is NULL");
    *out = nullptr;
    if (old->segs.empty()) return set_arg_err(1, "gp_vfe holds no observations");
    gp_ctx* c = gd.c;
    HIPCHK(hipSetDevice(c->device));
    gp_kernel k{};
    k.kind = old->kind; k.dtype = old->dtype; k.variance = old->variance; k.nscale = old->nscale;
    k.scale = old->scale.empty() ? nullptr : old->scale.data();
    gp_vfe* p = new gp_vfe();
    p->ctx = c;
    double obj = 0;
    int32_t rc = old->dtype == 0
                     ? vfe_fit_impl<double>(c, &k, nullptr, z2, nullptr, 0.0, nullptr, nullptr, old->approx, p, &obj, old, VFE_APPEND)
                     : vfe_fit_impl<float>(c, &k, nullptr, z2, nullptr, 0.0, nullptr, nullptr, old->approx, p, &obj, old, VFE_APPEND);
    if (rc != 0) {
        delete p;
        return rc;
    }
    if (objective_out) {
        if (old->dtype == 0) *(double*)objective_out = obj;
        else *(float*)objective_out = (float)obj;
    }
    c->refs++;
    reg_add(p);
    *out = p;
    return 0;
}

int32_t gp_vfe_predict(gp_vfe* p, const gp_points* xs, const void* pm, int32_t what, void* mean_out, void* var_out, void* cov_out) {
    Guard gd(p);
    if (!gd.ok) return set_arg_err(1, "not a live gp_vfe");
    RC(check_points(xs, 2));
    if (xs->d != p->d) return set_arg_err(2, "xs has a different D than the training inputs");
    if (what <= 0 || what > 7) return set_arg_err(4, "what must be a combination of 1|2|4");
    if ((what & 1) && !mean_out) return set_arg_err(5, "mean_out is NULL");
    if ((what & 2) && !var_out) return set_arg_err(6, "var_out is NULL");
    if ((what & 4) && !cov_out) return set_arg_err(7, "cov_out is NULL");
    gp_ctx* c = gd.c;
    HIPCHK(hipSetDevice(c->device));
    return p->dtype == 0 ? vfe_predict_impl<double>(p, xs, pm, what, mean_out, var_out, cov_out)
                         : vfe_predict_impl<float>(p, xs, pm, what, mean_out, var_out, cov_out);
}

static int32_t check_joint_args(const gp_points* xs, int d, const gp_noise* noise) {
    RC(check_points(xs, 2));
    if (xs->d != d) return set_arg_err(2, "xs has a different D than the training inputs");
    if (!noise || (noise->kind != 0 && noise->kind != 1) || (noise->kind == 1 && !noise->diag)) return set_arg_err(4, "bad noise");
    return 0;
}

int32_t gp_vfe_logpdf(gp_vfe* p, const gp_points* xs, const void* pm, const gp_noise* noise, const void* Y, int64_t ldy,
                      int32_t ncols, void* out) {
    Guard gd(p);
    if (!gd.ok) return set_arg_err(1, "not a live gp_vfe");
    RC(check_joint_args(xs, p->d, noise));
    if (!Y) return set_arg_err(5, "Y is NULL");
    if (ldy < xs->n) return set_arg_err(6, "ldy < n");
    if (ncols < 1) return set_arg_err(7, "ncols must be >= 1");
    if (!out) return set_arg_err(8, "out is NULL");
    gp_ctx* c = gd.c;
    HIPCHK(hipSetDevice(c->device));
    DevBufs bufs(c);
    Joint<double> J;
    const long R = round_up(ncols, 128);
    int32_t rc = p->dtype == 0 ? vfe_joint<double>(p, xs, pm, noise, R, bufs, J) : vfe_joint<float>(p, xs, pm, noise, R, bufs, J);
    if (rc == 0)
        rc = p->dtype == 0 ? joint_logpdf<double, double>(c, J, Y, ldy, ncols, out) : joint_logpdf<double, float>(c, J, Y, ldy, ncols, out);
    if (rc < 0) (void)hipStreamSynchronize(c->sm);
    return rc;
}

int32_t gp_vfe_rand(gp_vfe* p, const gp_points* xs, const void* pm, const gp_noise* noise, const void* xi, int32_t ncols, void* out) {
    Guard gd(p);
    if (!gd.ok) return set_arg_err(1, "not a live gp_vfe");
    RC(check_joint_args(xs, p->d, noise));
    if (!xi) return set_arg_err(5, "xi is NULL");
    if (ncols < 1) return set_arg_err(6, "ncols must be >= 1");
    if (!out) return set_arg_err(7, "out is NULL");
    gp_ctx* c = gd.c;
    HIPCHK(hipSetDevice(c->device));
    DevBufs bufs(c);
    Joint<double> J;
    int32_t rc = p->dtype == 0 ? vfe_joint<double>(p, xs, pm, noise, 0, bufs, J) : vfe_joint<float>(p, xs, pm, noise, 0, bufs, J);
    if (rc == 0)
        rc = p->dtype == 0 ? joint_rand<double, double>(c, J, bufs, xi, ncols, out) : joint_rand<double, float>(c, J, bufs, xi, ncols, out);
    if (rc < 0) (void)hipStreamSynchronize(c->sm);
    return rc;
}

int32_t gp_vfe_get(gp_vfe* p, void* alpha_out, void* meps_out) {
    Guard gd(p);
    if (!gd.ok) return set_arg_err(1, "not a live gp_vfe");
    gp_ctx* c = gd.c;
    HIPCHK(hipSetDevice(c->device));
    std::vector<double> h((size_t)p->mp * 3);
    HIPCHK(hipMemcpy(h.data(), p->vec, sizeof(double) * h.size(), hipMemcpyDeviceToHost));
    for (long i = 0; i < p->m; ++i) {
        if (alpha_out) {
            if (p->dtype == 0) ((double*)alpha_out)[i] = h[2 * p->mp + i];
            else ((float*)alpha_out)[i] = (float)h[2 * p->mp + i];
        }
        if (meps_out) {
            if (p->dtype == 0) ((double*)meps_out)[i] = h[p->mp + i];
            else ((float*)meps_out)[i] = (float)h[p->mp + i];
        }
    }
    return 0;
}

int32_t gp_vfe_get_factors(gp_vfe* p, void* U_out, void* LamU_out) {
    Guard gd(p);
    if (!gd.ok) return set_arg_err(1, "not a live gp_vfe");
    gp_ctx* c = gd.c;
    HIPCHK(hipSetDevice(c->device));
    const long m = p->m;
    // device row j of the row-major lower factor == host column j of the column-major upper factor (always fp64 on the device)
    std::vector<double> h;
    const void* srcs[2] = {p->Lz, p->Ld};
    void* dsts[2] = {U_out, LamU_out};
    for (int which = 0; which < 2; ++which) {
        if (!dsts[which]) continue;
        if (!srcs[which]) return set_arg_err(1, "the posterior holds no factor (objective-only fit)");
        h.resize((size_t)m * (size_t)m);
        HIPCHK(hipMemcpy2DAsync(h.data(), sizeof(double) * m, srcs[which], sizeof(double) * p->ld, sizeof(double) * m, m,
                                hipMemcpyDeviceToHost, c->sm));
        HIPCHK(hipStreamSynchronize(c->sm));
        for (long j = 0; j < m; ++j)
            for (long i = 0; i < m; ++i) {
                const double v = i <= j ? h[(size_t)j * m + i] : 0.0;
                if (p->dtype == 0) ((double*)dsts[which])[(size_t)j * m + i] = v;
                else ((float*)dsts[which])[(size_t)j * m + i] = (float)v;
            }
    }
    return 0;
}

int64_t gp_vfe_n(gp_vfe* p) {
    Guard gd(p);
    if (!gd.ok) return -1;
    return p->n_obs;
}

int32_t gp_vfe_get_by(gp_vfe* p, void* by_out) {
    Guard gd(p);
    if (!gd.ok) return set_arg_err(1, "not a live gp_vfe");
    if (!by_out) return set_arg_err(2, "b_y_out is NULL");
    gp_ctx* c = gd.c;
    HIPCHK(hipSetDevice(c->device));
    const size_t es = p->dtype == 0 ? 8 : 4;
    size_t off = 0;
    for (const auto& sg : p->segs) {  // one segment per fit / update, in arrival order
        if (sg->n > 0) HIPCHK(hipMemcpy((char*)by_out + off, sg->b, es * (size_t)sg->n, hipMemcpyDeviceToHost));
        off += es * (size_t)sg->n;
    }
    return 0;
}

int32_t gp_vfe_grad(gp_vfe* p, double* dvariance, double* dscale, double* dnoise_sum, void* dnoise_diag, void* dy, double* dz, int32_t z_layout,
                    void* dx, int32_t x_layout) {
    Guard gd(p);
    if (!gd.ok) return set_arg_err(1, "not a live gp_vfe");
    if (dz && p->dtype != 0)
        return set_arg_err(7, "pseudo-input gradients need an fp64 handle: the fp32-streamed B Bᵀ of an fp32 fit does not carry the cancellation between the K_zz and K_fz terms");
    if (dz && (z_layout < 0 || z_layout > 2 || (z_layout == 0 && p->d != 1))) return set_arg_err(8, "z_layout must be 0 (vector, D = 1), 1 (ColVecs) or 2 (RowVecs)");
    if (dx && (x_layout < 0 || x_layout > 2 || (x_layout == 0 && p->d != 1))) return set_arg_err(10, "x_layout must be 0 (vector, D = 1), 1 (ColVecs) or 2 (RowVecs)");
    gp_ctx* c = gd.c;
    HIPCHK(hipSetDevice(c->device));
    return p->dtype == 0 ? vfe_grad_impl<double>(p, dvariance, dscale, dnoise_sum, dnoise_diag, dy, dz, z_layout, dx, x_layout)
                         : vfe_grad_impl<float>(p, dvariance, dscale, dnoise_sum, dnoise_diag, dy, dz, z_layout, dx, x_layout);
}

int64_t gp_vfe_m(gp_vfe* p) {
    Guard gd(p);
    if (!gd.ok) return -1;
    return p->m;
}

int32_t gp_vfe_free(gp_vfe* p) {
    if (!p || !reg_take(p)) return set_arg_err(1, "not a live gp_vfe");
    gp_ctx* c = p->ctx;
    {
        std::lock_guard<std::mutex> l(c->mu);
        (void)hipSetDevice(c->device);
        vfe_release(p);
        delete p;
    }
    ctx_unref(c);
    return 0;
}

// ---- joint predictive logpdf / rand of an exact posterior --------------------------------------------------
int32_t gp_posterior_logpdf(gp_post* post, const gp_points* xs, const void* pm, const gp_noise* noise, const void* Y, int64_t ldy,
                            int32_t ncols, void* out) {
    Guard gd(post);
    if (!gd.ok) return set_arg_err(1, "not a live gp_post");
    RC(check_joint_args(xs, post->d, noise));
    if (!Y) return set_arg_err(5, "Y is NULL");
    if (ldy < xs->n) return set_arg_err(6, "ldy < n");
    if (ncols < 1) return set_arg_err(7, "ncols must be >= 1");
    if (!out) return set_arg_err(8, "out is NULL");
    gp_ctx* c = gd.c;
    HIPCHK(hipSetDevice(c->device));
    const bool dist = multi_can_solve(post) && xs->n <= 4096;  // held-out logpdf on the block-cyclic pieces: no gather
    if (!dist) RC(multi_gather(post));
    DevBufs bufs(c);
    const long R = round_up(ncols, 128);
    int32_t rc;
    if (post->dtype == 0) {
        Joint<double> J;
        rc = dist ? post_joint_dist(post, xs, pm, noise, R, bufs, J) : post_joint<double>(post, xs, pm, noise, R, bufs, J);
        if (rc == 0) rc = joint_logpdf<double, double>(c, J, Y, ldy, ncols, out);
    } else {
        Joint<float> J;
        rc = post_joint<float>(post, xs, pm, noise, R, bufs, J);
        if (rc == 0) rc = joint_logpdf<float, float>(c, J, Y, ldy, ncols, out);
    }
    if (rc < 0) (void)hipStreamSynchronize(c->sm);
    return rc;
}

int32_t gp_posterior_rand(gp_post* post, const gp_points* xs, const void* pm, const gp_noise* noise, const void* xi, int32_t ncols,
                          void* out) {
    Guard gd(post);
    if (!gd.ok) return set_arg_err(1, "not a live gp_post");
    RC(check_joint_args(xs, post->d, noise));
    if (!xi) return set_arg_err(5, "xi is NULL");
    if (ncols < 1) return set_arg_err(6, "ncols must be >= 1");
    if (!out) return set_arg_err(7, "out is NULL");
    gp_ctx* c = gd.c;
    HIPCHK(hipSetDevice(c->device));
    const bool dist = multi_can_solve(post) && xs->n <= 4096;  // posterior sampling on the block-cyclic pieces: no gather
    if (!dist) RC(multi_gather(post));
    DevBufs bufs(c);
    int32_t rc;
    if (post->dtype == 0) {
        Joint<double> J;
        rc = dist ? post_joint_dist(post, xs, pm, noise, 0, bufs, J) : post_joint<double>(post, xs, pm, noise, 0, bufs, J);
        if (rc == 0) rc = joint_rand<double, double>(c, J, bufs, xi, ncols, out);
    } else {
        Joint<float> J;
        rc = post_joint<float>(post, xs, pm, noise, 0, bufs, J);
        if (rc == 0) rc = joint_rand<float, float>(c, J, bufs, xi, ncols, out);
    }
    if (rc < 0) (void)hipStreamSynchronize(c->sm);
    return rc;
}

// ---- microbenchmarks / probes (tools/gpu_diag.py) ------------------------------------------------
int32_t gp_probe_mfma_f64(gp_ctx* c, const double* A_host, const double* B_host, double* D_host) {
    if (!c || !reg_has(c)) return set_arg_err(1, "not a live gp_ctx");
    std::lock_guard<std::mutex> l(c->mu);
    HIPCHK(hipSetDevice(c->device));
    double* buf;
    HIPCHK(hipMalloc((void**)&buf, sizeof(double) * (64 + 64 + 256)));
    HIPCHK(hipMemcpy(buf, A_host, sizeof(double) * 64, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(buf + 64, B_host, sizeof(double) * 64, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(mfma_probe_f64_kernel, dim3(1), dim3(64), 0, c->sm, buf, buf + 64, buf + 128);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(c->sm));
    HIPCHK(hipMemcpy(D_host, buf + 128, sizeof(double) * 256, hipMemcpyDeviceToHost));
    HIPCHK(hipFree(buf));
    return 0;
}
// returns measured TFLOP/s of back-to-back v_mfma_f64_16x16x4_f64 (all CUs, 2 blocks per CU)
int32_t gp_bench_mfma_f64(gp_ctx* c, int32_t iters, double* tflops_out) {
    if (!c || !reg_has(c)) return set_arg_err(1, "not a live gp_ctx");
    std::lock_guard<std::mutex> l(c->mu);
    HIPCHK(hipSetDevice(c->device));
    double* buf;
    HIPCHK(hipMalloc((void**)&buf, 64));
    hipEvent_t a, b;
    HIPCHK(hipEventCreate(&a));
    HIPCHK(hipEventCreate(&b));
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, c->device));
    const int blocks = prop.multiProcessorCount;  // one 1024-thread block per CU
    const size_t smem = 96 * 1024;
    HIPCHK(hipFuncSetAttribute((const void*)mfma_rate_f64_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    hipLaunchKernelGGL(mfma_rate_f64_kernel, dim3(blocks), dim3(1024), smem, c->sm, buf, 64);
    HIPCHK(hipEventRecord(a, c->sm));
    hipLaunchKernelGGL(mfma_rate_f64_kernel, dim3(blocks), dim3(1024), smem, c->sm, buf, iters);
    HIPCHK(hipEventRecord(b, c->sm));
    HIPCHK(hipStreamSynchronize(c->sm));
    float ms;
    HIPCHK(hipEventElapsedTime(&ms, a, b));
    const double flops = (double)blocks * 16.0 * (double)iters * 4.0 * 2.0 * 16 * 16 * 4;
    *tflops_out = flops / (ms * 1e-3) / 1e12;
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(b);
    HIPCHK(hipFree(buf));
    return 0;
}

// measured TFLOP/s of back-to-back fp32 MFMAs: variant 0 = v_mfma_f32_16x16x4_f32 (what the GEMM kernels issue), 1 = 32x32x2
int32_t gp_bench_mfma_f32(gp_ctx* c, int32_t variant, int32_t iters, double* tflops_out) {
    if (!c || !reg_has(c)) return set_arg_err(1, "not a live gp_ctx");
    std::lock_guard<std::mutex> l(c->mu);
    HIPCHK(hipSetDevice(c->device));
    float* buf;
    HIPCHK(hipMalloc((void**)&buf, 64));
    hipEvent_t a, b;
    HIPCHK(hipEventCreate(&a));
    HIPCHK(hipEventCreate(&b));
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, c->device));
    const int blocks = prop.multiProcessorCount;
    const size_t smem = 96 * 1024;
    auto run = [&](auto kern) -> int32_t {
        HIPCHK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(1024), smem, c->sm, buf, 64);
        HIPCHK(hipEventRecord(a, c->sm));
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(1024), smem, c->sm, buf, iters);
        HIPCHK(hipEventRecord(b, c->sm));
        HIPCHK(hipStreamSynchronize(c->sm));
        return 0;
    };
    RC(variant == 0 ? run(mfma_rate_f32_kernel<0>) : run(mfma_rate_f32_kernel<1>));
    float ms;
    HIPCHK(hipEventElapsedTime(&ms, a, b));
    const double per = variant == 0 ? 2.0 * 16 * 16 * 4 : 2.0 * 32 * 32 * 2;
    *tflops_out = (double)blocks * 16.0 * (double)iters * 4.0 * per / (ms * 1e-3) / 1e12;
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(b);
    HIPCHK(hipFree(buf));
    return 0;
}

// ---- device-level building blocks (multi-process block-cyclic driver) ------------------------------
static GridMap to_map(const gp_grid* g, long row0, long col0) {
    GridMap m = plain_map(0, row0, col0);
    if (g) {
        m.lower = g->lower;
        m.P = g->P; m.p = g->p; m.Q = g->Q; m.q = g->q;
        m.nb = (long)g->tb * 128;
    }
    return m;
}

int32_t gpd_assemble(gp_ctx* c, const gp_kernel* k, const double* x_dev, int64_t n_valid, int64_t n_pad, int32_t d,
                     const double* noise_dev, const gp_grid* g, double* a_loc, int64_t lda, int64_t m_loc,
                     int64_t n_loc) {
    if (!c || !reg_has(c)) return set_arg_err(1, "not a live gp_ctx");
    RC(check_kernel(k, d, 2));
    if (m_loc % 128 || n_loc % 128) return set_arg_err(11, "m_loc, n_loc must be multiples of 128");
    std::lock_guard<std::mutex> l(c->mu);
    HIPCHK(hipSetDevice(c->device));
    GridMap m = to_map(g, 0, 0);
    dim3 grid((unsigned)(n_loc / 128), (unsigned)(m_loc / 128));
    if (grid.x == 0 || grid.y == 0) return 0;
    launch_kmat<double>(grid, c->sm, a_loc, lda, x_dev, n_pad, x_dev, n_pad, d,
                       k->kind, k->variance, noise_dev, n_valid, n_valid, 1, m, (const double*)nullptr, (const double*)nullptr);
    HIPCHK(hipGetLastError());
    return 0;
}

int32_t gpd_potrf(gp_ctx* c, double* a, int64_t lda, int64_t m, int64_t n, int32_t* info_dev, int32_t col0,
                  int64_t n_valid, double* logdet_dev) {
    if (!c || !reg_has(c)) return set_arg_err(1, "not a live gp_ctx");
    if (n % 64 || m % 64 || m < n) return set_arg_err(4, "m, n must be multiples of 64 with m >= n");
    // the register-resident leaf and the in-panel update kernel move rows as 16-byte pieces (leaf.hpp ld4 / st4)
    if (((uintptr_t)a & 15) != 0) return set_arg_err(2, "a must be 16-byte aligned");
    if (lda % 2 != 0 || lda < n) return set_arg_err(3, "lda must be even and >= n");
    std::lock_guard<std::mutex> l(c->mu);
    HIPCHK(hipSetDevice(c->device));
    return potrf_rec<double>(c, c->sm, a, lda, 0, n, m, info_dev, col0, n_valid, logdet_dev);
}

int32_t gpd_trsm(gp_ctx* c, double* x, int64_t ldx, int64_t m, const double* lmat, int64_t ldl, int64_t n) {
    if (!c || !reg_has(c)) return set_arg_err(1, "not a live gp_ctx");
    if (n % 64 || m % 64) return set_arg_err(4, "m, n must be multiples of 64");
    if (m == 0) return 0;
    std::lock_guard<std::mutex> l(c->mu);
    HIPCHK(hipSetDevice(c->device));
    return trsm_rec<double>(c, c->sm, x, ldx, m, lmat, ldl, n);
}

int32_t gpd_inv_lower(gp_ctx* c, const double* lmat, int64_t ldl, int64_t nb, double* w, int64_t ldw, double* scratch1, double* scratch2) {
    if (!c || !reg_has(c)) return set_arg_err(1, "not a live gp_ctx");
    if (nb < 64 || nb % 64) return set_arg_err(4, "nb must be a multiple of 64");
    if (!lmat || !w || !scratch1) return set_arg_err(2, "l / w / scratch1 is NULL");
    std::lock_guard<std::mutex> l(c->mu);
    HIPCHK(hipSetDevice(c->device));
    return gpmi::eng_inv_lower(c, c->sm, lmat, ldl, nb, w, ldw, scratch1, scratch2);
}
int32_t gpd_trsm_inv(gp_ctx* c, double* x, int64_t ldx, int64_t m, const double* w, int64_t ldw, int64_t nb, double* scratch, int64_t lds) {
    if (!c || !reg_has(c)) return set_arg_err(1, "not a live gp_ctx");
    if (nb % 64 || m % 64) return set_arg_err(3, "m, nb must be multiples of 64");
    if (m == 0) return 0;
    std::lock_guard<std::mutex> l(c->mu);
    HIPCHK(hipSetDevice(c->device));
    return gpmi::eng_trsm_inv(c, c->sm, x, ldx, m, w, ldw, nb, scratch, lds);
}

int32_t gpd_gemm_nt(gp_ctx* c, double* cm, int64_t ldc, const double* a, int64_t lda, const double* b, int64_t ldb,
                    int64_t m, int64_t n, int64_t k, const gp_grid* g, int64_t row0, int64_t col0) {
    if (!c || !reg_has(c)) return set_arg_err(1, "not a live gp_ctx");
    if (m % 64 || n % 64 || k % 16) return set_arg_err(8, "m, n multiples of 64 and k multiple of 16 required");
    std::lock_guard<std::mutex> l(c->mu);
    HIPCHK(hipSetDevice(c->device));
    return launch_gemm<double>(c, c->sm, cm, ldc, a, lda, b, ldb, m, n, k, to_map(g, row0, col0));
}

int32_t gpd_trsv(gp_ctx* c, const double* lmat, int64_t ldl, int64_t np, double* r, int64_t ldr, int32_t nrhs,
                 int32_t forward) {
    if (!c || !reg_has(c)) return set_arg_err(1, "not a live gp_ctx");
    if (np % 128) return set_arg_err(4, "np must be a multiple of 128");
    std::lock_guard<std::mutex> l(c->mu);
    HIPCHK(hipSetDevice(c->device));
    return trsv<double>(c, c->sm, lmat, ldl, np, r, ldr, nrhs, forward != 0);
}

int32_t gpd_gemv_t(gp_ctx* c, const double* lmat, int64_t ldl, int64_t nrows, int64_t ncols, const double* a, double* r) {
    if (!c || !reg_has(c)) return set_arg_err(1, "not a live gp_ctx");
    if (nrows <= 0 || ncols <= 0) return 0;
    std::lock_guard<std::mutex> l(c->mu);
    HIPCHK(hipSetDevice(c->device));
    hipLaunchKernelGGL(gemv_t_kernel<double>, dim3((unsigned)((ncols + 255) / 256), (unsigned)((nrows + 63) / 64)), dim3(256),
                       0, c->sm, lmat, ldl, nrows, ncols, a, r);
    HIPCHK(hipGetLastError());
    return 0;
}

int32_t gpd_rowsumsq(gp_ctx* c, const double* x, int64_t ldx, int64_t nrows, int64_t ncols, double* out_dev) {
    if (!c || !reg_has(c)) return set_arg_err(1, "not a live gp_ctx");
    if (nrows <= 0) return 0;
    std::lock_guard<std::mutex> l(c->mu);
    HIPCHK(hipSetDevice(c->device));
    hipLaunchKernelGGL(rowsumsq_kernel<double>, dim3((unsigned)nrows), dim3(256), 0, c->sm, x, ldx, ncols, out_dev);
    HIPCHK(hipGetLastError());
    return 0;
}

int32_t gpd_gemm_time(gp_ctx* c, double* ms_out, int64_t* launches_out) {
    if (!c || !reg_has(c)) return set_arg_err(1, "not a live gp_ctx");
    std::lock_guard<std::mutex> l(c->mu);
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->sm));
    double tot = 0;
    for (auto& r : c->gemm_recs) {
        float ms = 0;
        HIPCHK(hipEventElapsedTime(&ms, r.a, r.b));
        tot += ms;
    }
    if (ms_out) *ms_out = tot;
    if (launches_out) *launches_out = (int64_t)c->gemm_recs.size();
    c->gemm_recs.clear();
    c->ev_used = 0;
    return 0;
}

int32_t gpd_sync(gp_ctx* c) {
    if (!c || !reg_has(c)) return set_arg_err(1, "not a live gp_ctx");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->sm));
    HIPCHK(hipStreamSynchronize(c->sp));
    return 0;
}

}  // extern "C"
