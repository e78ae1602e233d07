// engine.hpp — internal interface shared by the translation units of libgpmi355.so:
//   gpmi355.hip  kernels (kernels.hpp), the single-device engine and the C ABI
//   multi.hip    the multi-device 2D block-cyclic driver (host code only: RCCL / peer copies + calls into the engine)
// Nothing here is part of the public ABI (include/gpmi355.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <array>
#include <atomic>
#include <mutex>
#include <set>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/gpmi355.h"

namespace gpmi {
// 2D block-cyclic bookkeeping shared by kmat and gemm: local absolute index -> global index.
struct GridMap {
    int lower;       // 1: skip tiles strictly above the global diagonal
    int P, p, Q, q;  // process grid / my coordinates (1,0,1,0 on a single GPU)
    long nb;         // distribution block in elements (multiple of 128); ignored when P=Q=1
    long row0, col0; // local absolute index of the region's first row / column
    int compact;     // 1: 1-D grid enumerating only the tiles on/below the diagonal (single-GPU lower mode)
                     // 2 / 3: XCD-aware super-tile order (rectangular / lower trapezoid), see xcd_tile()
    int tn, dt;      // compact: number of tile columns, diagonal offset in tiles (row tile i has min(tn, i+dt+1) tiles)
    int tm;          // number of tile rows (modes 2, 3)
    int beta0;       // 1: C is overwritten with −A·Bᵀ (no preload of C)
    int ktri;        // 1: A is lower triangular (M×M, K = M): the k loop of row tile m0 stops at column m0 + 128
                     // 3: B is lower triangular (N×N, K = N): the k loop of COLUMN tile n0 stops at column n0 + 128 (column tiles rotated by the row tile)
                     // 2: A is upper triangular from row ktri_off on: the k loop of row tile m0 starts at column max(0, m0 − ktri_off)
    int nbatch;      // > 1: blockIdx.z = b selects an independent product over the k range [b·K, (b+1)·K) of A and B,
    long cstride;    //      written to C + b·cstride (split-K partial products of one SYRK, summed by the caller)
    long astride, bstride;  // != 0 with nbatch > 1: independent products instead — A + b·astride, B + b·bstride, the full k range each
    int ktri_off;    // ktri == 1 with A pointing at row ktri_off of the triangular matrix: row tile m0 stops at ktri_off + m0 + 128
                     // ktri == 2: rows [0, ktri_off) of A are dense, the upper-triangular block starts at row ktri_off (row tile m0 > ktri_off starts at column m0 − ktri_off)
};
}  // namespace gpmi

// ---- errors (thread-local text behind gp_last_error) ---------------------------------------------
int32_t set_hip_err(hipError_t e, const char* what, int line);
int32_t set_arg_err(int i, const char* msg);
int32_t set_err_text(int32_t status, const std::string& msg);  // returns status
#define HIPCHK(expr)                                                   \
    do {                                                               \
        hipError_t e_ = (expr);                                        \
        if (e_ != hipSuccess) return set_hip_err(e_, #expr, __LINE__); \
    } while (0)
#define RC(expr)                \
    do {                        \
        int32_t rc_ = (expr);   \
        if (rc_ != 0) return rc_; \
    } while (0)

static inline long round_up(long v, long m) { return (v + m - 1) / m * m; }
static const double LOG2PI = 1.8378770664093454835606594728112;

// ------------------------------------------------------------------------------------------------
// handles
// ------------------------------------------------------------------------------------------------
struct FreeBlock {
    void* p;
    size_t bytes;
};

struct gp_ctx {
    int device = 0;
    hipStream_t sm = nullptr;  // main stream (trailing updates, assembly, solves)
    hipStream_t sp = nullptr;  // panel stream (look-ahead)
    void* pin = nullptr;       // page-locked staging for N-long host vectors of the sparse fit (ctx_pinned); grows on demand, freed with the ctx
    size_t pin_bytes = 0;
    hipStream_t sq = nullptr;  // third stream (high priority), created on first use (VFE: the next chunk's triangular product beside the chunk SYRK — "vfe_dual")
    bool own_sm = false;
    std::mutex mu;
    long nb = -1;          // outer panel width: −1 automatic (nb_small below lookahead_min_n — one-stream schedule, wider panels halve the passes over the
                           // trailing matrix — nb_large from there on), 0 purely recursive, > 0 that width at every size
    long nb_small = 4096, nb_large = 2048;
    int lookahead = 1;
    long lookahead_min_n = 24576;  // the look-ahead pays from here on (N <= 16 384: 0.5-3 % slower with it since the register-resident leaf; measured round 4)
    int time_kernels = 0;
    int sched = 0;         // 0: whole panel on the panel stream (look-ahead); 1: diag-first, all-MFMA rows_below
    long trsv_nb = 256;    // diagonal block of the vector solves handled by one workgroup (the rest goes to the multi-CU update kernels)
    long leaf_group = 128; // columns factored left-looking by consecutive leaves (64 = every leaf followed by its own GEMM)
    int deterministic = 0; // 1: no floating-point atomics in the exact path (no stream-K tails, one thread per column in the backward sweep): bitwise repeatable
    int leaf_v2 = 1;       // fp64 leaves by panel64v2_kernel (register-resident leaf, round 4); 0: panel64_kernel
    int leaf_xr = 0;       // rows of X per leaf workgroup: 64 / 128; 0 = 64 while that gives at most two workgroups per CU, else 128
    int upd128 = 1;        // the K = N = 128 in-panel update between two 128-column leaves by panel_updk_kernel (leaf.hpp) instead of the tile GEMM
    long updk_max_k = 512;       // in-panel updates with 256 <= K = N <= this run through panel_updk_kernel (0: tile GEMM) ...
    long updk_tall_k = 256;      // ... K above this only while at most updk_tall_m rows are below (the stream-K tile GEMM wins on tall K = 512 launches)
    long updk_tall_m = 8192;
    int updk_rt = 0;             // rows per workgroup / 16 of panel_updk_kernel (0 auto, 4, 2, 1)
    int leaf_cols = 128;   // columns per register-resident leaf launch: 128 (one launch per 128-column group, no in-leaf pre-update) or 64
    int gemm_streamk = 1;  // persistent-grid GEMM with a stream-K tail for single-GPU maps (gemm_nt_sk_kernel) on launches of at
                           // most sk_max_tiles tiles: the few-tile in-panel GEMMs are cut along k over all CUs (−2…5 % at N <= 32 768)
    int num_cus = 256;
    long sk_max_tiles = 4096;  // stream-K only for launches of at most this many tiles (8 rounds): the persistent kernel is
    long sk_min_k = 0;     // launches with a shorter k range take the plain tile kernel (experiment knob)
                           // ~5 % slower than hardware dispatch on large launches, where the tail does not matter anyway
    int sk_scope = 0;      // > 0 inside single-stream entry points (predict / update / gradient): stream-K GEMM tails pay there
                           // (inside the factorisation the look-ahead stream already fills the tail of every trailing update)
    long gemm_pad_lds = 0; // extra dynamic LDS per GEMM workgroup: 20480 limits residency to ONE workgroup per CU (same speed —
                           // tools/overlap_probe.hip — and leaves room for concurrently running RCCL / copy kernels)
    long gemm_pad_f32 = 0;  // the fp32 default of the above: 0 = two workgroups per CU (1 % faster at C5 since the k loop is pipelined — 81.3 against 82.3 ms,
                            // two A/B pairs on one box, profiles/r5/sweep3.jsonl; with the round-2 loop one workgroup per CU was 5 % faster: 20480 restores it)
    int gemm_pipe = 1;     // k loop of the MFMA GEMMs software-pipelined across the step boundary (kernels.hpp gemm_kloop_pipe; 0: the round-2 loop)
    long dib_nb = 2048;    // forward solves X L⁻ᵀ against a RESIDENT factor (predictions, covariances, sequential updates, the gradient's L⁻ᵀ): sub-blocks of at
                           // most this many columns are solved by ONE triangular-k GEMM with the explicit inverse of the diagonal block (0: the recursion down to 64)
    bool gemm_pad_set = false;
    bool gemm_pad_user = false;  // "gemm_pad_lds" was set explicitly (otherwise: 0 for fp64, 20480 for fp32)
    long vfe_ks = 2048;    // VFE fp32: data points per fp32 partial product of the chunk SYRK (fp64 sums across partials)
    int vfe_overlap = 1;   // VFE: kmat / ystats / partial-sum adds on the second stream beside the chunk GEMMs (double buffers)
    int vfe_sk = 0;        // VFE: stream-K GEMM tails for the M×M side (K_zz / Λ_ε factorisations, inv(L_z))
    int vfe_dual = 0;      // VFE experiment switch (round 6, measured and left OFF): the triangular products Y(c) on a high-priority third stream beside the chunk SYRKs
                           // on the main stream, so that each launch's last partial round of workgroups is filled by the other — C5 +0.4…0.7 ms with the priority,
                           // −0.45 ms with both at equal priority (profiles/r6/c5_ab*.jsonl): the single-stream pass has no idle tail to fill
    long vfe_inv_nb = 512; // VFE prelude: inv(L_z) with the inverse diagonal blocks of this width built in one batched launch sequence (0: 64-wide leaves)
    int xcd_swizzle = 0;   // XCD-aware super-tile order of the MFMA gemm workgroups
    long xcd_min_tiles = 256;
    long ldpad = 32;       // elements of padding per row: de-aliases power-of-two strides across HBM channels
    gp_timings tm{};
    std::vector<FreeBlock> pool;               // cached free device blocks (true sizes)
    std::unordered_map<void*, size_t> blk;     // true size of every block handed out by ctx_alloc
    size_t pool_bytes = 0;
    size_t pool_cap = (size_t)96 << 30;        // bytes kept in the cache at most ("pool_cap_mb"; gp_ctx_trim drops it all)
    long vfe_chunk = 0;                        // data points per streamed VFE chunk (multiple of vfe_ks); 0: automatic — 16 384 (measured best at C5) × a power of two for fewer pseudo-points (vfe.hpp)
    std::vector<hipEvent_t> ev_pool;
    size_t ev_used = 0;
    struct GemmRec {
        hipEvent_t a, b;
        double flops, bytes;
        long M, N, K;
        int stream;  // 0 main, 1 panel
    };
    std::vector<GemmRec> gemm_recs;
    hipEvent_t ev_phase[4] = {nullptr, nullptr, nullptr, nullptr};
    int* info_dev = nullptr;
    int* ticket_dev = nullptr;   // load tickets of panel64_kernel ([0]: main stream, [32]: panel stream)
    void* w_ws = nullptr;        // I − inv(L_jj) tiles for the MFMA triangular solve (trtri_64 output)
    size_t w_ws_bytes = 0;
    double* scal_dev = nullptr;  // [0] logdet accumulator, [8..] sumsq outputs
    long scal_cap = 0;
    std::atomic<int> refs{1};
    bool dead = false;
    struct gp_multi* multi = nullptr;  // non-null: a multi-device context (multi.hip); this ctx then is rank 0's device context
};

struct SkScope {
    gp_ctx* c;
    explicit SkScope(gp_ctx* c_) : c(c_) { ++c->sk_scope; }
    ~SkScope() { --c->sk_scope; }
};


// ---- handle registry ------------------------------------------------------------------------------
extern std::mutex g_reg_mu;
extern std::set<void*> g_live;
void reg_add(void* p);
bool reg_take(void* p);
bool reg_has(void* p);
void ctx_unref(gp_ctx* c);
int32_t ctx_alloc(gp_ctx* c, size_t bytes, void** out);
void ctx_release(gp_ctx* c, void* p, size_t requested);
int32_t ctx_event(gp_ctx* c, hipEvent_t* out, bool timing);
int32_t ctx_scal(gp_ctx* c, long n);
int32_t ctx_prime(gp_ctx* c, long nb_hint);          // lazy workspaces + the hardware queues of sm / sp, created now
int32_t ctx_prime_stream(gp_ctx* c, hipStream_t s);  // one empty kernel on s (creates its hardware queue), drained
// Validates a handle (gp_ctx / gp_post / gp_vfe) and locks its ctx without racing a concurrent *_free / gp_ctx_destroy from
// another thread: the ctx is pinned under the registry mutex, locked, and the handle is checked again under the ctx lock
// (every *_free removes its handle from the registry BEFORE it takes the ctx lock to release the buffers).
struct Guard {
    gp_ctx* c = nullptr;
    std::unique_lock<std::mutex> lk;
    bool ok = false;
    static gp_ctx* ctx_of(gp_ctx* h) { return h; }
    template <class H> static gp_ctx* ctx_of(H* h) { return h->ctx; }
    template <class H> explicit Guard(H* h) {
        {
            std::lock_guard<std::mutex> l(g_reg_mu);
            if (!h || !g_live.count((void*)h)) return;
            c = ctx_of(h);
            c->refs++;
        }
        lk = std::unique_lock<std::mutex>(c->mu);
        ok = reg_has((void*)h) && !c->dead;
    }
    ~Guard() {
        if (lk.owns_lock()) lk.unlock();
        if (c) ctx_unref(c);
    }
};

// RAII owner of the device blocks of one call: everything still owned when it goes out of scope returns to the ctx cache
// (every early-return / error path included); keep() hands a block over to a handle.
// page-locked host staging of at least `bytes` (NULL above 1 GiB or when the allocation fails: the caller then stages through pageable memory); valid until the next
// ctx_pinned call on the ctx — callers hold the ctx lock and synchronise their uploads before they return
void* ctx_pinned(gp_ctx* c, size_t bytes);

struct DevBufs {
    gp_ctx* c;
    std::vector<void*> v;
    explicit DevBufs(gp_ctx* c_) : c(c_) {}
    DevBufs(const DevBufs&) = delete;
    int32_t get(size_t bytes, void** out) {
        *out = nullptr;
        int32_t rc = ctx_alloc(c, bytes ? bytes : 16, out);
        if (rc == 0) v.push_back(*out);
        return rc;
    }
    void* keep(void* p) {
        for (auto& q : v)
            if (q == p) q = nullptr;
        return p;
    }
    // ONE reusable scratch block per API call (the inverse-block solve's S panel: trsm_cached runs once per chunk of test points, every chunk's work is
    // ordered on one stream, so the chunks share it; it grows only if a later chunk is larger than every earlier one)
    void* scr = nullptr;
    size_t scr_bytes = 0;
    int32_t scratch(size_t bytes, void** out) {
        if (scr && scr_bytes >= bytes) {
            *out = scr;
            return 0;
        }
        int32_t rc = get(bytes, out);
        if (rc == 0) {
            scr = *out;
            scr_bytes = bytes;
        }
        return rc;
    }
    ~DevBufs() {
        for (void* q : v)
            if (q) ctx_release(c, q, 0);
    }
};

// −inv(L_bb) of a resident factor's diagonal blocks (lower, row-major [np + 128][ldw]; rows of block (j0, n) at j0): gpmi355.hip dib_build / trsm_cached
struct DibCache {
    void* w = nullptr;
    size_t bytes = 0;
    long nbi = 0, ldw = 0;  // nbi = −1: this factor keeps the substitution leaves (conditioning guard)
};

// ---- posterior handle -----------------------------------------------------------------------------
struct gp_multi;
struct gp_multi_post;  // multi.hip: block-cyclic pieces of a factor that has not been gathered yet
struct gp_post {
    gp_ctx* ctx;
    int dtype;
    long n, np, ld, mtot;
    int d;
    int kind;
    double variance;
    int nscale;
    std::vector<double> scale;
    void* A;
    size_t A_bytes;  // factor (+ RHS rows)
    void* xs;
    size_t xs_bytes;  // scaled train inputs [d][np]
    void* alpha;
    size_t alpha_bytes;  // [np]
    double logdet_half;  // Σ log L_ii
    DibCache dibc;            // −inv(L_bb) of the factor's diagonal blocks, built on the first forward solve against the resident factor ("dib_nb")
    gp_multi_post* pieces = nullptr;  // multi-device fit: the factor still lives as block-cyclic pieces (A == nullptr until gathered)
};

// ---- engine entry points used by multi.hip (fp64; work is issued on stream s of ctx c and not synchronised) ----
namespace gpmi {
GridMap plain_map(int lower, long row0, long col0);
int32_t eng_assemble(gp_ctx* c, hipStream_t s, int kind, double variance, const double* x_dev, long n_valid, long n_pad, int d,
                     const double* noise_dev, GridMap g, double* a_loc, long lda, long m_loc, long n_loc);
int32_t eng_potrf(gp_ctx* c, hipStream_t s, double* a, long lda, long m, long n, int* info_dev, long col0, long n_valid,
                  double* logdet_dev);
int32_t eng_trsm(gp_ctx* c, hipStream_t s, double* x, long ldx, long m, const double* l, long ldl, long n);
// w (nb × ldw) ← −inv(l) of one nb×nb lower block; iw: scratch of the same shape; v: a second one (level-wise batched inverse for nb = 64·2^m; NULL: recursion).  x ← x l⁻ᵀ as one triangular-k GEMM with it (sc: (m + 128) × lds scratch).
int32_t eng_inv_lower(gp_ctx* c, hipStream_t s, const double* l, long ldl, long nb, double* w, long ldw, double* iw, double* v_or_null);
int32_t eng_trsm_inv(gp_ctx* c, hipStream_t s, double* x, long ldx, long m, const double* w, long ldw, long nb, double* sc, long lds);
int32_t eng_gemm_nt(gp_ctx* c, hipStream_t s, double* cm, long ldc, const double* a, long lda, const double* b, long ldb, long m,
                    long n, long k, GridMap g);
int32_t eng_trsv(gp_ctx* c, hipStream_t s, const double* l, long ldl, long np, double* r, long ldr, int nrhs, bool forward);
int32_t eng_gemv_t(gp_ctx* c, hipStream_t s, const double* l, long ldl, long nrows, long ncols, const double* a, double* r);
int32_t eng_rowsumsq(gp_ctx* c, hipStream_t s, const double* x, long ldx, long nrows, long ncols, double* out_dev);
int32_t eng_add_vec(gp_ctx* c, hipStream_t s, double* dst, const double* src, long n);  // dst += src
// out (nr_pad × nc_pad, row-major) = variance·κ(‖xr_i − xc_j‖), zero where i >= nr_valid or j >= nc_valid (cross-Gram block)
int32_t eng_kcross(gp_ctx* c, hipStream_t s, int kind, double variance, const double* xr, long ldxr, long nr_valid, long nr_pad,
                   const double* xc, long ldxc, long nc_valid, long nc_pad, int d, double* out, long ld);
// out[r] = Σ_j variance·κ(‖xs_r − x_j‖) alpha_j for nrows points xs (dimension-major, stride ldxs): rows of K·alpha without K
int32_t eng_kvec(gp_ctx* c, hipStream_t s, const double* xs, long ldxs, const double* x, long ldx, int d, int kind, double variance,
                 long n, const double* alpha, double* out, long nrows);
// register-resident 64-column leaf (leaf.hip, its own translation unit): tile Cholesky + X L⁻ᵀ of the mrows rows below, fp64
int32_t launch_leaf_v2(hipStream_t s, double* Ajj, long lda, long mrows, int* info_dev, int col0, int n_valid, double* logdet_dev, int* ticket,
                       int kpre, int xr, int num_cus, int ncols);
int32_t launch_panel_updk(hipStream_t s, double* C, long ldc, const double* P, long ldp, long m, long N, long K, int rt, int num_cus);
// 2-D block copy by a kernel (16-B aligned rows, even cols): source may live on a peer device with peer access enabled
int32_t eng_copy2d(gp_ctx* c, hipStream_t s, double* dst, long dld, const double* src, long sld, long rows, long cols);
}  // namespace gpmi

// ---- multi-device contexts (multi.hip) --------------------------------------------------------------
void multi_destroy(gp_multi* m);
int32_t multi_fit(gp_ctx* c, const gp_kernel* k, const gp_points* x, const gp_noise* noise, const void* mean_or_null,
                  const void* Y, long ldy, int ncols, double* logpdf_out, double* terms_out /* nullable: [0] logdet, [1+s] sqmahal_s */,
                  gp_post* post, void* alpha_out);
void multi_trim(gp_ctx* c);  // gp_ctx_trim of every rank context
int32_t multi_gather(gp_post* post);  // block-cyclic pieces -> one row-major factor on the ctx's first device
bool multi_can_solve(gp_post* post);  // predictive variances on the distributed factor possible (pieces not gathered, same grid alive)
// var_sub[s] = Σ_c X[s][c]², cov_sub (nullable, ns×ns) = X Xᵀ with X = K_*x L⁻ᵀ solved on the block-cyclic pieces
int32_t multi_predict_var(gp_post* post, const double* xs_scaled, long ns_ld, long ns, double* var_sub, double* cov_sub);
// out = C \ B (n×ncols column-major host arrays) by a forward pass and ncols backward sweeps on the pieces
int32_t multi_solve(gp_post* post, const double* B, int ncols, double* out);
// out = L ξ (n×ncols column-major host arrays): every rank multiplies the blocks it holds, partial products summed on the host
int32_t multi_factor_mul(gp_post* post, const double* xi, int ncols, double* out);
// sequential conditioning on the pieces: the factor of `old` extended by new block rows into `post` (-1991: self-check failed)
int32_t multi_update(gp_post* old, const gp_points* x2, const gp_noise* noise2, const void* delta_all, gp_post* post, void* alpha_out,
                     double* logpdf_out);
void multi_post_release(gp_post* post);
int32_t multi_set_param(gp_ctx* c, const char* name, int64_t v);  // 1 = not a multi parameter
int32_t multi_get_param(gp_ctx* c, const char* name, int64_t* out);  // 1 = not a multi parameter
