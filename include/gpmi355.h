/* gpmi355.h — C ABI of libgpmi355.so: MI355X-native (gfx950) exact / sparse GP inference.
 *
 * This is the drop-in boundary for ONE hot path of AbstractGPs.jl (reference paths are relative
 * to the upstream repo): logpdf(fx, y)   src/finite_gp_projection.jl:306-311
 *                         posterior(fx,y) src/exact_gpr_posterior.jl:29-35
 * plus what callers do next (predictive mean/var/cov, src/exact_gpr_posterior.jl:60-90) and the
 * VFE/DTC sparse variant (src/sparse_approximations.jl:58-75, 183-217, 248-313).
 *
 * The reference has no FFI of its own (pure Julia); its documented plug-in point is "subtype
 * AbstractGP and implement the FiniteGP primary API" (docs/src/api.md:18-30, 49-73).  The Julia
 * shim that does that with `ccall` into these entry points is abstractgps.jl_amd/julia/HipGPs.jl;
 * a line-for-line Python ctypes mirror (abstractgps.jl_amd/api.py) is what the test-suite runs.
 *
 * Conventions
 *  - plain C, no C++/torch types; every function returns int32 status:
 *        0      success
 *        k > 0  LAPACK-style info: leading minor of order k is not positive definite
 *               (the shim throws LinearAlgebra.PosDefException(k) exactly like `cholesky` at
 *               src/finite_gp_projection.jl:308 / src/exact_gpr_posterior.jl:31)
 *        -i     (1 <= i < 1000) argument i invalid
 *        -1000-e HIP error e;  text in gp_last_error() (thread-local)
 *  - host pointers are borrowed for the duration of the call only; outputs are caller-allocated
 *    host buffers; device memory, streams and workspaces are owned by the handles.
 *  - dtype: 0 = f64, 1 = f32.  All host arrays of one call share the kernel's dtype
 *    (Float32 in -> Float32 out is a tested reference property, test/finite_gp_projection.jl:180-191).
 *  - matrices handed back to the host are COLUMN-MAJOR (Julia layout).
 *  - functions taking a gp_ctx are serialised per ctx by an internal mutex and may be called from
 *    any OS thread (hipSetDevice is issued on entry).
 */
#ifndef GPMI355_H
#define GPMI355_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GPMI355_ABI_VERSION 4

typedef struct gp_ctx gp_ctx;   /* device + streams + workspace                          */
typedef struct gp_post gp_post; /* PosteriorGP state: device-resident factor, alpha, x   */
typedef struct gp_vfe gp_vfe;   /* ApproxPosteriorGP state (VFE / DTC)                    */

/* Kernel descriptor: variance * base(kind) ∘ transform.   Replaces KernelFunctions.kernelmatrix
 * at src/base_gp.jl:70,72,74.  kind: 0 SqExponential exp(-d²/2); 1 Matern12 exp(-d);
 * 2 Matern32 (1+√3d)exp(-√3d); 3 Matern52 (1+√5d+5d²/3)exp(-√5d).
 * nscale: 0 none; 1 ScaleTransform(scale[0]) (with_lengthscale(k,ℓ) ≡ scale = 1/ℓ); D ARDTransform(scale[0..D)). */
typedef struct {
    int32_t kind;
    int32_t dtype;
    double variance;
    int32_t nscale;
    const double* scale;
} gp_kernel;

/* Inputs (host).  layout: 0 = Vector{T} (d must be 1); 1 = ColVecs(X), X is d×n column-major
 * (point-contiguous); 2 = RowVecs(X), X is n×d column-major (dimension-contiguous).
 * src/finite_gp_projection.jl:32-37. */
typedef struct {
    const void* data;
    int64_t n;
    int32_t d;
    int32_t layout;
} gp_points;

/* Observation noise Σy.  kind 0: σ²·I (Fill, src/finite_gp_projection.jl:19-21); kind 1: Diagonal(diag)
 * (length n, host, kernel dtype) (:13-15).  Dense Σy is not accelerated (shim falls back). */
typedef struct {
    int32_t kind;
    double s;
    const void* diag;
} gp_noise;

/* Per-phase wall times (ms, HIP events on the ctx streams) of the last fit/logpdf call, and the
 * accumulated duration / algorithmic FLOPs of the dominant kernel (gemm_nt trailing update) when
 * parameter "time_kernels" is 1.  Exact fits: assemble = Gram phase, potrf = factorisation, solve = vector solves.
 * Sparse fits (gp_vfe_fit / _update / _append): assemble = the M×M prelude (K_zz, its Cholesky, inv(L_z)), potrf = the streamed pass
 * over the data points, solve = the M×M side after it (Λ_ε = chol(I + B Bᵀ), the vector solves). */
typedef struct {
    double assemble_ms;
    double potrf_ms;
    double solve_ms;
    double total_ms;
    double gemm_ms;        /* Σ per-launch durations of gemm_nt_sub launches (time_kernels=1)  */
    double gemm_flops;     /* Σ algorithmic flops of those launches                               */
    int64_t gemm_launches;
    double gemm_bytes;     /* Σ algorithmic bytes of those launches: C read + C write + both operand panels once */
} gp_timings;

/* ---- context ------------------------------------------------------------------------------ */
/* stream_or_null: a hipStream_t owned by the caller to issue the main-stream work on (e.g. the
 * current torch stream in the multi-process driver); NULL = the ctx creates its own. */
int32_t gp_ctx_create(gp_ctx** out, int32_t device, void* stream_or_null);
/* Multi-device context: ONE caller thread (the Julia process of docs/src/api.md:18-30) drives ndev devices of a node.
 * devices[r] is the HIP device of rank r = p·Q + q of the P×Q process grid over which gp_posterior_fit / gp_logpdf partition
 * K + Σy 2D block-cyclically in nb×nb blocks (one internal host thread per rank; panel traffic by RCCL grouped send/recv over
 * xGMI, or by peer copies — environment GPMI_COMM=rccl|p2p overrides the automatic choice).  P = Q = 0: grid chosen for the
 * fabric (P = ndev, Q = 1 on the full-mesh xGMI node, see multi.hip); nb = 0: 1024.  The same device may be listed several
 * times ("virtual ranks": the full schedule on one GPU with same-device copies — how CI exercises it).  Every other entry
 * point works unchanged on such a ctx: fp64 fits are distributed, everything else (and everything downstream of a fit:
 * predictions, updates, sampling — the factor is gathered onto devices[0] on first need) runs on devices[0].
 * fp32 fits and fits with more than 128 right-hand-side columns also run on devices[0] (single-device engine).
 * Extra parameters: "lookahead_depth" (1..3, default 2), "dist_nb", "multi_gemm_streamk" (stream-K cuts inside the rank
 * contexts, default 0), "multi_timeout_s" (a rank that waits longer for a peer or for its own streams fails the fit instead
 * of hanging, default 600), "multi_check" (diagnostics: 1 marker/checker kernels around every event record / wait, 2 operand
 * buffers compared with the owners' blocks before every update, 4 NaN-poisoned operand buffers; findings fail the fit with
 * status -1990).  Environment: GPMI_COMM=rccl|p2p, GPMI_RCCL_LIB=<library to dlopen instead of librccl>, GPMI_COMM_PRIO=1
 * (high-priority comm stream), GPMI_TRACE_SCHEDULE=<file> (JSON-lines schedule trace of the last fit). */
int32_t gp_ctx_create_multi(gp_ctx** out, const int32_t* devices, int32_t ndev, int32_t P, int32_t Q, int32_t nb);
/* Grid / transport of a ctx (1×1, nb 0, comm 0 for a single-device ctx).  comm: 1 RCCL, 2 peer / same-device copies. */
int32_t gp_ctx_multi_info(gp_ctx* ctx, int32_t* P, int32_t* Q, int32_t* nb, int32_t* comm, int32_t* depth);
/* Multi-device fits check their result before handing it out ("multi_verify", default 1: δᵀα against ‖L⁻¹δ‖², and (K + Σy)α = δ on
 * every row with K·α recomputed from the inputs on the devices) and are repeated once when the check or the factorisation fails; a second failure is the
 * error (-1991, or the LAPACK info).  fits = fit attempts so far, retries = repetitions (0 on a healthy stack); any pointer may be NULL. */
int32_t gp_ctx_multi_stats(gp_ctx* ctx, int64_t* fits, int64_t* retries, int64_t* solves);
/* Predictive variances — and full covariances for up to 4 096 test points — of a multi-device posterior (gp_posterior_predict) are
 * computed on the distributed factor: a block forward solve with the factor left where the fit put it, N*×nb blocks of the solution
 * travelling, one SYRK per rank for the covariance — and with them gp_posterior_logpdf / gp_posterior_rand for up to 4 096 test points
 * ("multi_dist_predict", default 1; `solves` above counts the passes over the pieces).  On the pieces as well:
 *   gp_posterior_solve  (<= 128 columns): the columns travel as rows — forward pass, then one backward block sweep per column;
 *   gp_posterior_update (<= 4 096 new observations): the factor is EXTENDED where it lives — K(x2, x1) L11⁻ᵀ by the forward pass, every
 *                       rank keeping the rows of the new blocks it owns; chol(C22 − U12ᵀU12) on devices[0], its blocks sent to their
 *                       owners; α by a forward + backward pass over the extended pieces.  The new posterior is again a multi-device
 *                       posterior (each batch of observations ends in its own padded blocks).
 *   gp_posterior_factor_mul (<= 1 024 columns): every rank multiplies the blocks it holds with its share of ξ, the process rows'
 *                       partial products are summed on the host — no exchange between the ranks.
 * C.U (gp_posterior_get_factor) and covariances of more than 4 096 test points gather the factor onto devices[0] (real points only: the
 * padding inside the blocks is dropped).
 * gp_multi_solve_trace / _ex write the schedule of such a pass for a P×Q grid (dry run of the real rank threads, like
 * gp_multi_schedule_trace); flags: 1 = given right-hand sides, 2 = rows kept for an extended factor (sequential update),
 * 4 = followed by two backward sweeps. */
int32_t gp_multi_solve_trace(int32_t P, int32_t Q, int32_t nblk, const char* path);
int32_t gp_multi_solve_trace_ex(int32_t P, int32_t Q, int32_t nblk, int32_t flags, const char* path);
/* The schedule the multi-device driver issues for a P×Q grid over nblk block columns (look-ahead depth 1..3; comm 1 =
 * send/recv transport, 2 = copies; + 16 = the schedule of "multi_trsm_inv" = 0, + 32 = that of "multi_chain_cus" > 0), written as JSON lines to `path`: every stream operation with its block footprint, every
 * event record / wait, every transfer — produced by the SAME rank-thread code that drives the devices, run without a device
 * (works on a machine without a GPU).  tools/multi_schedule_check.py checks happens-before on it. */
int32_t gp_multi_schedule_trace(int32_t P, int32_t Q, int32_t nblk, int32_t depth, int32_t comm, const char* path);
int32_t gp_ctx_destroy(gp_ctx* ctx);
/* Tuning / diagnostic parameters (all optional; the GPMI_PARAMS="name=value,..." environment variable applies the same
 * names at gp_ctx_create):
 *   "nb"             outer panel width: −1 = automatic ("nb_small" for matrices below "lookahead_min_n", "nb_large" from there on), 0 = purely
 *                    recursive, > 0 = that width (multiple of 128) at every size                default −1
 *   "nb_small", "nb_large"  the automatic widths: below the look-ahead threshold the schedule is one stream and 4 096-column panels halve the
 *                    passes over the trailing matrix (C2 −2.5 %); above it 2 048 / 4 096 / 8 192 measure within ± 0.5 %   default 4096, 2048
 *   "lookahead"      next panel on a second, high-priority stream (0/1)                   default 1
 *   "lookahead_min_n" ... for matrices of at least this (padded) order                    default 24576
 *   "time_kernels"   bracket every MFMA GEMM launch with HIP events (gp_get_timings)      default 0
 *   "xcd_swizzle", "xcd_min_tiles"  XCD-aware super-tile workgroup order for large GEMM grids   default 0, 256
 *   "gemm_streamk"   persistent-grid GEMM with a stream-K tail on launches of <= sk_max_tiles tiles   default 1
 *                    (0 in the rank contexts of a multi-device ctx: "multi_gemm_streamk").  The tail adds its k-slices into C with
 *                    hardware floating-point atomics, so the order of summation depends on scheduling: results agree with the
 *                    oracle at the stated tolerances but are NOT bitwise reproducible from run to run; 0 removes this source of
 *                    variation from the factorisation (hardware-dispatched GEMMs only, 1-5 % slower at N <= 32 768; the backward
 *                    vector sweep still adds four partial products per column with atomics)
 *   "deterministic"  1: the exact path (gp_logpdf, gp_logpdf_terms, gp_posterior_fit and every method of its posterior) uses no
 *                    floating-point atomics — no stream-K tails whatever "gemm_streamk" says, one thread per column in the backward
 *                    sweep — so two calls with the same inputs on the same ctx return the same BITS (tests/test_gpu_api.py; the leaf's one
 *                    Σ log L_ii add per launch is issued by a single thread and the leaves of a fit are totally ordered, so its order is fixed).  The VFE
 *                    path, the gradient kernels and the multi-device backward sweep keep their atomics.  1-5 % slower at N <= 32 768.   default 0
 *   "leaf_v2", "leaf_xr"  fp64 leaves by the register-resident panel64v2_kernel (csrc/leaf.hpp) / rows of X per leaf workgroup (0 auto)   default 1, 0
 *   "leaf_cols"      columns per register-resident leaf launch (64 or 128; 128 = one workgroup chain per 128 columns)   default 128
 *   "upd128", "updk_max_k"  in-panel updates C[m×N] −= P·P[0:N]ᵀ of the panel recursion by panel_updk_kernel (csrc/leaf.hpp: register chain, 16-row
 *                    wave tiles) instead of the tile GEMM: K = N = 128 (on/off) and 256 <= K = N <= updk_max_k; K above "updk_tall_k" only
 *                    while at most "updk_tall_m" rows are below                          default 1, 512, 256, 8192
 *   "updk_rt"        rows per workgroup / 16 of that kernel (0 auto: tallest tile giving one workgroup per CU; 4, 2, 1)   default 0
 *   "sk_min_k"       smallest inner dimension that may use the stream-K GEMM                default 0
 *   "leaf_group"     columns factored left-looking by consecutive fused leaves (64/128/256/512)   default 128
 *   "trsv_nb"        diagonal block of the vector solves handled by one workgroup (128..1024)    default 256
 *   "gemm_pad_lds"   extra dynamic LDS bytes per GEMM workgroup; 20480 = one workgroup per CU (fp64: same speed on one
 *                    large launch, 4-7 % slower over a whole factorisation)                 default 0
 *   "gemm_pad_f32"   the same for the fp32 GEMMs unless "gemm_pad_lds" was set (0: two workgroups per CU — 1 % faster at C5 with the
 *                    pipelined k loop; with "gemm_pipe" = 0 one workgroup per CU, 20480, was 5 % faster)   default 0
 *   "gemm_pipe"      k loop of the MFMA GEMMs software-pipelined across the step boundary (csrc/kernels.hpp gemm_kloop_pipe: the fragments of
 *                    the next half-step are in registers before the barrier, operand DMA issued between MFMAs); 0 = the round-2 loop   default 1
 *   "kmat_rows"      Gram tiles row by row (distance, κ, store; one kernel instance per dimension bucket D <= 4 / 8 / 16, few registers) instead of
 *                    all 64 squared distances of a thread accumulated first (the form D > 16 always takes); process-wide   default 1
 *   "dib_nb"         forward solves X L⁻ᵀ against a resident factor (predictive variances / covariances, held-out logpdf, sampling, sequential
 *                    conditioning, the gradient's L⁻ᵀ; both factors of a VFE / DTC prediction): column blocks of at most this width are solved by ONE triangular-k MFMA GEMM with the
 *                    explicit inverse of the diagonal block — built once per posterior handle (np × (dib_nb + 32) elements, 3 % of the factor at
 *                    N = 65 536) on its first forward solve; 0 = the recursion down to 64-column TRSM leaves.  A product with an explicit inverse
 *                    carries an error of order cond(L_bb)·ε where substitution is backward stable: a handle whose factor has max |L_ii| / min |L_ii|
 *                    above 1e5 (cond(K + Σy) >= 1e10) keeps the substitution leaves                   default 2048
 *   "ldpad"          row padding in elements (multiple of 16)                             default 32
 *   "vfe_chunk"      data points per streamed VFE chunk (multiple of vfe_ks); 0 = automatic: 16 384 for M > 2 048 pseudo-points (measured best at C5),
 *                    × 2 … 16 for fewer (chunk × M kept at C5's footprint), at most the batch; a handle keeps the chunk of its first fit        default 0
 *   "vfe_ks"         fp32 VFE: data points per fp32 partial product of the chunk SYRK      default 2048
 *   "vfe_overlap"    VFE: kmat / reductions / partial-sum adds on a second stream beside the chunk GEMMs   default 1
 *   "vfe_dual"       VFE experiment switch: 1 = the triangular products of the chunks on a high-priority third stream beside the chunk SYRKs on the main stream
 *                    (needs "vfe_overlap").  Measured at C5 and left off: +0.4…0.7 ms (profiles/r6/c5_ab2.jsonl; both launches at equal priority: −0.45 ms)   default 0
 *   "vfe_inv_nb"     VFE prelude: inv(L_z) through inverse diagonal blocks of this width, built in one batched launch sequence, the levels above them one
 *                    triangular-k GEMM per block (≈ 40 launches at M = 4 096 instead of 127); 0 = the recursion down to 64-wide leaves   default 512
 *   "sk_max_tiles"   largest launch (in 128×128 tiles) that takes the persistent stream-K GEMM    default 4096
 *   "vfe_sk"         VFE: stream-K GEMM tails for the M×M side (K_zz / Λ_ε factorisations, inv(L_z))   default 0
 *   "copy_kernel"    multi-device: block copies by a kernel instead of hipMemcpy2DAsync            default 0
 *   "multi_leaf_cols"  multi-device: columns per register-resident leaf inside the rank contexts — 64 (46 KB of LDS: starts beside the bulk
 *                    update) rather than the single-device 128 (152 KB: waits for an empty CU, i.e. for the end of a GEMM launch)   default 64
 *   "multi_window"   multi-device: block steps a rank thread may queue ahead of its device         default 16
 *   "multi_trsm_inv" multi-device: the diagonal owner of block column k forms −inv(L_kk) (kept in a slot of its own) and THAT travels to the column's other
 *                    owners instead of L_kk; every owner then solves its rows below the block with ONE triangular-k MFMA GEMM + one copy instead of the
 *                    substitution recursion's nb/64 latency-bound leaf launches and as many few-tile GEMMs (nb = 1 024: 31 launches per rank and step -> 2).
 *                    Same conditioning rule as "dib_nb", decided from the inputs before the fit: every pivot of K + Σy lies in
 *                    [min Σy_ii, variance + max Σy_ii], and a fit with sqrt((variance + max Σy) / min Σy) > 1e5 keeps the substitution solve.  1: the owner forms the inverse
 *                    level by level (−inv and its transpose of every 64×64 tile, then 3 batched GEMMs per level: 13 launches at nb = 1 024; nb = 64·2^m), 2: by the
 *                    restricted-row recursion on the identity (33 launches; what 1 falls back to for other nb).   default 1
 *   "multi_chain_cus" multi-device: r > 0 reserves r CUs (r/8 of every XCD; multiple of 8, at most half the device) of every rank's GPU for the diagonal
 *                    block's chain (Cholesky of the nb×nb block + its inverse) on a CU-masked stream of its own; the panel and main streams' work then
 *                    runs on streams masked to the other CUs.  Measured on one GPU (profiles/r5/cumask_chain_probe.jsonl): a 1 024-column chain beside the bulk
 *                    update 3.6 ms unmasked -> 0.46 ms on 16 CUs, the update 4 % slower; masked HIP streams carry no priority, so the look-ahead update
 *                    loses its priority over the bulk update — the trade-off a multi-GPU run has to price (tools/scale_sweep.sh A/Bs it).   default 0
 *   "multi_debug_sync", "multi_inject_fault"  multi-device diagnostics: host synchronisation points of the rank threads (bit mask) /
 *                    hand the next fit's self-check a spoiled α once (tests/test_gpu_multi.py)      default 0, 0
 *   "pool_cap_mb"    device bytes (MiB) the ctx keeps cached for reuse after *_free; ONE block larger than the cap may stay cached (the 137 GB factor
 *                    of N = 131 072: 17.2 s per pair when it is returned to the driver and allocated again every fit, 11.0 s cached) — it takes the
 *                    place of everything else in the cache and is the last block to go; 0 caches nothing; gp_ctx_trim empties the cache; an
 *                    allocation that fails drops the cache and retries.   default 98304
 *   "pool_cached_mb", "pool_blocks"   read-only (gp_ctx_get_param): MiB and number of blocks in the cache now */
/* The defaults above, machine-readable (single-device parameters; gp_ctx_get_param reads the same names): the test-suite asserts before
 * every GPU test that the shared default context still has exactly these values, so that no test can leave a non-production setting
 * behind for the tests that follow it (tests/conftest.py). */
#define GPMI355_PARAM_DEFAULTS                                                                                                   \
    "nb=-1,nb_small=4096,nb_large=2048,lookahead=1,lookahead_min_n=24576,time_kernels=0,xcd_swizzle=0,xcd_min_tiles=256,gemm_streamk=1,sk_max_tiles=4096," \
    "sk_min_k=0,gemm_pipe=1,gemm_pad_f32=0,gemm_pad_lds=0,trsv_nb=256,deterministic=0,leaf_v2=1,leaf_xr=0,leaf_cols=128,"    \
    "updk_max_k=512,updk_rt=0,updk_tall_k=256,updk_tall_m=8192,upd128=1,leaf_group=128,ldpad=32,vfe_ks=2048,vfe_sk=0,"          \
    "vfe_overlap=1,vfe_dual=0,vfe_inv_nb=512,vfe_chunk=0,kmat_rows=1,dib_nb=2048,pool_cap_mb=98304"
int32_t gp_ctx_set_param(gp_ctx* ctx, const char* name, int64_t value);
/* Read a parameter back (same names; "gemm_pad_lds" reads 0 until it has been set explicitly).  Used by the test-suite to assert that
 * every GPU test starts from the documented defaults. */
int32_t gp_ctx_get_param(gp_ctx* ctx, const char* name, int64_t* value_out);
/* Return every cached (free) device block of the ctx to the HIP allocator — e.g. after freeing an N = 65 536 posterior
 * (34 GB) when another allocator in the process needs the memory. */
int32_t gp_ctx_trim(gp_ctx* ctx);
int32_t gp_get_timings(gp_ctx* ctx, gp_timings* out);
const char* gp_last_error(void);
int32_t gp_abi_version(void);

/* ---- kernelmatrix (parity / small N) ------------------------------------------------------- */
/* out (host, column-major): n×n for y == NULL (exactly symmetric), else x.n × y.n.
 * KernelFunctions.kernelmatrix(k, x[, y]) — src/base_gp.jl:70,74. */
int32_t gp_kernelmatrix(gp_ctx* ctx, const gp_kernel* k, const gp_points* x, const gp_points* y_or_null,
                        void* out);

/* ---- exact GP: logpdf / posterior --------------------------------------------------------- */
/* logpdf(f(x, Σy), Y): Y is n×ncols column-major with leading dimension ldy; out has ncols entries.
 * mean_or_null: prior mean vector m (length n) evaluated on the host, NULL = ZeroMean.
 * src/finite_gp_projection.jl:306-311, 325-326. */
int32_t gp_logpdf(gp_ctx* ctx, const gp_kernel* k, const gp_points* x, const gp_noise* noise,
                  const void* mean_or_null, const void* Y, int64_t ldy, int32_t ncols, void* out);

/* The two terms of logpdf separately — sqmahal(fx, Y) (src/finite_gp_projection.jl:313-326: ‖U⁻ᵀ(y_s − m)‖² per column) and
 * logdet(cov(fx)) (:310) — from ONE factorisation.  Y may be NULL (logdet only; sqmahal_out must then be NULL too).
 * logdet_out: 1 entry, sqmahal_out: ncols entries, kernel dtype; either may be NULL.  (gradlogpdf(fx, y), :328-337, is −α of
 * gp_posterior_fit.) */
int32_t gp_logpdf_terms(gp_ctx* ctx, const gp_kernel* k, const gp_points* x, const gp_noise* noise, const void* mean_or_null,
                        const void* Y_or_null, int64_t ldy, int32_t ncols, void* logdet_out_or_null, void* sqmahal_out_or_null);

/* posterior(f(x, Σy), y): ONE Gram assembly + ONE factorisation serve both results.
 * alpha_out_or_null: length n (α = C \ (y - m));  logpdf_out_or_null: 1 entry (= logpdf(fx, y)).
 * The factor stays on the device inside *out.  src/exact_gpr_posterior.jl:29-35. */
int32_t gp_posterior_fit(gp_ctx* ctx, const gp_kernel* k, const gp_points* x, const gp_noise* noise,
                         const void* mean_or_null, const void* y, gp_post** out, void* alpha_out_or_null,
                         void* logpdf_out_or_null);

/* Predictive quantities at xs (src/exact_gpr_posterior.jl:60-90).  what: bit0 mean, bit1 var,
 * bit2 full cov (ns×ns column-major).  prior_mean_xs_or_null: m(x*) evaluated on the host.
 * mean = m(x*) + K_*x α;  var = k** - colsumsq(U⁻ᵀ K_x*);  cov = K** - VᵀV. */
int32_t gp_posterior_predict(gp_post* post, const gp_points* xs, const void* prior_mean_xs_or_null,
                             int32_t what, void* mean_out, void* var_out, void* cov_out);
/* logpdf(post(x*, Σy*), Y*) on the device: the FiniteGP-over-PosteriorGP path of the reference — mean_and_cov
 * (src/exact_gpr_posterior.jl:78-83) + Σy* (src/finite_gp_projection.jl:133-136), cholesky (:308), logdet + _sqmahal
 * (:310, :325-326).  Y: ns × ncols column-major, leading dimension ldy; out: ncols entries.  The N*×N* predictive
 * covariance is built and factored in HBM against the resident N×N factor; nothing is refitted. */
int32_t gp_posterior_logpdf(gp_post* post, const gp_points* xs, const void* prior_mean_xs_or_null, const gp_noise* noise,
                            const void* Y, int64_t ldy, int32_t ncols, void* out);
/* rand(rng, post(x*, Σy*), ncols) with the standard normals supplied by the caller (src/finite_gp_projection.jl:233-237):
 * out[:, s] = m* + chol(C*).U' xi[:, s]; xi and out are ns × ncols column-major (leading dimension ns). */
int32_t gp_posterior_rand(gp_post* post, const gp_points* xs, const void* prior_mean_xs_or_null, const gp_noise* noise,
                          const void* xi, int32_t ncols, void* out);

/* Value and gradient of logpdf(f(x, Σy), y) — the pullback a ChainRules rrule for the accelerated logpdf needs (the
 * reference differentiates logpdf by AD: test/finite_gp_projection.jl:152-178 and the examples).  One factorisation, then
 * C⁻¹ = L⁻ᵀL⁻¹ (blocked TRSM on the identity + MFMA SYRK) and one fused pass over the lower triangle:
 *   ∂/∂θ = ½ Σ_ij (α_i α_j − C⁻¹_ij) ∂C_ij/∂θ.
 * Outputs (all optional except logpdf_out; kernel dtype unless noted):
 *   dvariance_out  double[1]        ∂/∂(kernel variance)
 *   dscale_out     double[nscale]   ∂/∂scale (ScaleTransform s, or ARDTransform v_p; any D: 16 dimensions per pass)
 *   dnoise_out     noise.kind 0: 1 entry ∂/∂σ² = ½(αᵀα − tr C⁻¹);  kind 1: n entries ½(α_i² − C⁻¹_ii)
 *   dy_out         n entries ∂/∂y = −α   (∂/∂m = +α for a mean vector m)
 *   dx_out         n×d entries ∂/∂x in the container layout of x (what a deep-kernel model back-propagates into its feature
 *                  map, examples/2-deep-kernel-learning/script.jl); the prior mean is taken as constant in x */
int32_t gp_logpdf_grad(gp_ctx* ctx, const gp_kernel* k, const gp_points* x, const gp_noise* noise, const void* mean_or_null,
                       const void* y, void* logpdf_out, double* dvariance_out, double* dscale_out, void* dnoise_out,
                       void* dy_out, void* dx_out);

/* Sequential conditioning, posterior(fx::FiniteGP{<:PosteriorGP}, y) (src/exact_gpr_posterior.jl:46-56): the resident
 * factor of `old` is extended by the bordered-Cholesky step update_chol (src/util/common_covmat_ops.jl:38-42):
 *   U12 = U11'\C12  (here: rows K(x2,x1)·L11⁻ᵀ by the blocked MFMA TRSM),  U22 = chol(C22 − U12'U12).
 * x2 / noise2: the new inputs and their noise;  delta_all: δ = vcat(δ_old, y2 − m(x2)) (length n_old + n2, host).
 * `old` stays valid.  alpha_out (n_old + n2) and logpdf_out (logpdf of all observations under the prior) optional. */
int32_t gp_posterior_update(gp_post* old, const gp_points* x2, const gp_noise* noise2, const void* delta_all,
                            gp_post** out, void* alpha_out_or_null, void* logpdf_out_or_null);

/* out[:, s] = C.U' * xi[:, s] (n×ncols column-major host arrays, leading dimension n): the sampling transform of
 * rand / _rand! (src/finite_gp_projection.jl:233-237, 271-277); the caller draws xi = randn(rng, n, ncols). */
int32_t gp_posterior_factor_mul(gp_post* post, const void* xi, int32_t ncols, void* out);

/* out[:, s] = C \ B[:, s] (n×ncols column-major host arrays) by forward + backward sweeps over the resident factor — what
 * gradlogpdf(fx, X) = C \ (m .- X) (src/finite_gp_projection.jl:328-337) needs for the columns beyond the fitted one, and
 * `post.data.C \ v` of user code. */
int32_t gp_posterior_solve(gp_post* post, const void* B, int32_t ncols, void* out);

/* C.U (n×n column-major upper, strictly-lower part zero) to the host — parity / debugging only. */
int32_t gp_posterior_get_factor(gp_post* post, void* U_out);
/* logdet(post.data.C) = 2 Σ log U_ii, kept from the fit (always double). */
int32_t gp_posterior_logdet(gp_post* post, double* out);
int64_t gp_posterior_n(gp_post* post);
int32_t gp_posterior_free(gp_post* post); /* NULL or already-freed handle: returns -1, never UB-free twice */

/* ---- VFE / DTC sparse approximation -------------------------------------------------------- */
/* posterior(VFE(f(z, jitter)), f(x, Σy), y)  (src/sparse_approximations.jl:58-75) and
 * approx_log_evidence / elbo (:248-254 VFE, :282-286 DTC) from the same intermediates.
 * approx: 0 VFE, 1 DTC.  objective_out_or_null: 1 entry.  K_xz is streamed in row blocks and never
 * materialised; only M×M matrices live in HBM. */
int32_t gp_vfe_fit(gp_ctx* ctx, const gp_kernel* k, const gp_points* x, const gp_points* z,
                   const gp_noise* noise, double jitter, const void* mean_or_null, const void* y,
                   int32_t approx, gp_vfe** out, void* objective_out_or_null);
/* update_posterior(f_post_approx, fx, y) with the same pseudo-points (src/sparse_approximations.jl:87-121): the streamed
 * reductions of `old` (B Bᵀ, B b_y, ‖B‖²_F, the Σy terms) are continued with the new observations and the M×M side is
 * re-finalised — equal to refitting on all observations.  objective_out: ELBO / DTC evidence of all observations. */
int32_t gp_vfe_update(gp_vfe* old, const gp_points* x2, const gp_noise* noise2, const void* mean2_or_null, const void* y2,
                      gp_vfe** out, void* objective_out_or_null);
/* update_posterior(f_post_approx, fz) — append pseudo-points (src/sparse_approximations.jl:131-176): bordered Cholesky of
 * K_zz against the resident factor (update_chol, src/util/common_covmat_ops.jl:38-42), then the retained observations are
 * streamed once more to form ONLY the new block rows of B Bᵀ / B b_y / ‖B‖²_F (the reference keeps B_εf, x, Σy, b_y in its
 * cache for this; here x, Σy^-1/2 and b_y stay on the device and B is never stored).  As in the reference (:138) the new
 * diagonal block C22 = cov(prior, z2) carries no jitter.  `old` stays valid.  objective_out: ELBO / DTC evidence with the enlarged pseudo-point set. */
int32_t gp_vfe_append(gp_vfe* old, const gp_points* z2, gp_vfe** out, void* objective_out_or_null);
/* mean / var / cov / mean_and_var / mean_and_cov (src/sparse_approximations.jl:183-217).  what: bit0 mean, bit1 var,
 * bit2 full cov (ns×ns column-major) = K** − AᵀA + (Λ_ε.U⁻ᵀA)ᵀ(Λ_ε.U⁻ᵀA). */
int32_t gp_vfe_predict(gp_vfe* post, const gp_points* xs, const void* prior_mean_xs_or_null, int32_t what,
                       void* mean_out, void* var_out, void* cov_out);
/* logpdf(f_post_approx(x*, Σy*), Y*) and rand(...) — same contracts as gp_posterior_logpdf / gp_posterior_rand. */
int32_t gp_vfe_logpdf(gp_vfe* post, const gp_points* xs, const void* prior_mean_xs_or_null, const gp_noise* noise,
                      const void* Y, int64_t ldy, int32_t ncols, void* out);
int32_t gp_vfe_rand(gp_vfe* post, const gp_points* xs, const void* prior_mean_xs_or_null, const gp_noise* noise,
                    const void* xi, int32_t ncols, void* out);
int64_t gp_vfe_m(gp_vfe* post); /* number of pseudo-points */
/* α = U \ m_ε (length M) and m_ε — cache fields of src/sparse_approximations.jl:73. */
int32_t gp_vfe_get(gp_vfe* post, void* alpha_out_or_null, void* m_eps_out_or_null);
/* The two M×M factors of the cache (src/sparse_approximations.jl:73: `U` = cholesky(cov(fz)).U and `Λ_ε.U`, read field by field by
 * test/sparse_approximations.jl:48-55, 76-83): M×M column-major UPPER triangular host arrays in the posterior's dtype (the strictly
 * lower part is zero-filled); either pointer may be NULL.  The device keeps both as fp64 row-major lower factors — byte-identical. */
int32_t gp_vfe_get_factors(gp_vfe* post, void* U_out_or_null, void* Lambda_U_out_or_null);
int64_t gp_vfe_n(gp_vfe* post); /* observations seen so far (fit + every update_posterior) */
/* b_y = U_y⁻ᵀ (y − m) of every observation seen so far, length gp_vfe_n(post) (cache field b_y, :66, :102), posterior's dtype. */
int32_t gp_vfe_get_by(gp_vfe* post, void* b_y_out);
/* Gradient of the objective the handle was fitted with — elbo for VFE (src/sparse_approximations.jl:248-254), approx_log_evidence for DTC (:282-286) — at the
 * handle's own parameters, after any number of gp_vfe_update / gp_vfe_append calls (the observations are retained on the device; B_εf is never stored: they are
 * streamed once more, one MFMA GEMM per chunk).  The reference has no hand-written adjoint: this is what an AD backend computes when
 * examples/0-intro-1d/script.jl:385-394 maximises the ELBO over kernel parameters and pseudo-points.  Every output may be NULL.
 *   dvariance      ∂/∂σ_k² (the ScaledKernel factor)             dscale [nscale]  ∂/∂s (ScaleTransform) or ∂/∂v_p (ARDTransform)
 *   dnoise_sum     Σ_i ∂/∂Σy_ii (the gradient for a scalar σ²)   dnoise_diag [n]  ∂/∂Σy_ii, the handle's dtype, n = gp_vfe_n(post), arrival order
 *   dy [n]         ∂/∂y_i (= −∂/∂m_i of the prior mean), the handle's dtype
 *   dz [m·d]       ∂/∂z in the container layout z_layout (0 vector, 1 ColVecs D×M column-major, 2 RowVecs M×D column-major), fp64.  fp64 handles only (status −7
 *                  otherwise): ∂/∂z is the small difference of the K_zz and K_fz terms, and the B Bᵀ an fp32 fit accumulated from fp32 products does not carry it
 *                  (measured at C5: 69 % off; the other outputs of an fp32 handle are within 2e-3 of the fp64 oracle's — the backward pass itself is always fp64)
 *   dx [n·d]       ∂/∂x in the container layout x_layout, the handle's dtype (adds one row reduction per observation to the streamed pass)
 * gp_get_timings afterwards: assemble = the M×M side, potrf = the streamed pass, solve = the K_zz term. */
int32_t gp_vfe_grad(gp_vfe* post, double* dvariance_or_null, double* dscale_or_null, double* dnoise_sum_or_null, void* dnoise_diag_or_null, void* dy_or_null,
                    double* dz_or_null, int32_t z_layout, void* dx_or_null, int32_t x_layout);
int32_t gp_vfe_free(gp_vfe* post);

/* ---- device-level building blocks ------------------------------------------------------------ */
/* The operations the in-library multi-device driver composes (csrc/multi.hip calls the same engine functions), exposed on DEVICE
 * memory so that they can be unit-tested one by one (tests/test_gpu_units.py) and driven by a caller that keeps its own
 * device-resident data.  All pointers below are DEVICE pointers (fp64), row-major with the given leading dimension, i.e.
 * a row-major lower factor L — memory-identical to Julia's column-major C.U.  Work is issued on the
 * ctx main stream and NOT synchronised (gpd_sync, or order it against your own stream work).
 * m, n multiples of 64 (gpd_trsv: np multiple of 128); k multiple of 16 (gemm). */

/* Fill local tiles of K + Σy.  rows: global indices row0 + i (i < m) mapped through the block-cyclic
 * map (global tile-row of local 128-tile t is ((t / tb) * P + p) * tb + t % tb); same for columns with
 * (Q, q).  x_dev is n_total×d in RowVecs layout (dimension-contiguous) already on the device;
 * noise_dev has n_total entries (padding rows get identity).  */
typedef struct {
    int32_t P, p, Q, q; /* process grid and my coordinates                                */
    int32_t tb;         /* 128-tiles per distribution block (NB / 128)                       */
    int32_t lower;      /* 1: skip 128-tiles strictly above the global diagonal             */
} gp_grid;

int32_t gpd_assemble(gp_ctx* ctx, const gp_kernel* k, const double* x_dev, int64_t n_valid, int64_t n_pad,
                     int32_t d, const double* noise_dev, const gp_grid* g, double* a_loc, int64_t lda,
                     int64_t m_loc, int64_t n_loc);
/* In-place lower Cholesky of the n×n diagonal block at a (rows/cols [0,n)), and of the m-n rows
 * below it (X ← X L⁻ᵀ) when m > n.  info_dev: device int32, set to (col0 + failing column, 1-based)
 * on the first non-positive pivot.  logdet_dev += Σ log L_ii over columns col0+i < n_valid.
 * m, n multiples of 64; `a` 16-byte aligned and lda even (the leaf kernels move rows as 16-byte pieces): anything else is refused (−2 / −3). */
int32_t gpd_potrf(gp_ctx* ctx, double* a, int64_t lda, int64_t m, int64_t n, int32_t* info_dev, int32_t col0,
                  int64_t n_valid, double* logdet_dev);
/* X (m×n) ← X · L⁻ᵀ with L the n×n lower factor (row-major, ldl). */
int32_t gpd_trsm(gp_ctx* ctx, double* x, int64_t ldx, int64_t m, const double* l, int64_t ldl, int64_t n);
/* The two pieces of the multi-device panel step ("multi_trsm_inv"; csrc/multi.hip):
 *   gpd_inv_lower: W (nb × ldw, lower) ← −inv(L) for the nb×nb lower block L.  W must be zero above its diagonal on entry and, like both scratch blocks, followed by
 *                  128 finite slack rows; scratch1 / scratch2: (nb + 128) × ldw doubles, ZEROED by the caller before the first use (their untouched triangles must stay
 *                  zero).  nb = 64·2^m with scratch2 != NULL: level by level in 1 + 3·log2(nb/64) batched launches; otherwise (scratch2 NULL or another nb, nb a
 *                  multiple of 64): the restricted-row recursion on the identity.
 *   gpd_trsm_inv : X (m × nb) ← X · L⁻ᵀ = −X · Wᵀ with that W as ONE triangular-k GEMM into scratch ((m + 128) × lds doubles) + one copy back. */
int32_t gpd_inv_lower(gp_ctx* ctx, const double* l, int64_t ldl, int64_t nb, double* w, int64_t ldw, double* scratch1, double* scratch2_or_null);
int32_t gpd_trsm_inv(gp_ctx* ctx, double* x, int64_t ldx, int64_t m, const double* w, int64_t ldw, int64_t nb, double* scratch, int64_t lds);
/* C (m×n) -= A (m×k) · B (n×k)ᵀ.  With g != NULL and g->lower, 64×64 sub-tiles strictly above the
 * global diagonal are skipped; row0/col0 = local absolute index of C's first row/column (mapped to
 * global indices through g).  g == NULL: plain rectangular update. */
int32_t gpd_gemm_nt(gp_ctx* ctx, double* c, int64_t ldc, const double* a, int64_t lda, const double* b,
                    int64_t ldb, int64_t m, int64_t n, int64_t k, const gp_grid* g_or_null, int64_t row0,
                    int64_t col0);
/* nrhs vectors stored as rows r[s*ldr + i], i < np: forward (L z = r) or backward (Lᵀ a = r) solve in place. */
int32_t gpd_trsv(gp_ctx* ctx, const double* l, int64_t ldl, int64_t np, double* r, int64_t ldr, int32_t nrhs,
                 int32_t forward);
/* r[j] -= Σ_{i<nrows} l[i*ldl + j] · a[i]  for j < ncols  (block row of Lᵀ times a vector; backward sweep). */
int32_t gpd_gemv_t(gp_ctx* ctx, const double* l, int64_t ldl, int64_t nrows, int64_t ncols, const double* a,
                   double* r);
/* out_dev[i] = Σ_{c<ncols} x[i*ldx + c]² for i < nrows. */
int32_t gpd_rowsumsq(gp_ctx* ctx, const double* x, int64_t ldx, int64_t nrows, int64_t ncols, double* out_dev);
int32_t gpd_sync(gp_ctx* ctx);
/* With parameter "time_kernels" = 1 every gpd_gemm_nt launch is bracketed by HIP events on the ctx
 * stream.  This synchronises the stream, returns the summed launch durations (ms) and the launch count since the
 * previous call, and clears the records (the caller knows the algorithmic flops of its own launches). */
int32_t gpd_gemm_time(gp_ctx* ctx, double* ms_out, int64_t* launches_out);

/* The multi-device transport's library on ONE device: dlopen (librccl, or the library GPMI_RCCL_LIB names), the seven entry points,
 * ncclCommInitAll(1) and one grouped ncclSend / ncclRecv pair of `count` doubles from the rank to itself, every element compared.
 * max_abs_err_out_or_null: largest |received − sent| (0 when the element type constant and the posting rules are what the driver
 * assumes).  −1996: the library is unavailable; −1998: an RCCL call failed (text in gp_last_error()). */
int32_t gp_rccl_selftest(int32_t device, int64_t count, double* max_abs_err_out_or_null);

/* ---- probes used by tools/gpu_diag.py and bench.py ------------------------------------------ */
/* D(16×16) = A(16×4)·B(4×16), all row-major host arrays: checks the f64 MFMA lane maps. */
int32_t gp_probe_mfma_f64(gp_ctx* ctx, const double* a_host, const double* b_host, double* d_host);
/* measured TFLOP/s of back-to-back v_mfma_f64_16x16x4_f64 on all CUs (the roofline's measured ceiling). */
int32_t gp_bench_mfma_f64(gp_ctx* ctx, int32_t iters, double* tflops_out);
/* the same for fp32: variant 0 = v_mfma_f32_16x16x4_f32, 1 = v_mfma_f32_32x32x2_f32 (the C5 roofline's measured ceiling). */
int32_t gp_bench_mfma_f32(gp_ctx* ctx, int32_t variant, int32_t iters, double* tflops_out);

#ifdef __cplusplus
}
#endif
#endif /* GPMI355_H */
