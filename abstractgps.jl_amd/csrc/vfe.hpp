// vfe.hpp — VFE / DTC sparse approximation on the device (included by gpmi355.hip).
//
// Reference: posterior(::Union{VFE,DTC}, fx, y)      src/sparse_approximations.jl:58-75
//            update_posterior (new observations)      :87-121      update_posterior (new pseudo-points)  :131-176
//            approx_log_evidence / elbo              :248-254 (VFE), :282-286 (DTC), :289-305
//            predictive mean / cov / var / joint      :183-217
//
// Formulation (S = Σy^-1/2 diagonal, L_z L_zᵀ = K_zz + jitter·I, so U = L_zᵀ), the reference's own:
//   B = U⁻ᵀ (S K_xz)ᵀ,  D = B Bᵀ + I = Λ_ε,  c = B b_y,  m_ε = Λ_ε \ c,  α = U \ m_ε.
//   The N-long pass is streamed in chunks of CH data points and never holds an N×M matrix.  L_z is the same for
//   every chunk, so its inverse is formed once (fp64: I·L_z⁻ᵀ by the blocked TRSM, transposed, rounded to T) and the
//   per-chunk triangular solve becomes ONE MFMA GEMM with a triangular k range (no 64-wide leaf chain per chunk):
//     Xc  = S_c K(x_c, z)                   kmat with a row scale          (CH × M, dtype T)
//     Y   = −inv(L_z) Xcᵀ = −B_c            gemm_nt_dma, beta0 + ktri      (M × CH, data points contiguous)
//     D_acc −= Y Yᵀ                         MFMA gemm (NT), fp64 accumulation (SYRK over the data points)
//     c_acc −= Y b_c,  rowss += rowsumsq(Y) ystats (one pass over Y)
//   (T = f32 or f64).  The M×M side (K_zz, both Choleskys, all vector solves) is always fp64, and so are the
//   accumulators of the N-long reductions (D_acc, c_acc, ‖B‖²_F): in fp32 mode the operands stream in fp32 through
//   the fp32 MFMA in K = 2 048 partial products that are summed into the fp64 accumulator; L_z is rounded to fp32 for
//   the streamed TRSM only.  ‖A‖²_F = tr(B Bᵀ) = Σ rowss (ELBO trace term, :251).
//
// The handle keeps what the reference keeps in its cache (:73, :118): the scaled observations x, Σy^-1/2 and b_y (small
// N-vectors; B_εf itself is never stored), so that both update_posterior forms continue from the resident state:
//   new observations   -> stream only the new points into D_acc / c_acc / rowss, re-finalise the M×M side
//   new pseudo-points  -> bordered Cholesky of K_zz (update_chol, src/util/common_covmat_ops.jl:38-42), then re-stream
//                         the retained observations computing ONLY the new block rows of B Bᵀ, B b_y and ‖B‖²
#pragma once

struct ObsSeg {  // one batch of observations, device resident: T xs[d][npad] (scaled), T rs[npad] = σ⁻¹, T b[npad] = σ⁻¹δ
    gp_ctx* ctx = nullptr;
    void *xs = nullptr, *rs = nullptr, *b = nullptr;
    long n = 0, npad = 0;
    ~ObsSeg() {  // runs under the ctx lock (gp_vfe_free / error paths of a locked call)
        ctx_release(ctx, xs, 0);
        ctx_release(ctx, rs, 0);
        ctx_release(ctx, b, 0);
    }
};

struct gp_vfe {
    gp_ctx* ctx = nullptr;
    int dtype = 0;
    long m = 0, mp = 0, ld = 0;
    int d = 0, kind = 0;
    double variance = 1, jitter = 0;
    int nscale = 0;
    std::vector<double> scale;
    void *Lz = nullptr, *Ld = nullptr;  // (mp + 256) × ld doubles: chol(K_zz + jitter I), Λ_ε factor
    void* zs = nullptr;                 // scaled inducing inputs, double [d][mp]
    void* vec = nullptr;                // double [4][mp]: rows c, m_ε, α, spare
    int approx = 0;
    long n_obs = 0;
    long chunk = 16384;  // data points per streamed chunk of THIS handle (chosen by its first fit; its updates / appends / gradient keep it: the retained segments are padded to it)
    // streaming state before the M×M finalisation: −B Bᵀ (lower), B b_y, per-row ‖B‖², inv(L_z) and scaled z in T
    void *Dacc = nullptr, *cacc = nullptr, *rowss = nullptr, *Li = nullptr, *zsT = nullptr;
    std::vector<std::shared_ptr<ObsSeg>> segs;  // every observation seen so far (shared between a posterior and its updates)
    double logdet_sy = 0, dd = 0, tr_kff = 0, trZ = 0;
    DibCache dib_z, dib_d;  // inverse diagonal blocks of Lz / Ld for the predictive solves (built on the first prediction; a new handle after every update)
};

static void vfe_release(gp_vfe* p) {  // under the ctx lock
    gp_ctx* c = p->ctx;
    ctx_release(c, p->Lz, 0);
    ctx_release(c, p->Ld, 0);
    ctx_release(c, p->zs, 0);
    ctx_release(c, p->vec, 0);
    ctx_release(c, p->Dacc, 0);
    ctx_release(c, p->cacc, 0);
    ctx_release(c, p->rowss, 0);
    ctx_release(c, p->Li, 0);
    ctx_release(c, p->zsT, 0);
    if (p->dib_z.w) ctx_release(c, p->dib_z.w, p->dib_z.bytes);
    if (p->dib_d.w) ctx_release(c, p->dib_d.w, p->dib_d.bytes);
    p->dib_z = DibCache();
    p->dib_d = DibCache();
    p->segs.clear();
}

// Iw ((mp + 256) × ld doubles, L_b bytes) ← L⁻ᵀ (upper, row-major) of the lower factor L: I · L⁻ᵀ by the restricted-row recursion.  "vfe_inv_nb" >= 128: the inverse
// diagonal blocks of L all at once (dib_build: every launch of the restricted-row recursion carries the batch; L_bb⁻ᵀ lands on Iw's diagonal), then the levels
// above them with ONE triangular-k GEMM per block — the mechanism of the gradient's L⁻ᵀ (grad_impl).  Used by the fit's prelude (inv(L_z)) and by vfe_grad_impl.
static int32_t vfe_upper_inverse(gp_ctx* c, hipStream_t s, const double* L, long ld, long mp, double* Iw, size_t L_b, DevBufs& bufs) {
    HIPCHK(hipMemsetAsync(Iw, 0, L_b, s));
    hipLaunchKernelGGL(identity_kernel<double>, dim3((unsigned)((mp + 255) / 256), (unsigned)mp), dim3(256), 0, s, Iw, ld, mp);
    HIPCHK(hipGetLastError());
    const long inb = c->vfe_inv_nb;
    if (inb >= 128 && mp >= 2 * inb) {
        const long ldw = inb + c->ldpad;
        const size_t wb = sizeof(double) * (size_t)(mp + 128) * ldw;
        void *Wn_v = 0, *Iw2_v = 0, *Sw_v = 0;
        RC(bufs.get(wb, &Wn_v));
        RC(bufs.get(wb, &Iw2_v));
        RC(bufs.get(wb, &Sw_v));
        RC(dib_build<double>(c, s, L, ld, mp, inb, (double*)Wn_v, ldw, (double*)Iw2_v, Iw, ld));
        DibArgs<double> dib;
        dib.W = (const double*)Wn_v; dib.ldw = ldw; dib.nbi = inb; dib.S = (double*)Sw_v; dib.lds = ldw;
        RC(trsm_upper_rec<double>(c, s, Iw, ld, L, ld, 0, mp, dib));
    } else {
        RC(trsm_upper_rec<double>(c, s, Iw, ld, L, ld, 0, mp));
    }
    return 0;
}

enum { VFE_FIT = 0, VFE_UPDATE = 1, VFE_APPEND = 2 };

// mode VFE_FIT:    x, y, noise, z, jitter given; prev == NULL
// mode VFE_UPDATE: x, y, noise = the NEW observations; z == NULL; kernel / jitter / z from prev
// mode VFE_APPEND: z = the NEW pseudo-points; x == NULL; everything else from prev
template <typename T>
static int32_t vfe_fit_impl(gp_ctx* c, const gp_kernel* k, const gp_points* x, const gp_points* z, const gp_noise* noise,
                            double jitter, const void* mean_or_null, const void* yv, int approx, gp_vfe* out,
                            double* objective, const gp_vfe* prev = nullptr, int mode = VFE_FIT) {
    const long m_old = prev ? prev->m : 0;
    const long m2 = (mode == VFE_APPEND) ? z->n : 0;
    const long m = mode == VFE_FIT ? z->n : m_old + m2;
    const long mp = round_up(m, 128);
    const int d = prev ? prev->d : x->d;
    // data points per streamed chunk: the handle's own once it exists; else the ctx parameter "vfe_chunk" (> 0), or automatic (0, the default): 16 384 — measured
    // best at C5 (M = 4 096) — times a power of two that keeps chunk × M at C5's footprint for fewer pseudo-points (M = 64 … 256 and N = 2·10⁶: 123 chunks of
    // 16 384 cost 13.5 / 17.0 ms per fit pass, 16 of 131 072 6.4 / ≈ 12 — launches of a few hundred µs of work each), never more than the batch itself
    long CH = prev ? prev->chunk : c->vfe_chunk;
    if (CH <= 0) {
        long f = 1;
        while (f < 16 && 2 * f * mp <= 4096) f *= 2;
        CH = std::min(16384 * f, round_up(std::max(x ? x->n : 1L, 1L), 16384));
    }
    const long KS = std::min(c->vfe_ks, CH);    // fp32: data points per fp32 partial product
    const int NBAT = (int)(CH / KS);
    if (CH % KS) return set_arg_err(1, "vfe_chunk must be a multiple of vfe_ks");
    if (c->vfe_sk) ++c->sk_scope;
    struct SkOff {
        gp_ctx* c;
        ~SkOff() {
            if (c->vfe_sk) --c->sk_scope;
        }
    } sk_off{c};
    const long n = x ? x->n : 0, npad = round_up(std::max(n, 1L), CH);
    const long ld = mp + c->ldpad;              // M×M matrices and the CH×M chunk
    const T* y = (const T*)yv;
    const T* mean = (const T*)mean_or_null;
    constexpr bool is_f64 = sizeof(T) == 8;
    if (prev) jitter = prev->jitter;

    c->ev_used = 0;
    c->gemm_recs.clear();
    for (auto& e : c->ev_phase)
        if (!e) HIPCHK(hipEventCreate(&e));
    if (!c->info_dev) HIPCHK(hipMalloc((void**)&c->info_dev, sizeof(int)));
    RC(ctx_scal(c, 16 + 128));

    // ---- host marshalling
    std::vector<T> znew_h, host_pageable;
    T *xs_h = nullptr, *rs_h = nullptr, *b_h = nullptr;   // the N-long host vectors: page-locked staging of the ctx when it is to be had (three 16 MB uploads from
                                                          // pageable memory cost ≈ 10 ms at N = 2·10⁶; C5's are 1–3 MB and ride behind the prelude either way)
    double logdet_sy = prev ? prev->logdet_sy : 0, dd = prev ? prev->dd : 0, tr_kff = prev ? prev->tr_kff : 0;
    // the N-long host marshalling (scaled inputs, Σy^-1/2, b_y, the scalar sums: ≈ 1 ms at N = 262 144) runs AFTER the M×M prelude has been queued — the
    // device factors K_zz and inverts L_z meanwhile
    auto marshal_x = [&]() -> int32_t {
        if (!x) return 0;
        const size_t cnt = (size_t)(d + 2) * (size_t)npad;
        xs_h = (T*)ctx_pinned(c, sizeof(T) * cnt);
        if (!xs_h) {
            host_pageable.resize(cnt);
            xs_h = host_pageable.data();
        }
        rs_h = xs_h + (size_t)d * npad;   // s_i = σ_i⁻¹
        b_h = rs_h + npad;                // b_i = s_i δ_i  (b_y, :66)
        scale_points_into<T>(k, x, npad, xs_h);
        std::fill(rs_h + n, rs_h + npad, T(0));
        std::fill(b_h + n, b_h + npad, T(0));
        if (noise->kind == 0) {  // Σy = σ² I: one square root, one logarithm (the general loop below spends ≈ 15 ns per observation on them: 30 ms at N = 2·10⁶)
            const double s2 = noise->s;
            if (!(s2 > 0)) return n > 0 ? 1 : 0;  // chol(Σy) fails at the first observation (reference :61 / :296)
            const double si = 1.0 / std::sqrt(s2);
            std::fill(rs_h, rs_h + n, (T)si);
            double acc = 0;
            if (mean) {
                for (long i = 0; i < n; ++i) {
                    const T bi = (T)((double)(T)(y[i] - mean[i]) * si);
                    b_h[i] = bi;
                    acc += (double)bi * (double)bi;
                }
            } else {
                for (long i = 0; i < n; ++i) {
                    const T bi = (T)((double)y[i] * si);
                    b_h[i] = bi;
                    acc += (double)bi * (double)bi;
                }
            }
            dd += acc;
            logdet_sy += (double)n * std::log(s2);
            tr_kff += (double)n * (k->variance / s2);  // tr_Cf_invΣy :307-313
            return 0;
        }
        for (long i = 0; i < n; ++i) {
            const double s2 = (double)((const T*)noise->diag)[i];
            if (!(s2 > 0)) return 1 + (int32_t)i;  // chol(Σy) fails at i (reference :61 / :296)
            const double delta = (double)(T)(y[i] - (mean ? mean[i] : T(0)));
            const double si = 1.0 / std::sqrt(s2);
            rs_h[i] = (T)si;
            b_h[i] = (T)(delta * si);
            logdet_sy += std::log(s2);
            dd += (double)b_h[i] * (double)b_h[i];
            tr_kff += k->variance / s2;  // tr_Cf_invΣy :307-313
        }
        return 0;
    };
    const long m2p = round_up(std::max(m2, 1L), 128);
    const long znp = mode == VFE_FIT ? mp : m2p;  // leading dimension of the freshly scaled inducing inputs
    if (z) scale_points<T>(k, z, znp, znew_h);
    std::vector<double> znewD_h(znew_h.begin(), znew_h.end());
    std::vector<double> jit_h((size_t)mp, 0.0);
    for (long i = 0; i < m; ++i) jit_h[i] = jitter;

    const size_t zsT_b = sizeof(T) * (size_t)d * mp, zsD_b = sizeof(double) * (size_t)d * mp;
    const size_t D_b = sizeof(double) * (size_t)(mp + 128) * ld;
    const long ldy = CH + c->ldpad;             // Y = −B_c (M × CH)
    const size_t X_b = sizeof(T) * (size_t)(CH + 128) * ld, L_b = sizeof(double) * (size_t)(mp + 128 + 128) * ld;
    const size_t Y_b = sizeof(T) * (size_t)(mp + 128) * ldy, Li_b = sizeof(T) * (size_t)(mp + 128) * ld;
    const size_t vT_b = sizeof(double) * (size_t)mp, vD_b = sizeof(double) * (size_t)mp * 4, jit_b = sizeof(double) * (size_t)mp;
    void *zsT_v = 0, *zsD_v = 0, *D_v = 0, *X_v = 0, *Lz_v = 0, *Ld_v = 0, *cT_v = 0, *rss_v = 0, *vec_v = 0, *jit_v = 0, *Y_v = 0,
         *Li_v = 0, *I_v = 0, *S_v = 0, *zn_v = 0, *znD_v = 0, *X2_v = 0, *Y2_v = 0, *S2_v = 0;
    DevBufs bufs(c);
    std::shared_ptr<ObsSeg> seg;
    if (x) {
        seg = std::make_shared<ObsSeg>();
        seg->ctx = c;
        seg->n = n;
        seg->npad = npad;
        RC(ctx_alloc(c, sizeof(T) * (size_t)d * (size_t)npad, &seg->xs));
        RC(ctx_alloc(c, sizeof(T) * (size_t)npad, &seg->rs));
        RC(ctx_alloc(c, sizeof(T) * (size_t)npad, &seg->b));
    }
    RC(bufs.get(zsT_b, &zsT_v));
    RC(bufs.get(zsD_b, &zsD_v));
    RC(bufs.get(D_b, &D_v));
    RC(bufs.get(X_b, &X_v));
    RC(bufs.get(L_b, &Lz_v));
    RC(bufs.get(L_b, &Ld_v));
    RC(bufs.get(Y_b, &Y_v));
    RC(bufs.get(Li_b, &Li_v));
    if (mode != VFE_UPDATE) RC(bufs.get(L_b, &I_v));
    if (!is_f64) RC(bufs.get((size_t)NBAT * Li_b, &S_v));
    if (c->vfe_overlap) {  // second set of chunk buffers for the overlapped helpers
        RC(bufs.get(X_b, &X2_v));
        RC(bufs.get(Y_b, &Y2_v));
        if (!is_f64) RC(bufs.get((size_t)NBAT * Li_b, &S2_v));
    }
    RC(bufs.get(vT_b, &cT_v));
    RC(bufs.get(vT_b, &rss_v));
    RC(bufs.get(vD_b, &vec_v));
    RC(bufs.get(jit_b, &jit_v));
    if (z && mode == VFE_APPEND) {
        RC(bufs.get(sizeof(T) * znew_h.size(), &zn_v));
        RC(bufs.get(sizeof(double) * znewD_h.size(), &znD_v));
    }
    double* Lz = (double*)Lz_v;
    double* Ld = (double*)Ld_v;
    double* vec = (double*)vec_v;  // rows: [0] c , [1] w→m_ε , [2] α , [3] spare
    double scal_h[16] = {0};
    std::vector<double> rss_h((size_t)mp, 0.0);
    int info_h = 0;
    hipStream_t s = c->sm;
    if (prev && mode == VFE_UPDATE && (prev->mp != mp || prev->ld != ld)) return set_arg_err(1, "stale gp_vfe state");
    for (auto& sg : (prev ? prev->segs : std::vector<std::shared_ptr<ObsSeg>>()))
        if (sg->npad % CH) return set_arg_err(1, "gp_vfe was built with a different vfe_chunk");
    const long n_all = (prev ? prev->n_obs : 0) + n;
    // rows of D_acc / c_acc / rowss that the streamed pass (re)computes: everything for a fit / new observations (the
    // accumulators continue), the block rows from the first new pseudo-point's 128-tile on for an append
    const long row_lo = (mode == VFE_APPEND) ? (m_old / 128) * 128 : 0;

    // One chunk loop over a batch of observations: rows >= row_lo of D_acc, c_acc, rowss.  The two MFMA GEMMs of a chunk run
    // back to back on the main stream; the bandwidth-bound helpers run on the ctx's second stream beside them, on double
    // buffers: kmat of chunk c+1 and the partial-sum adds of chunk c overlap the GEMMs of their neighbours ("vfe_overlap").
    hipStream_t sa = c->vfe_overlap ? c->sp : s;
    const bool ovl = sa != s;
    // "vfe_dual" (experiment switch, default 0): the triangular products Y(c) = −inv(L_z) X(c)ᵀ on a third, HIGH-priority stream sy, the chunk SYRKs on the main
    // stream — Y(c+1) is independent of SYRK(c), and the idea was that each launch's last partial round of workgroups (SYRK: 528 lower tiles × 8 partials = 4 224
    // workgroups = 8.25 rounds of the 512 slots) is filled by the other launch.  Measured at C5 in one process (tools/c5_ab.py, profiles/r6/c5_ab*.jsonl): with
    // the priority +0.4…0.7 ms, with both launches at equal priority −0.45 ms of 79.5 — the single-stream pass has no idle tail worth filling (the helper
    // kernels of the second stream already run in it).  Same buffers, same arithmetic, same order of the sums into D_acc either way.
    hipStream_t sy = s;
    if (ovl && c->vfe_dual) RC(ctx_third_stream(c, &sy));
    const bool dual = sy != s;
    hipEvent_t evSy[2] = {nullptr, nullptr};  // SYRK(c) has finished reading Y[bb]
    void* Xb[2] = {X_v, X2_v};
    void* Yb[2] = {Y_v, Y2_v};
    void* Sb[2] = {S_v, S2_v};
    hipEvent_t evK[2] = {nullptr, nullptr}, evYs[2] = {nullptr, nullptr}, evAdd[2] = {nullptr, nullptr};
    hipEvent_t ev_z = nullptr;      // the scaled pseudo-inputs are on the device (fit mode)
    bool first_kmat_done = false;   // chunk 0's kmat was issued beside the prelude
    long cidx = 0;
    auto kmat_chunk = [&](const ObsSeg& sg, long c0, int bb) -> int32_t {
        GridMap g = plain_map(0, c0, 0);
        dim3 grid((unsigned)(mp / 128), (unsigned)(CH / 128));
        launch_kmat<T>(grid, sa, (T*)Xb[bb], ld, (const T*)sg.xs, sg.npad, (const T*)zsT_v, mp, d,
                           k->kind, (T)k->variance, (const T*)nullptr, sg.n, m, 0, g, (const T*)nullptr, (const T*)sg.rs);
        HIPCHK(hipGetLastError());
        if (ovl) {
            RC(ctx_event(c, &evK[bb], false));
            HIPCHK(hipEventRecord(evK[bb], sa));
        }
        return 0;
    };
    auto stream_seg = [&](const ObsSeg& sg, const ObsSeg* next_sg) -> int32_t {
        for (long c0 = 0; c0 < sg.npad; c0 += CH) {
            if (c0 >= sg.n) break;  // nothing but padding in the remaining chunks
            const int bb = ovl ? (int)(cidx & 1) : 0;
            if (!ovl || (cidx == 0 && !first_kmat_done)) RC(kmat_chunk(sg, c0, bb));  // (overlapped mode: later chunks were prefetched below, the first beside the prelude)
            if (ovl) {
                HIPCHK(hipStreamWaitEvent(sy, evK[bb], 0));
                if (evYs[bb]) HIPCHK(hipStreamWaitEvent(sy, evYs[bb], 0));  // chunk cidx−2 is done with Y[bb]
                if (dual && evSy[bb]) HIPCHK(hipStreamWaitEvent(sy, evSy[bb], 0));  // ... its SYRK too
            }
            {
                GridMap gy = plain_map(0, 0, 0);
                gy.beta0 = 1;
                gy.ktri = 1;
                RC(launch_gemm<T>(c, sy, (T*)Yb[bb], ldy, (const T*)Li_v, ld, (const T*)Xb[bb], ld, mp, CH, mp, gy));   // Y = −B_c
            }
            if (ovl) {
                hipEvent_t e;
                RC(ctx_event(c, &e, false));
                HIPCHK(hipEventRecord(e, sy));
                HIPCHK(hipStreamWaitEvent(sa, e, 0));
                if (dual) HIPCHK(hipStreamWaitEvent(s, e, 0));
            }
            hipLaunchKernelGGL(ystats_kernel<T>, dim3((unsigned)(mp - row_lo)), dim3(256), 0, sa, (const T*)Yb[bb], ldy, CH,
                               (const T*)sg.b + c0, row_lo, (double*)cT_v, (double*)rss_v);   // c += B_c b_c ; ‖B‖² rows (fp64)
            HIPCHK(hipGetLastError());
            if (ovl) {
                RC(ctx_event(c, &evYs[bb], false));
                HIPCHK(hipEventRecord(evYs[bb], sa));
                // prefetch the next chunk's S·K(x_c, z) into the other X buffer (its last reader, the previous Y GEMM, is done)
                const long c1 = c0 + CH;
                if (c1 < sg.npad && c1 < sg.n) RC(kmat_chunk(sg, c1, bb ^ 1));
                else if (next_sg) RC(kmat_chunk(*next_sg, 0, bb ^ 1));
            }
            const T* Yr = (const T*)Yb[bb] + row_lo * ldy;
            if constexpr (is_f64) {
                RC((launch_gemm<T, double>(c, s, (double*)D_v + row_lo * ld, ld, Yr, ldy, (const T*)Yb[bb], ldy, mp - row_lo, mp, CH,
                                           plain_map(1, row_lo, 0))));
                if (dual) {
                    RC(ctx_event(c, &evSy[bb], false));
                    HIPCHK(hipEventRecord(evSy[bb], s));
                }
            } else {
                // fp32: the chunk's SYRK runs on the LDS-DMA kernel into fp32 scratch — NBAT partial products over KS data points
                // each in ONE launch (blockIdx.z) — which one pass then adds into the fp64 accumulator: fp64 sums across
                // partials and chunks, fp32 MFMA within a partial
                GridMap gs = plain_map(1, row_lo, 0);
                gs.beta0 = 1;
                gs.nbatch = NBAT;
                gs.cstride = (long)(mp + 128) * ld;
                if (ovl && evAdd[bb]) HIPCHK(hipStreamWaitEvent(s, evAdd[bb], 0));  // chunk cidx−2's partials have been added
                RC(launch_gemm<T>(c, s, (T*)Sb[bb] + row_lo * ld, ld, Yr, ldy, (const T*)Yb[bb], ldy, mp - row_lo, mp, KS, gs));
                if (ovl) {
                    hipEvent_t e;
                    RC(ctx_event(c, &e, false));
                    HIPCHK(hipEventRecord(e, s));
                    HIPCHK(hipStreamWaitEvent(sa, e, 0));
                    if (dual) evSy[bb] = e;
                }
                hipLaunchKernelGGL(add_lower_batched_kernel<T>, dim3((unsigned)((mp + 1023) / 1024), (unsigned)(mp - row_lo)), dim3(256),
                                   0, sa, (const T*)Sb[bb], gs.cstride, NBAT, ld, (double*)D_v, ld, mp, row_lo);
                HIPCHK(hipGetLastError());
                if (ovl) {
                    RC(ctx_event(c, &evAdd[bb], false));
                    HIPCHK(hipEventRecord(evAdd[bb], sa));
                }
            }
            ++cidx;
        }
        return 0;
    };
    auto stream_join = [&]() -> int32_t {  // everything the helpers accumulated is visible to the main stream
        if (!ovl) return 0;
        hipEvent_t e;
        RC(ctx_event(c, &e, false));
        HIPCHK(hipEventRecord(e, sa));
        HIPCHK(hipStreamWaitEvent(s, e, 0));
        if (dual) {
            RC(ctx_event(c, &e, false));
            HIPCHK(hipEventRecord(e, sy));
            HIPCHK(hipStreamWaitEvent(s, e, 0));
        }
        return 0;
    };

    int32_t rc = [&]() -> int32_t {
        HIPCHK(hipEventRecord(c->ev_phase[0], s));
        HIPCHK(hipMemcpyAsync(jit_v, jit_h.data(), jit_b, hipMemcpyHostToDevice, s));
        HIPCHK(hipMemsetAsync(c->info_dev, 0, sizeof(int), s));
        HIPCHK(hipMemsetAsync(c->scal_dev, 0, sizeof(double) * 16, s));
        HIPCHK(hipMemsetAsync(D_v, 0, D_b, s));
        HIPCHK(hipMemsetAsync(cT_v, 0, vT_b, s));
        HIPCHK(hipMemsetAsync(rss_v, 0, vT_b, s));
        HIPCHK(hipMemsetAsync(vec_v, 0, vD_b, s));
        if (mode == VFE_UPDATE) {  // resume: factor of K_zz, its inverse, the scaled inducing inputs and the running sums come from prev
            HIPCHK(hipMemcpyAsync(Lz_v, prev->Lz, L_b, hipMemcpyDeviceToDevice, s));
            HIPCHK(hipMemcpyAsync(Li_v, prev->Li, Li_b, hipMemcpyDeviceToDevice, s));
            HIPCHK(hipMemcpyAsync(zsT_v, prev->zsT, zsT_b, hipMemcpyDeviceToDevice, s));
            HIPCHK(hipMemcpyAsync(zsD_v, prev->zs, zsD_b, hipMemcpyDeviceToDevice, s));
            HIPCHK(hipMemcpyAsync(D_v, prev->Dacc, D_b, hipMemcpyDeviceToDevice, s));
            HIPCHK(hipMemcpyAsync(cT_v, prev->cacc, vT_b, hipMemcpyDeviceToDevice, s));
            HIPCHK(hipMemcpyAsync(rss_v, prev->rowss, vT_b, hipMemcpyDeviceToDevice, s));
        } else {
            if (mode == VFE_FIT) {
                HIPCHK(hipMemcpyAsync(zsT_v, znew_h.data(), zsT_b, hipMemcpyHostToDevice, s));
                HIPCHK(hipMemcpyAsync(zsD_v, znewD_h.data(), zsD_b, hipMemcpyHostToDevice, s));
                if (ovl) {  // the first chunk's K(x_c, z) needs the pseudo-inputs only: the helper stream may start it while the prelude runs
                    RC(ctx_event(c, &ev_z, false));
                    HIPCHK(hipEventRecord(ev_z, s));
                }
                // ---- L_z = chol(K_zz + jitter I), fp64                                                :62
                GridMap g = plain_map(1, 0, 0);
                dim3 grid((unsigned)(mp / 128), (unsigned)(mp / 128));
                launch_kmat<double>(grid, s, Lz, ld, (const double*)zsD_v, mp,
                                   (const double*)zsD_v, mp, d, k->kind, k->variance, (const double*)jit_v, m, m, 1, g,
                                   (const double*)nullptr, (const double*)nullptr);
                HIPCHK(hipGetLastError());
                RC(potrf_full<double>(c, Lz, ld, mp, mp, c->info_dev, m, c->scal_dev + 0));
            } else {  // VFE_APPEND: z = vcat(z_old, z_new); U = update_chol(U11, C12, C22)               :131-141
                const long mpo = prev->mp, ldo = prev->ld;
                HIPCHK(hipMemcpyAsync(zn_v, znew_h.data(), sizeof(T) * znew_h.size(), hipMemcpyHostToDevice, s));
                HIPCHK(hipMemcpyAsync(znD_v, znewD_h.data(), sizeof(double) * znewD_h.size(), hipMemcpyHostToDevice, s));
                HIPCHK(hipMemsetAsync(zsT_v, 0, zsT_b, s));
                HIPCHK(hipMemsetAsync(zsD_v, 0, zsD_b, s));
                HIPCHK(hipMemcpy2DAsync(zsT_v, sizeof(T) * mp, prev->zsT, sizeof(T) * mpo, sizeof(T) * m_old, d, hipMemcpyDeviceToDevice, s));
                HIPCHK(hipMemcpy2DAsync((T*)zsT_v + m_old, sizeof(T) * mp, zn_v, sizeof(T) * m2p, sizeof(T) * m2, d, hipMemcpyDeviceToDevice, s));
                HIPCHK(hipMemcpy2DAsync(zsD_v, sizeof(double) * mp, prev->zs, sizeof(double) * mpo, sizeof(double) * m_old, d, hipMemcpyDeviceToDevice, s));
                HIPCHK(hipMemcpy2DAsync((double*)zsD_v + m_old, sizeof(double) * mp, znD_v, sizeof(double) * m2p, sizeof(double) * m2, d, hipMemcpyDeviceToDevice, s));
                // X = K(z_new, z_old) L11⁻ᵀ = U12ᵀ (in Ld's storage, free until the finalisation), S = C22 − U12ᵀU12 (in I's)
                double* Xb = Ld;
                double* Sb = (double*)I_v;
                const long ldx = ldo, lds = m2p + c->ldpad;
                {
                    GridMap g = plain_map(0, 0, 0);
                    dim3 grid((unsigned)(mpo / 128), (unsigned)(m2p / 128));
                    launch_kmat<double>(grid, s, Xb, ldx, (const double*)znD_v, m2p,
                                       (const double*)prev->zs, mpo, d, k->kind, k->variance, (const double*)nullptr, m2, m_old, 0, g,
                                       (const double*)nullptr, (const double*)nullptr);
                    HIPCHK(hipGetLastError());
                }
                RC(trsm_rec<double>(c, s, Xb, ldx, m2p, (const double*)prev->Lz, ldo, mpo));
                {
                    GridMap g = plain_map(1, 0, 0);
                    dim3 grid((unsigned)(m2p / 128), (unsigned)(m2p / 128));
                    // C22 = _symmetric(cov(prior, z_new)) carries NO jitter in the reference (src/sparse_approximations.jl:138)
                    launch_kmat<double>(grid, s, Sb, lds, (const double*)znD_v, m2p,
                                       (const double*)znD_v, m2p, d, k->kind, k->variance, (const double*)nullptr, m2, m2, 1, g,
                                       (const double*)nullptr, (const double*)nullptr);
                    HIPCHK(hipGetLastError());
                }
                RC(launch_gemm<double>(c, s, Sb, lds, Xb, ldx, Xb, ldx, m2p, m2p, mpo, plain_map(1, 0, 0)));
                RC(potrf_full<double>(c, Sb, lds, m2p, m2p, c->info_dev, m2, c->scal_dev + 0));
                HIPCHK(hipMemsetAsync(Lz_v, 0, L_b, s));
                HIPCHK(hipMemcpy2DAsync(Lz, sizeof(double) * ld, prev->Lz, sizeof(double) * ldo, sizeof(double) * m_old, m_old, hipMemcpyDeviceToDevice, s));
                HIPCHK(hipMemcpy2DAsync(Lz + m_old * ld, sizeof(double) * ld, Xb, sizeof(double) * ldx, sizeof(double) * m_old, m2, hipMemcpyDeviceToDevice, s));
                HIPCHK(hipMemcpy2DAsync(Lz + m_old * ld + m_old, sizeof(double) * ld, Sb, sizeof(double) * lds, sizeof(double) * m2, m2, hipMemcpyDeviceToDevice, s));
                if (mp > m) {
                    hipLaunchKernelGGL(pad_identity_kernel<double>, dim3((unsigned)((mp + 255) / 256), (unsigned)(mp - m)), dim3(256), 0, s,
                                       Lz, ld, m, mp);
                    HIPCHK(hipGetLastError());
                }
                // the finished block rows of the accumulators carry over (rows < row_lo; D_acc is lower, so columns < row_lo too)
                if (row_lo > 0) {
                    HIPCHK(hipMemcpy2DAsync(D_v, sizeof(double) * ld, prev->Dacc, sizeof(double) * ldo, sizeof(double) * row_lo, row_lo, hipMemcpyDeviceToDevice, s));
                    HIPCHK(hipMemcpyAsync(cT_v, prev->cacc, sizeof(double) * row_lo, hipMemcpyDeviceToDevice, s));
                    HIPCHK(hipMemcpyAsync(rss_v, prev->rowss, sizeof(double) * row_lo, hipMemcpyDeviceToDevice, s));
                }
            }
            // ---- inv(L_z): W = I · L_z⁻ᵀ (upper), transposed into Ld's storage (free until the SYRK is done), rounded to T
            double* Iw = (double*)I_v;
            RC(vfe_upper_inverse(c, s, Lz, ld, mp, Iw, L_b, bufs));
            hipLaunchKernelGGL(transpose_f64_kernel, dim3((unsigned)(mp / 32), (unsigned)(mp / 32)), dim3(256), 0, s, Iw, ld, Ld, ld, mp);
            HIPCHK(hipGetLastError());
            HIPCHK(hipMemsetAsync(Li_v, 0, Li_b, s));
            hipLaunchKernelGGL((convert_kernel<double, T>), convert_grid((long)mp * ld), dim3(256), 0, s, Ld,
                               (T*)Li_v, (long)mp * ld, 1.0);
            HIPCHK(hipGetLastError());
        }
        // ---- the observations: marshalled on the host while the device works through the prelude queued above, uploaded on the helper stream (the first
        //      consumers — kmat, ystats — run there; the main stream joins through their events)
        RC(marshal_x());
        if (x) {
            HIPCHK(hipMemcpyAsync(seg->xs, xs_h, sizeof(T) * (size_t)d * (size_t)npad, hipMemcpyHostToDevice, sa));
            HIPCHK(hipMemcpyAsync(seg->rs, rs_h, sizeof(T) * (size_t)npad, hipMemcpyHostToDevice, sa));
            HIPCHK(hipMemcpyAsync(seg->b, b_h, sizeof(T) * (size_t)npad, hipMemcpyHostToDevice, sa));
        }
        if (ev_z && x && mode == VFE_FIT && n > 0) {  // chunk 0's kmat beside the prelude (0.34 ms at C5); everything else of the helper stream waits for the prelude below
            HIPCHK(hipStreamWaitEvent(sa, ev_z, 0));
            RC(kmat_chunk(*seg, 0, 0));
            first_kmat_done = true;
        }
        HIPCHK(hipEventRecord(c->ev_phase[1], s));
        // ---- streamed pass over the data points                                                   :64-71
        if (ovl) {  // the helper stream starts after the M×M prelude and the uploads
            hipEvent_t e;
            RC(ctx_event(c, &e, false));
            HIPCHK(hipEventRecord(e, s));
            HIPCHK(hipStreamWaitEvent(sa, e, 0));
            if (dual) HIPCHK(hipStreamWaitEvent(sy, e, 0));
        }
        if (mode == VFE_APPEND) {
            for (size_t si = 0; si < prev->segs.size(); ++si)
                RC(stream_seg(*prev->segs[si], si + 1 < prev->segs.size() ? prev->segs[si + 1].get() : nullptr));
        } else {
            RC(stream_seg(*seg, nullptr));
        }
        RC(stream_join());
        HIPCHK(hipEventRecord(c->ev_phase[2], s));
        hipLaunchKernelGGL((convert_kernel<double, double>), convert_grid(mp), dim3(256), 0, s,
                           (const double*)cT_v, vec, mp, 1.0);                                                              // c = B b_y
        HIPCHK(hipGetLastError());
        // ---- D = I + B Bᵀ (fp64, symmetric), Λ_ε = chol(D)                                       :68-69
        hipLaunchKernelGGL(neg_sym_to_f64_kernel<double>, dim3((unsigned)((mp + 255) / 256), (unsigned)mp), dim3(256), 0, s,
                           (const double*)D_v, ld, Ld, ld, mp);
        HIPCHK(hipGetLastError());
        hipLaunchKernelGGL(diag_shift_trace_kernel, dim3(1), dim3(256), 0, s, Ld, ld, mp, m, 1.0, c->scal_dev + 2);
        HIPCHK(hipGetLastError());
        RC(potrf_full<double>(c, Ld, ld, mp, mp, c->info_dev + 0, mp, c->scal_dev + 1));
        // ---- vectors
        HIPCHK(hipMemcpyAsync(vec + mp, vec, sizeof(double) * mp, hipMemcpyDeviceToDevice, s));
        RC(trsv<double>(c, s, Ld, ld, mp, vec + mp, mp, 1, true));                // w = L_D⁻¹ c
        hipLaunchKernelGGL(rowsumsq_kernel<double>, dim3(1), dim3(256), 0, s, vec + mp, mp, mp, c->scal_dev + 3);
        HIPCHK(hipGetLastError());
        RC(trsv<double>(c, s, Ld, ld, mp, vec + mp, mp, 1, false));               // m_ε = L_D⁻ᵀ w      :71
        HIPCHK(hipMemcpyAsync(vec + 2 * mp, vec + mp, sizeof(double) * mp, hipMemcpyDeviceToDevice, s));
        RC(trsv<double>(c, s, Lz, ld, mp, vec + 2 * mp, mp, 1, false));           // α = L_z⁻ᵀ m_ε      :73
        HIPCHK(hipEventRecord(c->ev_phase[3], s));
        HIPCHK(hipMemcpyAsync(&info_h, c->info_dev, sizeof(int), hipMemcpyDeviceToHost, s));
        HIPCHK(hipMemcpyAsync(scal_h, c->scal_dev, sizeof(double) * 16, hipMemcpyDeviceToHost, s));
        HIPCHK(hipMemcpyAsync(rss_h.data(), rss_v, vT_b, hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        float ms;
        HIPCHK(hipEventElapsedTime(&ms, c->ev_phase[0], c->ev_phase[1]));
        c->tm.assemble_ms = ms;  // the M×M prelude: K_zz, its Cholesky, inv(L_z)
        HIPCHK(hipEventElapsedTime(&ms, c->ev_phase[1], c->ev_phase[2]));
        c->tm.potrf_ms = ms;     // the streamed pass over the data points
        HIPCHK(hipEventElapsedTime(&ms, c->ev_phase[2], c->ev_phase[3]));
        c->tm.solve_ms = ms;     // the M×M side after it: Λ_ε = chol(I + B Bᵀ), the vector solves
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
        }
        return 0;
    }();
    if (rc != 0) {
        (void)hipStreamSynchronize(c->sm);
        (void)hipStreamSynchronize(c->sp);
        return rc;
    }
    // a failing factorisation: K_zz (+ the bordered block of an append) or Λ_ε — reported like the reference's cholesky calls
    if (info_h != 0) return (mode == VFE_APPEND && info_h <= m2) ? (int32_t)(m_old + info_h) : info_h;
    // objective: dtc = -½ (N log2π + logdet Σy + logdet Λ_ε + ‖δ_s‖² − ‖Λ_ε.U⁻ᵀ A δ_s‖²)          :302-303
    //            elbo = dtc − ½ (tr(K_ff Σy⁻¹) − ‖A‖²_F),  ‖A‖²_F = ‖B‖²_F                          :251
    double trZ = 0;
    for (long i = 0; i < m; ++i) trZ += rss_h[i];
    const double logdet_lam = 2.0 * scal_h[1], quad = scal_h[3];
    double obj = -0.5 * ((double)n_all * LOG2PI + logdet_sy + logdet_lam + dd - quad);
    if (approx == 0) obj -= 0.5 * (tr_kff - trZ);
    if (objective) *objective = obj;
    if (out) {
        out->ctx = c;
        out->dtype = k->dtype;
        out->m = m; out->mp = mp; out->ld = ld; out->d = d; out->kind = k->kind;
        out->variance = k->variance; out->nscale = k->nscale; out->jitter = jitter;
        out->scale.clear();
        if (k->scale && k->nscale > 0) out->scale.assign(k->scale, k->scale + k->nscale);
        out->Lz = bufs.keep(Lz_v);
        out->Ld = bufs.keep(Ld_v);
        out->zs = bufs.keep(zsD_v);
        out->vec = bufs.keep(vec_v);
        out->approx = approx;
        out->n_obs = n_all;
        out->chunk = CH;
        out->Dacc = bufs.keep(D_v);
        out->cacc = bufs.keep(cT_v);
        out->rowss = bufs.keep(rss_v);
        out->Li = bufs.keep(Li_v);
        out->zsT = bufs.keep(zsT_v);
        out->segs.clear();
        if (prev) out->segs = prev->segs;
        if (seg) out->segs.push_back(seg);
        out->logdet_sy = logdet_sy; out->dd = dd; out->tr_kff = tr_kff; out->trZ = trZ;
    }
    return 0;
}

// Joint predictive distribution of the approximate posterior at x* on the device (always fp64 on the M side):
//   A = U⁻ᵀ K_z*  ->  rows X1 = K_*z L_z⁻ᵀ,  X2 = X1 L_D⁻ᵀ;   cov = K** − X1 X1ᵀ + X2 X2ᵀ (+ Σy*)        :187-190, :205-210
template <typename T>
static int32_t vfe_joint(gp_vfe* p, const gp_points* xs, const void* pm, const gp_noise* noise, long R, DevBufs& bufs,
                         Joint<double>& J) {
    gp_ctx* c = p->ctx;
    SkScope sk(c);
    const long m = p->m, mp = p->mp, ld = p->ld;
    const long ns = xs->n, nsp = round_up(ns, 128);
    const int d = p->d;
    gp_kernel k{};
    k.kind = p->kind; k.dtype = 0; k.variance = p->variance; k.nscale = p->nscale;
    k.scale = p->scale.empty() ? nullptr : p->scale.data();
    std::vector<T> xsT;
    scale_points<T>(&k, xs, nsp, xsT);  // inputs arrive in T; scale in T (as the fit did), then widen
    std::vector<double> xs_h(xsT.begin(), xsT.end()), nz_h;
    noise_to<double, T>(noise, ns, nsp, nz_h);
    const double* vec = (const double*)p->vec;
    const long ldc = nsp + c->ldpad;
    const size_t X_b = sizeof(double) * (size_t)(nsp + 128) * ld;
    void *xs_v = 0, *nz_v = 0, *m_v = 0, *X1_v = 0, *X2_v = 0, *X2n_v = 0;
    RC(bufs.get(sizeof(double) * xs_h.size(), &xs_v));
    RC(bufs.get(sizeof(double) * (size_t)nsp, &nz_v));
    RC(bufs.get(sizeof(double) * (size_t)nsp, &m_v));
    RC(bufs.get(X_b, &X1_v));
    RC(bufs.get(X_b, &X2_v));
    RC(bufs.get(X_b, &X2n_v));
    RC(bufs.get(sizeof(double) * (size_t)(nsp + R + 128) * ldc, &J.C));
    J.ns = ns; J.nsp = nsp; J.ld = ldc; J.R = R;
    double* X1 = (double*)X1_v;
    double* X2 = (double*)X2_v;
    double* Cm = (double*)J.C;
    hipStream_t s = c->sm;
    c->ev_used = 0;
    c->gemm_recs.clear();
    std::vector<double> m_h((size_t)ns);
    HIPCHK(hipMemcpyAsync(xs_v, xs_h.data(), sizeof(double) * xs_h.size(), hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(nz_v, nz_h.data(), sizeof(double) * (size_t)nsp, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(kvec_kernel<double>, dim3((unsigned)ns), dim3(256), 0, s, (const double*)xs_v, nsp, (const double*)p->zs, mp, d,
                       p->kind, p->variance, m, vec + 2 * mp, (double*)m_v);                                     // K_*z α
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(m_h.data(), m_v, sizeof(double) * (size_t)ns, hipMemcpyDeviceToHost, s));
    {
        GridMap g = plain_map(0, 0, 0);
        dim3 grid((unsigned)(mp / 128), (unsigned)(nsp / 128));
        launch_kmat<double>(grid, s, X1, ld, (const double*)xs_v, nsp, (const double*)p->zs, mp, d,
                           p->kind, p->variance, (const double*)nullptr, ns, m, 0, g, (const double*)nullptr, (const double*)nullptr);
        HIPCHK(hipGetLastError());
    }
    RC(trsm_cached<double>(c, s, X1, ld, nsp, (const double*)p->Lz, ld, mp, p->m, p->dib_z, bufs));
    HIPCHK(hipMemcpyAsync(X2_v, X1_v, X_b, hipMemcpyDeviceToDevice, s));
    RC(trsm_cached<double>(c, s, X2, ld, nsp, (const double*)p->Ld, ld, mp, p->m, p->dib_d, bufs));
    {
        const long cnt = (long)(nsp + 128) * ld;
        hipLaunchKernelGGL((convert_kernel<double, double>), convert_grid(cnt), dim3(256), 0, s, (const double*)X2,
                           (double*)X2n_v, cnt, -1.0);
        HIPCHK(hipGetLastError());
    }
    {
        GridMap g = plain_map(1, 0, 0);
        dim3 grid((unsigned)(nsp / 128), (unsigned)(nsp / 128));
        launch_kmat<double>(grid, s, Cm, ldc, (const double*)xs_v, nsp, (const double*)xs_v, nsp, d,
                           p->kind, p->variance, (const double*)nz_v, ns, ns, 1, g, (const double*)nullptr, (const double*)nullptr);
        HIPCHK(hipGetLastError());
    }
    HIPCHK(hipMemsetAsync(Cm + nsp * ldc, 0, sizeof(double) * (size_t)(R + 128) * ldc, s));
    RC(launch_gemm<double>(c, s, Cm, ldc, X1, ld, X1, ld, nsp, nsp, mp, plain_map(1, 0, 0)));                     // − AᵀA
    RC(launch_gemm<double>(c, s, Cm, ldc, (const double*)X2n_v, ld, X2, ld, nsp, nsp, mp, plain_map(1, 0, 0)));  // + (Λ_ε.U⁻ᵀA)ᵀ(·)
    HIPCHK(hipStreamSynchronize(s));
    const T* prior = (const T*)pm;
    J.mean.resize((size_t)ns);
    for (long i = 0; i < ns; ++i) J.mean[i] = (prior ? (double)prior[i] : 0.0) + m_h[i];
    return 0;
}

template <typename T>
static int32_t vfe_predict_impl(gp_vfe* p, const gp_points* xs, const void* pm, int what, void* mean_out, void* var_out,
                                void* cov_out) {
    gp_ctx* c = p->ctx;
    if (what & 4) {  // full covariance (+ mean): the joint, copied out and mirrored on the host                :187-190, :205-210
        DevBufs bufs(c);
        Joint<double> J;
        int32_t rc = vfe_joint<T>(p, xs, pm, nullptr, 0, bufs, J);
        if (rc != 0) {
            (void)hipStreamSynchronize(c->sm);
            return rc;
        }
        const long ns = J.ns;
        std::vector<double> Ch((size_t)ns * ns);
        HIPCHK(hipMemcpy2DAsync(Ch.data(), sizeof(double) * ns, J.C, sizeof(double) * J.ld, sizeof(double) * ns, ns, hipMemcpyDeviceToHost,
                                c->sm));
        HIPCHK(hipStreamSynchronize(c->sm));
        T* co = (T*)cov_out;
        for (long i = 0; i < ns; ++i)
            for (long j = 0; j <= i; ++j) co[i + j * ns] = co[j + i * ns] = (T)Ch[(size_t)i * ns + j];
        if (what & 1)
            for (long i = 0; i < ns; ++i) ((T*)mean_out)[i] = (T)J.mean[i];
        if (what & 2)
            for (long i = 0; i < ns; ++i) ((T*)var_out)[i] = (T)Ch[(size_t)i * ns + i];
        return 0;
    }
    const long m = p->m, mp = p->mp, ld = p->ld;
    const long ns = xs->n, nsp = round_up(ns, 128);
    const int d = p->d;
    gp_kernel k{};
    k.kind = p->kind; k.dtype = 0; k.variance = p->variance; k.nscale = p->nscale;
    k.scale = p->scale.empty() ? nullptr : p->scale.data();
    // inputs arrive in T; scale in T (as the fit did), then widen
    std::vector<T> xsT;
    scale_points<T>(&k, xs, nsp, xsT);
    std::vector<double> xs_h(xsT.begin(), xsT.end());
    const double* vec = (const double*)p->vec;
    const size_t xs_b = sizeof(double) * xs_h.size(), X_b = sizeof(double) * (size_t)(nsp + 128) * ld,
                 o_b = sizeof(double) * (size_t)nsp * 3;
    void *xs_v = 0, *X_v = 0, *o_v = 0;
    DevBufs bufs(c);
    RC(bufs.get(xs_b, &xs_v));
    RC(bufs.get(X_b, &X_v));
    RC(bufs.get(o_b, &o_v));
    double* X = (double*)X_v;
    double* o = (double*)o_v;
    hipStream_t s = c->sm;
    c->ev_used = 0;
    c->gemm_recs.clear();
    std::vector<double> oh((size_t)nsp * 3, 0.0);
    int32_t rc = [&]() -> int32_t {
        HIPCHK(hipMemcpyAsync(xs_v, xs_h.data(), xs_b, hipMemcpyHostToDevice, s));
        if (what & 1) {  // mean = m(x*) + K_*z α                                                  :183-185
            hipLaunchKernelGGL(kvec_kernel<double>, dim3((unsigned)ns), dim3(256), 0, s, (const double*)xs_v, nsp,
                               (const double*)p->zs, mp, d, p->kind, p->variance, m, vec + 2 * mp, o);
            HIPCHK(hipGetLastError());
        }
        if (what & 2) {  // var = k** − ‖A‖²_col + ‖Λ_ε.U⁻ᵀ A‖²_col, Aᵀ = K_*z L_z⁻ᵀ                 :192-195
            GridMap g = plain_map(0, 0, 0);
            dim3 grid((unsigned)(mp / 128), (unsigned)(nsp / 128));
            launch_kmat<double>(grid, s, X, ld, (const double*)xs_v, nsp,
                               (const double*)p->zs, mp, d, p->kind, p->variance, (const double*)nullptr, ns, m, 0, g,
                               (const double*)nullptr, (const double*)nullptr);
            HIPCHK(hipGetLastError());
            RC(trsm_cached<double>(c, s, X, ld, nsp, (const double*)p->Lz, ld, mp, p->m, p->dib_z, bufs));
            hipLaunchKernelGGL(rowsumsq_kernel<double>, dim3((unsigned)nsp), dim3(256), 0, s, X, ld, mp, o + nsp);
            HIPCHK(hipGetLastError());
            RC(trsm_cached<double>(c, s, X, ld, nsp, (const double*)p->Ld, ld, mp, p->m, p->dib_d, bufs));
            hipLaunchKernelGGL(rowsumsq_kernel<double>, dim3((unsigned)nsp), dim3(256), 0, s, X, ld, mp, o + 2 * nsp);
            HIPCHK(hipGetLastError());
        }
        HIPCHK(hipMemcpyAsync(oh.data(), o, o_b, hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        return 0;
    }();
    if (rc != 0) {
        (void)hipStreamSynchronize(s);
        return rc;
    }
    const T* prior = (const T*)pm;
    if (what & 1)
        for (long i = 0; i < ns; ++i) ((T*)mean_out)[i] = (T)((prior ? (double)prior[i] : 0.0) + oh[i]);
    if (what & 2)
        for (long i = 0; i < ns; ++i) ((T*)var_out)[i] = (T)(p->variance - oh[nsp + i] + oh[2 * nsp + i]);
    return 0;
}

// ------------------------------------------------------------------------------------------------
// Gradient of the sparse objective of a fitted handle (gp_vfe_grad): elbo (VFE, src/sparse_approximations.jl:248-254) or approx_log_evidence (DTC, :282-286)
// against the kernel variance, the Scale / ARD parameters, the noise variances, y, the pseudo-inputs z and (optionally) the inputs x — what an AD backend
// computes when examples/0-intro-1d/script.jl:385-394 maximises the ELBO over kernel parameters and pseudo-points.
//   With ψ = K_zf Σy⁻¹ K_fz, φ = K_zf Σy⁻¹ δ, Q = K_zz + ψ = L_z A L_zᵀ (A = I + B Bᵀ = Λ_ε) and ν = Q⁻¹φ = α (the posterior's own α):
//     L = −½ [N log 2π + log|Q| − log|K_zz| + log|Σy| + δᵀΣy⁻¹δ − φᵀQ⁻¹φ] − ½ [tr(Σy⁻¹K_ff) − tr(K_zz⁻¹ψ)]         (second bracket: VFE only)
//     G_ψ  = ∂L/∂ψ    = ½ [W H_ψ Wᵀ − ααᵀ],   H_ψ  = I − A⁻¹ (VFE) | −A⁻¹ (DTC),            W = L_z⁻ᵀ
//     G_zz = ∂L/∂K_zz = ½ [W H_zz Wᵀ − ααᵀ],  H_zz = I − A⁻¹ − B Bᵀ (VFE) | I − A⁻¹ (DTC)
//     ∂L/∂K_fz = Σy⁻¹ (2 K_fz G_ψ + δ αᵀ),   ∂L/∂k_ii = −½/σ_i² (VFE)
//     ∂L/∂σ_i² = −½/σ_i² + ½ δ_i²/σ_i⁴ [+ ½ k_ii/σ_i⁴ VFE] − (k_iᵀ G_ψ k_i)/σ_i⁴ − (αᵀk_i) δ_i/σ_i⁴,     ∂L/∂y_i = −(δ_i − αᵀk_i)/σ_i²
//   The M×M side is fp64 (two triangular inverses, A⁻¹, two conjugations: ≈ 10 M³ flops); the N-long side streams the retained observations once more in the
//   fit's chunks: kmat (S K_c) → ONE MFMA GEMM T̃ = (S K_c) G_ψ (2·CH·M² flops) → vgrad_kernel (κ, dκ recomputed; every reduction fp64).  The pass is fp64 for
//   fp32 handles too (their retained inputs are widened chunk by chunk): G_ψ carries K_zz⁻¹ (entries up to 1/jitter) against which S K_c cancels to O(1) —
//   rounded to fp32 it cost 6–8 % of the variance / pseudo-input gradients at N = 3 000 already (measured, round 6), at the fp32 MFMA rate that is not a trade.
//   (oracle: gp_oracle.elbo_grad — dense N×N calculus on the textbook form; tests/test_gpu_vfe_grad.py)
// ------------------------------------------------------------------------------------------------
template <typename T, bool XG>
static int32_t launch_vgrad(hipStream_t s, dim3 grid, const T* Cm, long ldc, int explicit_w, const T* xr, long ldxr, const T* xc, long ldxc, int d, int kind,
                            double variance, int nscale, const double* scale, const T* rs, const T* bv, const double* nu, long nr, long nc, double* g,
                            double* gz, long ldgz, double zfac, double* rowq, double* rowp, double* gx, long ldgx) {
#define GPMI_VGRAD_FAST(ND_)                                                                                                                                  \
    hipLaunchKernelGGL((vgrad_fast_kernel<T, ND_, XG>), grid, dim3(256), 0, s, Cm, ldc, explicit_w, xr, ldxr, xc, ldxc, d, kind, (T)variance, nscale, scale, \
                       rs, bv, nu, nr, nc, g, gz, ldgz, zfac, rowq, rowp, gx, ldgx)
    if (d <= 16 && (ldc & 1) == 0) {   // the scratch-free form (16-byte loads of the weights: even leading dimension)
        if (d <= 4) GPMI_VGRAD_FAST(4);
        else if (d <= 8) GPMI_VGRAD_FAST(8);
        else GPMI_VGRAD_FAST(16);
        HIPCHK(hipGetLastError());
        return 0;
    }
#undef GPMI_VGRAD_FAST
    for (int p0 = 0; p0 < d; p0 += 16) {
        hipLaunchKernelGGL((vgrad_kernel<T, 16, true, XG>), grid, dim3(256), 0, s, Cm, ldc, explicit_w, xr, ldxr, xc, ldxc, d, kind, (T)variance, nscale, scale,
                           rs, bv, nu, nr, nc, g, gz, ldgz, zfac, rowq, rowp, gx, ldgx, p0);
        HIPCHK(hipGetLastError());
    }
    return 0;
}

template <typename T>
static int32_t vfe_grad_impl(gp_vfe* p, double* dvar, double* dscale, double* dnoise_sum, void* dnoise, void* dy, double* dz, int z_layout, void* dx,
                             int x_layout) {
    gp_ctx* c = p->ctx;
    const long m = p->m, mp = p->mp, ld = p->ld, CH = p->chunk;
    const int d = p->d;
    const bool vfe = p->approx == 0;
    constexpr bool is_f64 = sizeof(T) == 8;
    for (auto& sg : p->segs)
        if (sg->npad % CH) return set_arg_err(1, "gp_vfe was built with a different vfe_chunk");
    hipStream_t s = c->sm;
    c->ev_used = 0;
    c->gemm_recs.clear();
    for (auto& e : c->ev_phase)
        if (!e) HIPCHK(hipEventCreate(&e));
    const int nsc = std::max(p->nscale, 1);
    const size_t L_b = sizeof(double) * (size_t)(mp + 128 + 128) * ld;
    const size_t X_b = sizeof(double) * (size_t)(CH + 128) * ld, g_b = sizeof(double) * (size_t)(2 + nsc + 2), gz_b = sizeof(double) * (size_t)d * mp;
    void *W_v = 0, *V_v = 0, *H_v = 0, *R_v = 0, *E_v = 0, *Gp_v = 0, *Gz_v = 0, *X_v = 0, *X2_v = 0, *C2_v = 0, *xc_v = 0, *rc_v = 0, *bc_v = 0, *C_v = 0, *g_v = 0, *gz_v = 0, *sc_v = 0;
    DevBufs bufs(c);
    RC(bufs.get(L_b, &W_v));
    RC(bufs.get(L_b, &V_v));
    RC(bufs.get(L_b, &H_v));
    RC(bufs.get(L_b, &R_v));
    RC(bufs.get(L_b, &E_v));
    RC(bufs.get(L_b, &Gp_v));
    RC(bufs.get(L_b, &Gz_v));
    RC(bufs.get(X_b, &X_v));
    if (!is_f64) {  // the chunk's inputs, Σy^-1/2 and b_y widened to fp64
        RC(bufs.get(sizeof(double) * (size_t)d * CH * 2, &xc_v));
        RC(bufs.get(sizeof(double) * (size_t)CH * 2, &rc_v));
        RC(bufs.get(sizeof(double) * (size_t)CH * 2, &bc_v));
    }
    RC(bufs.get(X_b, &C_v));
    RC(bufs.get(X_b, &X2_v));  // double buffers in either mode ("vfe_overlap" = 0 issues the same sequence to one stream)
    RC(bufs.get(X_b, &C2_v));
    RC(bufs.get(g_b, &g_v));
    RC(bufs.get(gz_b, &gz_v));
    RC(bufs.get(sizeof(double) * (size_t)nsc, &sc_v));
    double *W = (double*)W_v, *V = (double*)V_v, *H = (double*)H_v, *R = (double*)R_v, *E = (double*)E_v, *Gp = (double*)Gp_v, *Gz = (double*)Gz_v;
    const double* Lz = (const double*)p->Lz;
    const double* Ld = (const double*)p->Ld;
    const double* alpha = (const double*)p->vec + 2 * mp;
    std::vector<double> sc_h((size_t)nsc, 1.0);
    for (int q = 0; q < p->nscale; ++q) sc_h[q] = p->scale[q];
    std::vector<double> g_h((size_t)(2 + nsc + 2), 0.0), gz_h((size_t)d * mp, 0.0);   // [0] ∂/∂variance, [2 + p] ∂/∂scale_p, [2 + nsc] Σ ∂/∂σ_i², [3 + nsc] Σ σ_i⁻²
    // per observation (concatenated over the segments, arrival order): what the host needs to finish ∂/∂σ_i², ∂/∂y_i, ∂/∂x_i
    long n_all = 0;
    for (auto& sg : p->segs) n_all += sg->n;
    std::vector<double> gx_all(dx ? (size_t)n_all * d : 0);
    const dim3 sq((unsigned)((mp + 255) / 256), (unsigned)mp);
    auto conj = [&](const double* Hm, double* G) -> int32_t {  // G (full, symmetric) = ½ W Hm Wᵀ − ½ ααᵀ
        GridMap gr = plain_map(0, 0, 0);
        gr.beta0 = 1;
        gr.ktri = 2;                                                                           // W upper: k starts at the row tile
        RC(launch_gemm<double>(c, s, R, ld, W, ld, Hm, ld, mp, mp, mp, gr));                   // R = −W Hm        (Hm symmetric)
        GridMap ge = plain_map(1, 0, 0);
        ge.beta0 = 1;
        RC(launch_gemm<double>(c, s, E, ld, R, ld, W, ld, mp, mp, mp, ge));                    // E = −R Wᵀ = W Hm Wᵀ (lower)
        hipLaunchKernelGGL(sym_combine_kernel, sq, dim3(256), 0, s, G, ld, mp, (const double*)E, ld, 0.5, (const double*)nullptr, 0L, 0.0, 0.0, alpha, -0.5);
        HIPCHK(hipGetLastError());
        return 0;
    };
    int32_t rc = [&]() -> int32_t {
        HIPCHK(hipEventRecord(c->ev_phase[0], s));
        HIPCHK(hipMemcpyAsync(sc_v, sc_h.data(), sizeof(double) * (size_t)nsc, hipMemcpyHostToDevice, s));
        HIPCHK(hipMemsetAsync(g_v, 0, g_b, s));
        HIPCHK(hipMemsetAsync(gz_v, 0, gz_b, s));
        for (void* q : {H_v, R_v, E_v, Gp_v, Gz_v}) HIPCHK(hipMemsetAsync(q, 0, L_b, s));   // the GEMMs over-read whole operand tiles: defined slack rows
        HIPCHK(hipMemsetAsync(C_v, 0, X_b, s));
        if (C2_v) HIPCHK(hipMemsetAsync(C2_v, 0, X_b, s));
        HIPCHK(hipMemsetAsync(X_v, 0, X_b, s));
        if (X2_v) HIPCHK(hipMemsetAsync(X2_v, 0, X_b, s));
        // ---- M×M side, fp64
        RC(vfe_upper_inverse(c, s, Lz, ld, mp, W, L_b, bufs));                                  // W = L_z⁻ᵀ
        RC(vfe_upper_inverse(c, s, Ld, ld, mp, V, L_b, bufs));                                  // V = Λ_ε.L⁻ᵀ
        {
            GridMap gw = plain_map(1, 0, 0);
            gw.ktri = 2;
            gw.beta0 = 1;
            RC(launch_gemm<double>(c, s, E, ld, V, ld, V, ld, mp, mp, mp, gw));                 // E = −V Vᵀ = −A⁻¹ (lower)
        }
        // H_ψ, then H_zz (padding rows: A⁻¹ = I and B Bᵀ = 0 there, so the VFE forms vanish on the padding; the DTC H_ψ keeps −1 on the padded diagonal,
        // which only reaches padded columns of T̃ and padded entries of G — never read)
        hipLaunchKernelGGL(sym_combine_kernel, sq, dim3(256), 0, s, H, ld, mp, (const double*)E, ld, 1.0, (const double*)nullptr, 0L, 0.0, vfe ? 1.0 : 0.0,
                           (const double*)nullptr, 0.0);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(V, E, L_b, hipMemcpyDeviceToDevice, s));                          // −A⁻¹ (lower) survives the first conjugation in V's storage
        RC(conj(H, Gp));
        hipLaunchKernelGGL(sym_combine_kernel, sq, dim3(256), 0, s, H, ld, mp, (const double*)V, ld, 1.0, vfe ? (const double*)p->Dacc : (const double*)nullptr, ld,
                           1.0, 1.0, (const double*)nullptr, 0.0);                              // Dacc = −B Bᵀ (lower)
        HIPCHK(hipGetLastError());
        RC(conj(H, Gz));
        HIPCHK(hipEventRecord(c->ev_phase[1], s));
        // ---- the observations, chunk by chunk: the MFMA GEMMs back to back on one stream; the next chunk's kmat (and, for fp32 handles, the widening of its
        //      inputs) and the previous chunk's vgrad on another beside them, on double buffers ("vfe_overlap")
        struct Ck {
            const ObsSeg* sg;
            long c0;
            double *rq, *rp, *gx;
        };
        std::vector<Ck> cks;
        std::vector<std::array<void*, 3>> segb;
        for (auto& sgp : p->segs) {
            const ObsSeg& sg = *sgp;
            void *rq_v = 0, *rp_v = 0, *gx_v = 0;
            const size_t r_b = sizeof(double) * (size_t)std::max(sg.npad, 1L);
            RC(bufs.get(r_b, &rq_v));
            RC(bufs.get(r_b, &rp_v));
            if (dx) RC(bufs.get(r_b * d, &gx_v));
            HIPCHK(hipMemsetAsync(rq_v, 0, r_b, s));
            HIPCHK(hipMemsetAsync(rp_v, 0, r_b, s));
            if (dx) HIPCHK(hipMemsetAsync(gx_v, 0, r_b * d, s));
            segb.push_back({rq_v, rp_v, gx_v});
            for (long c0 = 0; c0 < sg.npad && c0 < sg.n; c0 += CH) cks.push_back({&sg, c0, (double*)rq_v + c0, (double*)rp_v + c0, dx ? (double*)gx_v + c0 : nullptr});
        }
        // One stream.  The two-stream forms were measured at C5 (fp64 handle, tools/c5_grad_probe.py): helpers on the ctx's high-priority stream beside the GEMMs
        // 146.8 ms against 142.5 serial; GEMMs on the high-priority stream, helpers on the main one 142.6 / 143.3; and after the helper kernel lost its scratch
        // traffic (vgrad_fast_kernel: 1.1 -> 0.25 ms per chunk) 132.4 with two streams against 129.1 with one — what is left beside the GEMM (kmat + vgrad ≈ 0.4 ms
        // of 8.1 per chunk) costs more as a co-runner than in line.  The double buffers and events below stay for the experiment switch GPMI_VGRAD_STREAMS=2.
        static const bool two_streams = [] { const char* e = getenv("GPMI_VGRAD_STREAMS"); return e && e[0] == '2'; }();
        const bool ovl = two_streams && c->vfe_overlap != 0;
        hipStream_t sa = s, sg = ovl ? c->sp : s;
        void* Xb[2] = {X_v, X2_v};
        void* Cb[2] = {C_v, C2_v};
        hipEvent_t evK[2] = {nullptr, nullptr}, evG[2] = {nullptr, nullptr}, evV[2] = {nullptr, nullptr};
        struct In {
            const double *xr, *rs, *b;
            long ldxr, nr;
        } in[2];
        if (ovl) {  // the GEMM stream starts after the M×M side (G_ψ) and the zeroing above
            hipEvent_t e;
            RC(ctx_event(c, &e, false));
            HIPCHK(hipEventRecord(e, s));
            HIPCHK(hipStreamWaitEvent(sg, e, 0));
        }
        auto prep = [&](size_t ci) -> int32_t {  // S K(x_c, z) of chunk ci into X[ci & 1], on the helper stream
            const int bb = (int)(ci & 1);
            const ObsSeg& sg = *cks[ci].sg;
            const long c0 = cks[ci].c0;
            if (ovl && evG[bb]) HIPCHK(hipStreamWaitEvent(sa, evG[bb], 0));  // the GEMM of chunk ci − 2 has read X[bb]
            In& I = in[bb];
            I.nr = std::min(CH, sg.n - c0);
            if constexpr (is_f64) {
                I.xr = (const double*)sg.xs + c0; I.ldxr = sg.npad; I.rs = (const double*)sg.rs + c0; I.b = (const double*)sg.b + c0;
            } else {
                double* xc = (double*)xc_v + (size_t)bb * d * CH;
                double* rc2 = (double*)rc_v + (size_t)bb * CH;
                double* bc2 = (double*)bc_v + (size_t)bb * CH;
                for (int q = 0; q < d; ++q) {
                    hipLaunchKernelGGL((convert_kernel<T, double>), convert_grid(CH), dim3(256), 0, sa, (const T*)sg.xs + (long)q * sg.npad + c0, xc + (long)q * CH, CH, 1.0);
                    HIPCHK(hipGetLastError());
                }
                hipLaunchKernelGGL((convert_kernel<T, double>), convert_grid(CH), dim3(256), 0, sa, (const T*)sg.rs + c0, rc2, CH, 1.0);
                HIPCHK(hipGetLastError());
                hipLaunchKernelGGL((convert_kernel<T, double>), convert_grid(CH), dim3(256), 0, sa, (const T*)sg.b + c0, bc2, CH, 1.0);
                HIPCHK(hipGetLastError());
                I.xr = xc; I.ldxr = CH; I.rs = rc2; I.b = bc2;
            }
            GridMap g = plain_map(0, 0, 0);
            dim3 grid((unsigned)(mp / 128), (unsigned)(CH / 128));
            launch_kmat<double>(grid, sa, (double*)Xb[bb], ld, I.xr, I.ldxr, (const double*)p->zs, mp, d, p->kind, p->variance, (const double*)nullptr, I.nr, m, 0, g,
                                (const double*)nullptr, I.rs);
            HIPCHK(hipGetLastError());
            if (ovl) {
                RC(ctx_event(c, &evK[bb], false));
                HIPCHK(hipEventRecord(evK[bb], sa));
            }
            return 0;
        };
        if (!cks.empty()) RC(prep(0));
        for (size_t ci = 0; ci < cks.size(); ++ci) {
            const int bb = (int)(ci & 1);
            if (ovl) {
                HIPCHK(hipStreamWaitEvent(sg, evK[bb], 0));
                if (evV[bb]) HIPCHK(hipStreamWaitEvent(sg, evV[bb], 0));  // vgrad of chunk ci − 2 has read C[bb]
            }
            {
                GridMap g = plain_map(0, 0, 0);
                g.beta0 = 1;
                RC(launch_gemm<double>(c, sg, (double*)Cb[bb], ld, (const double*)Xb[bb], ld, (const double*)Gp, ld, CH, mp, mp, g));   // C = −(S K_c) G_ψ
            }
            if (ovl) {
                RC(ctx_event(c, &evG[bb], false));
                HIPCHK(hipEventRecord(evG[bb], sg));
            }
            const In I = in[bb];
            if (ci + 1 < cks.size()) RC(prep(ci + 1));  // beside this chunk's GEMM
            if (ovl) HIPCHK(hipStreamWaitEvent(sa, evG[bb], 0));
            dim3 grid((unsigned)(mp / 128), (unsigned)((I.nr + 127) / 128));
            if (dx)
                RC((launch_vgrad<double, true>(sa, grid, (const double*)Cb[bb], ld, 0, I.xr, I.ldxr, (const double*)p->zs, mp, d, p->kind, p->variance, p->nscale,
                                               (const double*)sc_v, I.rs, I.b, alpha, I.nr, m, (double*)g_v, (double*)gz_v, mp, 1.0, cks[ci].rq, cks[ci].rp,
                                               cks[ci].gx, cks[ci].sg->npad)));
            else
                RC((launch_vgrad<double, false>(sa, grid, (const double*)Cb[bb], ld, 0, I.xr, I.ldxr, (const double*)p->zs, mp, d, p->kind, p->variance, p->nscale,
                                                (const double*)sc_v, I.rs, I.b, alpha, I.nr, m, (double*)g_v, (double*)gz_v, mp, 1.0, cks[ci].rq, cks[ci].rp,
                                                (double*)nullptr, 0L)));
            if (ovl) {
                RC(ctx_event(c, &evV[bb], false));
                HIPCHK(hipEventRecord(evV[bb], sa));
            }
        }
        // (every GEMM is followed by its vgrad on the main stream, which waited for it: nothing to join)
        // ---- per observation: ∂/∂σ_i², ∂/∂y_i and the two N-long sums on the device; only what the caller asked for comes back
        long off = 0;
        for (size_t si = 0; si < p->segs.size(); ++si) {
            const ObsSeg& sg = *p->segs[si];
            if (sg.n == 0) continue;
            void *dn_v = 0, *dy_v = 0;
            RC(bufs.get(sizeof(T) * (size_t)sg.n, &dn_v));
            RC(bufs.get(sizeof(T) * (size_t)sg.n, &dy_v));
            hipLaunchKernelGGL(vgrad_finish_kernel<T>, dim3((unsigned)std::min<long>((sg.n + 255) / 256, 2048)), dim3(256), 0, s, (const T*)sg.rs, (const T*)sg.b,
                               (const double*)segb[si][0], (const double*)segb[si][1], sg.n, p->variance, vfe ? 1 : 0, (T*)dn_v, (T*)dy_v, (double*)g_v + 2 + nsc);
            HIPCHK(hipGetLastError());
            if (dnoise) HIPCHK(hipMemcpyAsync((T*)dnoise + off, dn_v, sizeof(T) * (size_t)sg.n, hipMemcpyDeviceToHost, s));
            if (dy) HIPCHK(hipMemcpyAsync((T*)dy + off, dy_v, sizeof(T) * (size_t)sg.n, hipMemcpyDeviceToHost, s));
            if (dx) {
                std::vector<double> gx_h((size_t)sg.npad * d);
                HIPCHK(hipMemcpyAsync(gx_h.data(), segb[si][2], sizeof(double) * (size_t)sg.npad * d, hipMemcpyDeviceToHost, s));
                HIPCHK(hipStreamSynchronize(s));
                for (int q = 0; q < d; ++q)
                    for (long i = 0; i < sg.n; ++i) gx_all[(size_t)q * n_all + off + i] = gx_h[(size_t)q * sg.npad + i];
            }
            off += sg.n;
        }
        HIPCHK(hipEventRecord(c->ev_phase[2], s));
        // ---- K_zz: explicit weights G_zz over the full square; z_j enters through both arguments (factor 2 on the column role)
        {
            dim3 grid((unsigned)(mp / 128), (unsigned)((m + 127) / 128));
            RC((launch_vgrad<double, false>(s, grid, (const double*)Gz, ld, 1, (const double*)p->zs, mp, (const double*)p->zs, mp, d, p->kind, p->variance, p->nscale,
                                            (const double*)sc_v, (const double*)nullptr, (const double*)nullptr, (const double*)nullptr, m, m, (double*)g_v,
                                            (double*)gz_v, mp, 2.0, (double*)nullptr, (double*)nullptr, (double*)nullptr, 0L)));
        }
        HIPCHK(hipMemcpyAsync(g_h.data(), g_v, g_b, hipMemcpyDeviceToHost, s));
        HIPCHK(hipMemcpyAsync(gz_h.data(), gz_v, gz_b, hipMemcpyDeviceToHost, s));
        HIPCHK(hipEventRecord(c->ev_phase[3], s));
        HIPCHK(hipStreamSynchronize(s));
        float ms;
        HIPCHK(hipEventElapsedTime(&ms, c->ev_phase[0], c->ev_phase[1]));
        c->tm.assemble_ms = ms;  // the M×M side
        HIPCHK(hipEventElapsedTime(&ms, c->ev_phase[1], c->ev_phase[2]));
        c->tm.potrf_ms = ms;     // the streamed pass
        HIPCHK(hipEventElapsedTime(&ms, c->ev_phase[2], c->ev_phase[3]));
        c->tm.solve_ms = ms;     // the K_zz term
        HIPCHK(hipEventElapsedTime(&ms, c->ev_phase[0], c->ev_phase[3]));
        c->tm.total_ms = ms;
        return 0;
    }();
    if (rc != 0) {
        (void)hipStreamSynchronize(c->sm);
        return rc;
    }
    if (dvar) *dvar = g_h[0] - (vfe ? 0.5 * g_h[3 + nsc] : 0.0);   // − ½ Σ_i ∂k_ii/∂σ_k² / σ_i²: the trace term
    if (dscale)
        for (int q = 0; q < p->nscale; ++q) dscale[q] = g_h[2 + q];
    if (dnoise_sum) *dnoise_sum = g_h[2 + nsc];
    if (dz)  // the container layout of the pseudo-inputs (src/finite_gp_projection.jl:32-37): 0 vector, 1 ColVecs (D×M column-major), 2 RowVecs (M×D column-major)
        for (int q = 0; q < d; ++q)
            for (long j = 0; j < m; ++j) {
                const double v = gz_h[(size_t)q * mp + j];
                if (z_layout == 0) dz[j] = v;
                else if (z_layout == 1) dz[(long)q + j * d] = v;
                else dz[j + (long)q * m] = v;
            }
    if (dx)
        for (int q = 0; q < d; ++q)
            for (long i = 0; i < n_all; ++i) {
                const T v = (T)gx_all[(size_t)q * n_all + i];
                if (x_layout == 0) ((T*)dx)[i] = v;
                else if (x_layout == 1) ((T*)dx)[(long)q + i * d] = v;
                else ((T*)dx)[i + (long)q * n_all] = v;
            }
    return 0;
}
