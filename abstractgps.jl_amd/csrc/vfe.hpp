// vfe.hpp — VFE / DTC sparse approximation on the device (included by gpmi355.hip).
//
// Reference: posterior(::Union{VFE,DTC}, fx, y)      src/sparse_approximations.jl:58-75
//            approx_log_evidence / elbo              :248-254 (VFE), :282-286 (DTC), :289-305
//            predictive mean / var                   :183-185, :192-195, :212-217
//
// Formulation (S = Σy^-1/2 diagonal, L_z L_zᵀ = K_zz + jitter·I, so U = L_zᵀ), the reference's own:
//   B = U⁻ᵀ (S K_xz)ᵀ,  D = B Bᵀ + I = Λ_ε,  c = B b_y,  m_ε = Λ_ε \ c,  α = U \ m_ε.
//   The N-long pass is streamed in chunks of CH data points and never holds an N×M matrix.  L_z is the same for
//   every chunk, so its inverse is formed once (fp64: I·L_z⁻ᵀ by the blocked TRSM, transposed, rounded to T) and the
//   per-chunk triangular solve becomes ONE MFMA GEMM with a triangular k range (no 64-wide leaf chain per chunk):
//     Xc  = S_c K(x_c, z)                   kmat with a row scale          (CH × M, dtype T)
//     Y   = −inv(L_z) Xcᵀ = −B_c            gemm_nt_dma, beta0 + ktri      (M × CH, data points contiguous)
//     D_acc −= Y Yᵀ                         MFMA gemm (NT), fp64 accumulation (SYRK over the data points)
//     c_acc −= Y b_c                        rowdot_sub
//   (T = f32 or f64).  The M×M side (K_zz, both Choleskys, all vector solves) is always fp64, and so are the
//   accumulators of the N-long reductions (D_acc, c_acc, ‖B‖²_F): in fp32 mode the operands stream in fp32 through
//   the fp32 MFMA, whose chain is flushed into fp64 every 256 data points; L_z is rounded to fp32 for the streamed
//   TRSM only.  ‖A‖²_F = tr(B Bᵀ) = −tr(D_acc) (ELBO trace term, :251).
#pragma once

struct gp_vfe {
    gp_ctx* ctx;
    int dtype;
    long m, mp, ld;
    int d, kind;
    double variance;
    int nscale;
    std::vector<double> scale;
    void *Lz, *Ld;  // mp × ld doubles (+128 slack rows)
    size_t L_bytes;
    void* zs;  // scaled inducing inputs, double [d][mp]
    size_t zs_bytes;
    void *alpha, *meps;  // double [mp]
    size_t vec_bytes;
    // streaming state kept for update_posterior (new observations, src/sparse_approximations.jl:87-121): the accumulators
    // of the N-long reductions before the M×M finalisation, and what the chunk loop needs (inv(L_z), scaled z in T)
    int approx;
    long n_obs;
    void *Dacc, *cacc, *Li, *zsT;
    size_t D_bytes, c_bytes, Li_bytes, zsT_bytes;
    double logdet_sy, dd, tr_kff, trZ;
};

template <typename T>
static int32_t vfe_fit_impl(gp_ctx* c, const gp_kernel* k, const gp_points* x, const gp_points* z, const gp_noise* noise,
                            double jitter, const void* mean_or_null, const void* yv, int approx, gp_vfe* out,
                            double* objective, const gp_vfe* prev = nullptr) {
    // prev != NULL: continue the streamed reductions of an existing fit with the new observations (x, y) — z, k, jitter
    // are then taken from prev (z == NULL)
    const long n = x->n, m = prev ? prev->m : z->n, mp = round_up(m, 128);
    const int d = x->d;
    const long CH = 8192;                       // columns (data points) per streamed chunk
    const long npad = round_up(n, CH);
    const long ld = mp + c->ldpad;              // M×M matrices and the CH×M chunk
    const T* y = (const T*)yv;
    const T* mean = (const T*)mean_or_null;

    c->ev_used = 0;
    c->gemm_recs.clear();
    for (auto& e : c->ev_phase)
        if (!e) HIPCHK(hipEventCreate(&e));
    if (!c->info_dev) HIPCHK(hipMalloc((void**)&c->info_dev, sizeof(int)));
    RC(ctx_scal(c, 16 + 128));

    // ---- host marshalling
    std::vector<T> xs_h, zsT_h;
    scale_points<T>(k, x, npad, xs_h);
    if (!prev) scale_points<T>(k, z, mp, zsT_h);
    else zsT_h.assign((size_t)d * mp, T(0));
    std::vector<double> zsD_h(zsT_h.begin(), zsT_h.end());
    std::vector<T> rs_h((size_t)npad, T(0)), b_h((size_t)npad, T(0));  // s_i = σ_i⁻¹ ; b_i = s_i δ_i  (b_y, :66)
    double logdet_sy = prev ? prev->logdet_sy : 0, dd = prev ? prev->dd : 0, tr_kff = prev ? prev->tr_kff : 0;
    for (long i = 0; i < n; ++i) {
        const double s2 = noise->kind == 0 ? noise->s : (double)((const T*)noise->diag)[i];
        if (!(s2 > 0)) return 1 + (int32_t)i;  // chol(Σy) fails at i (reference :61 / :296)
        const double delta = (double)(T)(y[i] - (mean ? mean[i] : T(0)));
        const double si = 1.0 / std::sqrt(s2);
        rs_h[i] = (T)si;
        b_h[i] = (T)(delta * si);
        logdet_sy += std::log(s2);
        dd += (double)b_h[i] * (double)b_h[i];
        tr_kff += k->variance / s2;  // tr_Cf_invΣy :307-313
    }
    std::vector<double> jit_h((size_t)mp, 0.0);
    for (long i = 0; i < m; ++i) jit_h[i] = jitter;

    const size_t xs_b = sizeof(T) * xs_h.size(), zsT_b = sizeof(T) * zsT_h.size(), zsD_b = sizeof(double) * zsD_h.size();
    const size_t rs_b = sizeof(T) * (size_t)npad, D_b = sizeof(double) * (size_t)(mp + 128) * ld;
    const long ldy = CH + c->ldpad;             // Y = −B_c (M × CH)
    const size_t X_b = sizeof(T) * (size_t)(CH + 128) * ld, L_b = sizeof(double) * (size_t)(mp + 128 + 128) * ld;
    const size_t Y_b = sizeof(T) * (size_t)(mp + 128) * ldy, Li_b = sizeof(T) * (size_t)(mp + 128) * ld;
    const size_t vT_b = sizeof(double) * (size_t)mp, vD_b = sizeof(double) * (size_t)mp * 4, jit_b = sizeof(double) * (size_t)mp;
    void *xs_v = 0, *zsT_v = 0, *zsD_v = 0, *rs_v = 0, *b_v = 0, *D_v = 0, *X_v = 0, *Lz_v = 0, *Ld_v = 0, *LzT_v = 0, *cT_v = 0,
         *vec_v = 0, *jit_v = 0, *Y_v = 0, *Li_v = 0, *I_v = 0, *S_v = 0;
    constexpr bool is_f64 = sizeof(T) == 8;
    RC(ctx_alloc(c, xs_b, &xs_v));
    RC(ctx_alloc(c, zsT_b, &zsT_v));
    RC(ctx_alloc(c, zsD_b, &zsD_v));
    RC(ctx_alloc(c, rs_b, &rs_v));
    RC(ctx_alloc(c, rs_b, &b_v));
    RC(ctx_alloc(c, D_b, &D_v));
    RC(ctx_alloc(c, X_b, &X_v));
    RC(ctx_alloc(c, L_b, &Lz_v));
    RC(ctx_alloc(c, L_b, &Ld_v));
    RC(ctx_alloc(c, Y_b, &Y_v));
    RC(ctx_alloc(c, Li_b, &Li_v));
    RC(ctx_alloc(c, L_b, &I_v));
    if (!is_f64) RC(ctx_alloc(c, 4 * Li_b, &S_v));
    RC(ctx_alloc(c, vT_b, &cT_v));
    RC(ctx_alloc(c, vD_b, &vec_v));
    RC(ctx_alloc(c, jit_b, &jit_v));
    double* Lz = (double*)Lz_v;
    double* Ld = (double*)Ld_v;
    double* vec = (double*)vec_v;  // rows: [0] c , [1] w→m_ε , [2] α , [3] spare
    double scal_h[16] = {0};
    int info_h = 0;
    hipStream_t s = c->sm;
    const double trZ_prev = prev ? prev->trZ : 0.0;
    if (prev && prev->D_bytes != D_b) return set_arg_err(1, "stale gp_vfe state");
    const long n_all = (prev ? prev->n_obs : 0) + n;

    int32_t rc = [&]() -> int32_t {
        HIPCHK(hipEventRecord(c->ev_phase[0], s));
        HIPCHK(hipMemcpyAsync(xs_v, xs_h.data(), xs_b, hipMemcpyHostToDevice, s));
        HIPCHK(hipMemcpyAsync(zsT_v, zsT_h.data(), zsT_b, hipMemcpyHostToDevice, s));
        HIPCHK(hipMemcpyAsync(zsD_v, zsD_h.data(), zsD_b, hipMemcpyHostToDevice, s));
        HIPCHK(hipMemcpyAsync(rs_v, rs_h.data(), rs_b, hipMemcpyHostToDevice, s));
        HIPCHK(hipMemcpyAsync(b_v, b_h.data(), rs_b, hipMemcpyHostToDevice, s));
        HIPCHK(hipMemcpyAsync(jit_v, jit_h.data(), jit_b, hipMemcpyHostToDevice, s));
        HIPCHK(hipMemsetAsync(c->info_dev, 0, sizeof(int), s));
        HIPCHK(hipMemsetAsync(c->scal_dev, 0, sizeof(double) * 16, s));
        HIPCHK(hipMemsetAsync(D_v, 0, D_b, s));
        HIPCHK(hipMemsetAsync(cT_v, 0, vT_b, s));
        HIPCHK(hipMemsetAsync(vec_v, 0, vD_b, s));
        if (prev) {  // resume: factor of K_zz, its inverse, the scaled inducing inputs and the running sums come from prev
            HIPCHK(hipMemcpyAsync(Lz_v, prev->Lz, L_b, hipMemcpyDeviceToDevice, s));
            HIPCHK(hipMemcpyAsync(Li_v, prev->Li, Li_b, hipMemcpyDeviceToDevice, s));
            HIPCHK(hipMemcpyAsync(zsT_v, prev->zsT, zsT_b, hipMemcpyDeviceToDevice, s));
            HIPCHK(hipMemcpyAsync(zsD_v, prev->zs, zsD_b, hipMemcpyDeviceToDevice, s));
            HIPCHK(hipMemcpyAsync(D_v, prev->Dacc, D_b, hipMemcpyDeviceToDevice, s));
            HIPCHK(hipMemcpyAsync(cT_v, prev->cacc, vT_b, hipMemcpyDeviceToDevice, s));
            HIPCHK(hipMemcpyAsync(c->scal_dev + 4, &trZ_prev, sizeof(double), hipMemcpyHostToDevice, s));
        } else {
        // ---- L_z = chol(K_zz + jitter I), fp64                                                    :62
        {
            GridMap g = plain_map(1, 0, 0);
            dim3 grid((unsigned)(mp / 128), (unsigned)(mp / 128));
            hipLaunchKernelGGL(kmat_kernel<double>, grid, dim3(256), 0, s, Lz, ld, (const double*)zsD_v, mp,
                               (const double*)zsD_v, mp, d, k->kind, k->variance, (const double*)jit_v, m, m, 1, g,
                               (const double*)nullptr, (const double*)nullptr);
            HIPCHK(hipGetLastError());
        }
        RC(potrf_full<double>(c, Lz, ld, mp, mp, c->info_dev, m, c->scal_dev + 0));
        // ---- inv(L_z): W = I · L_z⁻ᵀ (upper), transposed into Ld's storage (free until the SYRK is done), rounded to T
        {
            double* Iw = (double*)I_v;
            HIPCHK(hipMemsetAsync(I_v, 0, L_b, s));
            hipLaunchKernelGGL(identity_kernel<double>, dim3((unsigned)((mp + 255) / 256), (unsigned)mp), dim3(256), 0, s, Iw, ld, mp);
            HIPCHK(hipGetLastError());
            RC(trsm_upper_rec<double>(c, s, Iw, ld, Lz, ld, 0, mp));
            hipLaunchKernelGGL(transpose_f64_kernel, dim3((unsigned)(mp / 32), (unsigned)(mp / 32)), dim3(256), 0, s, Iw, ld, Ld, ld, mp);
            HIPCHK(hipGetLastError());
            HIPCHK(hipMemsetAsync(Li_v, 0, Li_b, s));
            hipLaunchKernelGGL((convert_kernel<double, T>), dim3((unsigned)(((long)mp * ld + 255) / 256)), dim3(256), 0, s, Ld,
                               (T*)Li_v, (long)mp * ld, 1.0);
            HIPCHK(hipGetLastError());
        }
        }  // !prev
        HIPCHK(hipEventRecord(c->ev_phase[1], s));
        // ---- streamed pass over the N data points                                                :64-71
        for (long c0 = 0; c0 < npad; c0 += CH) {
            GridMap g = plain_map(0, c0, 0);
            dim3 grid((unsigned)(mp / 128), (unsigned)(CH / 128));
            hipLaunchKernelGGL(kmat_kernel<T>, grid, dim3(256), 0, s, (T*)X_v, ld, (const T*)xs_v, npad, (const T*)zsT_v, mp, d,
                               k->kind, (T)k->variance, (const T*)nullptr, n, m, 0, g, (const T*)nullptr, (const T*)rs_v);
            HIPCHK(hipGetLastError());
            {
                GridMap gy = plain_map(0, 0, 0);
                gy.beta0 = 1;
                gy.ktri = 1;
                RC(launch_gemm<T>(c, s, (T*)Y_v, ldy, (const T*)Li_v, ld, (const T*)X_v, ld, mp, CH, mp, gy));   // Y = −B_c
            }
            hipLaunchKernelGGL(sumsq_accum_kernel<T>, dim3((unsigned)mp), dim3(256), 0, s, (const T*)Y_v, ldy, CH,
                               c->scal_dev + 4);                                                   // ‖A‖²_F in fp64
            HIPCHK(hipGetLastError());
            if constexpr (is_f64) {
                RC((launch_gemm<T, double>(c, s, (double*)D_v, ld, (const T*)Y_v, ldy, (const T*)Y_v, ldy, mp, mp, CH,
                                           plain_map(1, 0, 0))));
            } else {
                // fp32: the chunk's SYRK runs on the LDS-DMA kernel into an fp32 scratch (−Y Yᵀ over this chunk's 8 192 data
                // points only), which is then added into the fp64 accumulator — fp64 sums across chunks, fp32 MFMA within
                // four K = 2 048 partial products in ONE launch (blockIdx.z): 4 × 528 lower tiles fill the 512 workgroup
                // slots four times over instead of 1.03 times; the partials are summed into the fp64 accumulator below
                constexpr long KS = 2048;
                constexpr int NB4 = (int)(8192 / KS);
                GridMap gs = plain_map(1, 0, 0);
                gs.beta0 = 1;
                gs.nbatch = NB4;
                gs.cstride = (long)(mp + 128) * ld;
                RC(launch_gemm<T>(c, s, (T*)S_v, ld, (const T*)Y_v, ldy, (const T*)Y_v, ldy, mp, mp, KS, gs));
                for (int b = 0; b < NB4; ++b) {
                    hipLaunchKernelGGL(add_lower_to_f64_kernel<T>, dim3((unsigned)((mp + 255) / 256), (unsigned)mp), dim3(256), 0,
                                       s, (const T*)S_v + (long)b * gs.cstride, ld, (double*)D_v, ld, mp);
                    HIPCHK(hipGetLastError());
                }
            }
            hipLaunchKernelGGL(rowdot_sub_kernel<T>, dim3((unsigned)mp), dim3(256), 0, s, (const T*)Y_v, ldy, CH,
                               (const T*)b_v + c0, (double*)cT_v);                                 // cT −= Y b_c = +B_c b_c
            HIPCHK(hipGetLastError());
        }
        hipLaunchKernelGGL((convert_kernel<double, double>), dim3((unsigned)((mp + 255) / 256)), dim3(256), 0, s,
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
        HIPCHK(hipEventRecord(c->ev_phase[2], s));
        HIPCHK(hipEventRecord(c->ev_phase[3], s));
        HIPCHK(hipMemcpyAsync(&info_h, c->info_dev, sizeof(int), hipMemcpyDeviceToHost, s));
        HIPCHK(hipMemcpyAsync(scal_h, c->scal_dev, sizeof(double) * 16, hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        float ms;
        HIPCHK(hipEventElapsedTime(&ms, c->ev_phase[0], c->ev_phase[1]));
        c->tm.assemble_ms = ms;  // K_zz + its Cholesky
        HIPCHK(hipEventElapsedTime(&ms, c->ev_phase[1], c->ev_phase[2]));
        c->tm.potrf_ms = ms;     // streamed pass + M×M side
        c->tm.solve_ms = 0;
        HIPCHK(hipEventElapsedTime(&ms, c->ev_phase[0], c->ev_phase[3]));
        c->tm.total_ms = ms;
        c->tm.gemm_ms = 0;
        c->tm.gemm_flops = 0;
        c->tm.gemm_launches = (int64_t)c->gemm_recs.size();
        for (auto& r : c->gemm_recs) {
            HIPCHK(hipEventElapsedTime(&ms, r.a, r.b));
            c->tm.gemm_ms += ms;
            c->tm.gemm_flops += r.flops;
        }
        return 0;
    }();
    if (rc != 0) {
        (void)hipStreamSynchronize(c->sm);
        (void)hipStreamSynchronize(c->sp);
    }
    ctx_release(c, xs_v, xs_b);
    ctx_release(c, rs_v, rs_b);
    ctx_release(c, b_v, rs_b);
    ctx_release(c, X_v, X_b);
    ctx_release(c, Y_v, Y_b);
    ctx_release(c, I_v, L_b);
    ctx_release(c, S_v, 4 * Li_b);
    ctx_release(c, jit_v, jit_b);
    if (rc == 0 && info_h != 0) rc = info_h;
    if (rc != 0 || !out) {
        ctx_release(c, Lz_v, L_b);
        ctx_release(c, Ld_v, L_b);
        ctx_release(c, zsD_v, zsD_b);
        ctx_release(c, vec_v, vD_b);
        ctx_release(c, D_v, D_b);
        ctx_release(c, cT_v, vT_b);
        ctx_release(c, Li_v, Li_b);
        ctx_release(c, zsT_v, zsT_b);
        if (rc != 0) return rc;
    }
    // objective: dtc = -½ (N log2π + logdet Σy + logdet Λ_ε + ‖δ_s‖² − ‖Λ_ε.U⁻ᵀ A δ_s‖²)          :302-303
    //            elbo = dtc − ½ (tr(K_ff Σy⁻¹) − ‖A‖²_F),  ‖A‖²_F = tr(D − I)                      :251
    const double logdet_lam = 2.0 * scal_h[1], trZ = scal_h[4], quad = scal_h[3];  // trZ = ‖B‖²_F = ‖A‖²_F
    double obj = -0.5 * ((double)n_all * LOG2PI + logdet_sy + logdet_lam + dd - quad);
    if (approx == 0) obj -= 0.5 * (tr_kff - trZ);
    if (objective) *objective = obj;
    if (out) {
        out->dtype = k->dtype;
        out->m = m; out->mp = mp; out->ld = ld; out->d = d; out->kind = k->kind;
        out->variance = k->variance; out->nscale = k->nscale;
        out->scale.clear();
        if (k->scale && k->nscale > 0) out->scale.assign(k->scale, k->scale + k->nscale);
        out->Lz = Lz_v; out->Ld = Ld_v; out->L_bytes = L_b;
        out->zs = zsD_v; out->zs_bytes = zsD_b;
        out->alpha = vec + 2 * mp; out->meps = vec + mp; out->vec_bytes = vD_b;
        // keep the base pointer of the vector block for release
        out->alpha = (void*)vec;  // block base; α at +2mp, m_ε at +mp
        out->approx = approx;
        out->n_obs = n_all;
        out->Dacc = D_v; out->D_bytes = D_b;
        out->cacc = cT_v; out->c_bytes = vT_b;
        out->Li = Li_v; out->Li_bytes = Li_b;
        out->zsT = zsT_v; out->zsT_bytes = zsT_b;
        out->logdet_sy = logdet_sy; out->dd = dd; out->tr_kff = tr_kff; out->trZ = trZ;
    }
    return 0;
}

template <typename T>
static int32_t vfe_predict_impl(gp_vfe* p, const gp_points* xs, const void* pm, int what, void* mean_out, void* var_out) {
    gp_ctx* c = p->ctx;
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
    const double* vec = (const double*)p->alpha;
    const size_t xs_b = sizeof(double) * xs_h.size(), X_b = sizeof(double) * (size_t)(nsp + 128) * ld,
                 o_b = sizeof(double) * (size_t)nsp * 3;
    void *xs_v = 0, *X_v = 0, *o_v = 0;
    RC(ctx_alloc(c, xs_b, &xs_v));
    RC(ctx_alloc(c, X_b, &X_v));
    RC(ctx_alloc(c, o_b, &o_v));
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
            hipLaunchKernelGGL(kmat_kernel<double>, grid, dim3(256), 0, s, X, ld, (const double*)xs_v, nsp,
                               (const double*)p->zs, mp, d, p->kind, p->variance, (const double*)nullptr, ns, m, 0, g,
                               (const double*)nullptr, (const double*)nullptr);
            HIPCHK(hipGetLastError());
            RC(trsm_rec<double>(c, s, X, ld, nsp, (const double*)p->Lz, ld, mp));
            hipLaunchKernelGGL(rowsumsq_kernel<double>, dim3((unsigned)nsp), dim3(256), 0, s, X, ld, mp, o + nsp);
            HIPCHK(hipGetLastError());
            RC(trsm_rec<double>(c, s, X, ld, nsp, (const double*)p->Ld, ld, mp));
            hipLaunchKernelGGL(rowsumsq_kernel<double>, dim3((unsigned)nsp), dim3(256), 0, s, X, ld, mp, o + 2 * nsp);
            HIPCHK(hipGetLastError());
        }
        HIPCHK(hipMemcpyAsync(oh.data(), o, o_b, hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        return 0;
    }();
    if (rc != 0) (void)hipStreamSynchronize(s);
    ctx_release(c, xs_v, xs_b);
    ctx_release(c, X_v, X_b);
    ctx_release(c, o_v, o_b);
    if (rc != 0) return rc;
    const T* prior = (const T*)pm;
    if (what & 1)
        for (long i = 0; i < ns; ++i) ((T*)mean_out)[i] = (T)((prior ? (double)prior[i] : 0.0) + oh[i]);
    if (what & 2)
        for (long i = 0; i < ns; ++i) ((T*)var_out)[i] = (T)(p->variance - oh[nsp + i] + oh[2 * nsp + i]);
    return 0;
}
