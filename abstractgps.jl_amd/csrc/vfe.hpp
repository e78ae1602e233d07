// vfe.hpp — VFE / DTC sparse approximation on the device (included by gpmi355.hip).
//
// Reference: posterior(::Union{VFE,DTC}, fx, y)      src/sparse_approximations.jl:58-75
//            approx_log_evidence / elbo              :248-254 (VFE), :282-286 (DTC), :289-305
//            predictive mean / var                   :183-185, :192-195, :212-217
//
// Formulation (S = Σy^-1/2 diagonal, L_z L_zᵀ = K_zz + jitter·I, so U = L_zᵀ):
//   the reference forms B = U⁻ᵀ (S K_xz)ᵀ (M×N, a TRSM over the long N dimension) and D = B Bᵀ + I.
//   Here the N-long pass is ONE streamed SYRK:  G = (K_zx S)(K_zx S)ᵀ  accumulated over column chunks
//   of W = K_zx S (kmat with a column scale → MFMA gemm_nt, dtype T = f32 or f64), and the triangular
//   work moves to the small M×M side in fp64:  D = I + L_z⁻¹ G L_z⁻ᵀ  (two right-solves around a
//   transpose).  This halves the N·M² flops and never holds an N×M matrix (one M×chunk tile is live).
//   Vectors: v = K_zx S² δ (kvec), c = L_z⁻¹ v (= B b_y), w = L_D⁻¹ c, m_ε = L_D⁻ᵀ w, α = L_z⁻ᵀ m_ε.
//   ‖A‖²_F = tr(D) − M (trace term of the ELBO, :251).
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
};

template <typename T>
static int32_t vfe_fit_impl(gp_ctx* c, const gp_kernel* k, const gp_points* x, const gp_points* z, const gp_noise* noise,
                            double jitter, const void* mean_or_null, const void* yv, int approx, gp_vfe* out,
                            double* objective) {
    const long n = x->n, m = z->n, mp = round_up(m, 128);
    const int d = x->d;
    const long CH = 8192;                       // columns (data points) per streamed chunk
    const long npad = round_up(n, CH);
    const long ld = mp + c->ldpad;              // M×M matrices
    const long ldw = CH + c->ldpad;             // W chunk
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
    scale_points<T>(k, z, mp, zsT_h);
    std::vector<double> zsD_h(zsT_h.begin(), zsT_h.end());
    std::vector<T> cs_h((size_t)npad, T(0)), t_h((size_t)npad, T(0));  // s_i = σ_i⁻¹ ; t_i = s_i² δ_i
    double logdet_sy = 0, dd = 0, tr_kff = 0;
    for (long i = 0; i < n; ++i) {
        const double s2 = noise->kind == 0 ? noise->s : (double)((const T*)noise->diag)[i];
        if (!(s2 > 0)) return 1 + (int32_t)i;  // chol(Σy) fails at i (reference :61 / :296)
        const double delta = (double)(T)(y[i] - (mean ? mean[i] : T(0)));
        cs_h[i] = (T)(1.0 / std::sqrt(s2));
        t_h[i] = (T)(delta / s2);
        logdet_sy += std::log(s2);
        dd += delta * delta / s2;
        tr_kff += k->variance / s2;  // tr_Cf_invΣy :307-313
    }
    std::vector<double> jit_h((size_t)mp, 0.0);
    for (long i = 0; i < m; ++i) jit_h[i] = jitter;

    const size_t xs_b = sizeof(T) * xs_h.size(), zsT_b = sizeof(T) * zsT_h.size(), zsD_b = sizeof(double) * zsD_h.size();
    const size_t cs_b = sizeof(T) * (size_t)npad, G_b = sizeof(T) * (size_t)(mp + 128) * ld;
    const size_t W_b = sizeof(T) * (size_t)(mp + 128) * ldw, L_b = sizeof(double) * (size_t)(mp + 128 + 128) * ld;
    const size_t vT_b = sizeof(T) * (size_t)mp, vD_b = sizeof(double) * (size_t)mp * 4, jit_b = sizeof(double) * (size_t)mp;
    void *xs_v = 0, *zsT_v = 0, *zsD_v = 0, *cs_v = 0, *t_v = 0, *G_v = 0, *W_v = 0, *Lz_v = 0, *Ld_v = 0, *Y_v = 0, *vT_v = 0,
         *vec_v = 0, *jit_v = 0;
    RC(ctx_alloc(c, xs_b, &xs_v));
    RC(ctx_alloc(c, zsT_b, &zsT_v));
    RC(ctx_alloc(c, zsD_b, &zsD_v));
    RC(ctx_alloc(c, cs_b, &cs_v));
    RC(ctx_alloc(c, cs_b, &t_v));
    RC(ctx_alloc(c, G_b, &G_v));
    RC(ctx_alloc(c, W_b, &W_v));
    RC(ctx_alloc(c, L_b, &Lz_v));
    RC(ctx_alloc(c, L_b, &Ld_v));
    RC(ctx_alloc(c, L_b, &Y_v));
    RC(ctx_alloc(c, vT_b, &vT_v));
    RC(ctx_alloc(c, vD_b, &vec_v));
    RC(ctx_alloc(c, jit_b, &jit_v));
    double* Lz = (double*)Lz_v;
    double* Ld = (double*)Ld_v;
    double* Yb = (double*)Y_v;
    double* vec = (double*)vec_v;  // rows: [0] v→c , [1] w→m_ε , [2] α , [3] spare
    double scal_h[16] = {0};
    int info_h = 0;
    hipStream_t s = c->sm;

    int32_t rc = [&]() -> int32_t {
        HIPCHK(hipEventRecord(c->ev_phase[0], s));
        HIPCHK(hipMemcpyAsync(xs_v, xs_h.data(), xs_b, hipMemcpyHostToDevice, s));
        HIPCHK(hipMemcpyAsync(zsT_v, zsT_h.data(), zsT_b, hipMemcpyHostToDevice, s));
        HIPCHK(hipMemcpyAsync(zsD_v, zsD_h.data(), zsD_b, hipMemcpyHostToDevice, s));
        HIPCHK(hipMemcpyAsync(cs_v, cs_h.data(), cs_b, hipMemcpyHostToDevice, s));
        HIPCHK(hipMemcpyAsync(t_v, t_h.data(), cs_b, hipMemcpyHostToDevice, s));
        HIPCHK(hipMemcpyAsync(jit_v, jit_h.data(), jit_b, hipMemcpyHostToDevice, s));
        HIPCHK(hipMemsetAsync(c->info_dev, 0, sizeof(int), s));
        HIPCHK(hipMemsetAsync(c->scal_dev, 0, sizeof(double) * 16, s));
        HIPCHK(hipMemsetAsync(G_v, 0, G_b, s));
        HIPCHK(hipMemsetAsync(vec_v, 0, vD_b, s));
        // ---- streamed SYRK over the N data points: G_acc -= W Wᵀ, W = K(z, x_chunk) · S_chunk
        for (long c0 = 0; c0 < npad; c0 += CH) {
            GridMap g = plain_map(0, 0, c0);
            dim3 grid((unsigned)(CH / 128), (unsigned)(mp / 128));
            hipLaunchKernelGGL(kmat_kernel<T>, grid, dim3(256), 0, s, (T*)W_v, ldw, (const T*)zsT_v, mp, (const T*)xs_v,
                               npad, d, k->kind, (T)k->variance, (const T*)nullptr, m, n, 0, g, (const T*)cs_v);
            HIPCHK(hipGetLastError());
            RC(launch_gemm<T>(c, s, (T*)G_v, ld, (const T*)W_v, ldw, (const T*)W_v, ldw, mp, mp, CH, plain_map(1, 0, 0)));
        }
        // v = K_zx S² δ
        hipLaunchKernelGGL(kvec_kernel<T>, dim3((unsigned)m), dim3(256), 0, s, (const T*)zsT_v, mp, (const T*)xs_v, npad, d,
                           k->kind, (T)k->variance, n, (const T*)t_v, (T*)vT_v);
        HIPCHK(hipGetLastError());
        hipLaunchKernelGGL((convert_kernel<T, double>), dim3((unsigned)((m + 255) / 256)), dim3(256), 0, s, (const T*)vT_v, vec,
                           m);
        HIPCHK(hipGetLastError());
        HIPCHK(hipEventRecord(c->ev_phase[1], s));
        // ---- M×M side, fp64.  L_z = chol(K_zz + jitter I)
        {
            GridMap g = plain_map(1, 0, 0);
            dim3 grid((unsigned)(mp / 128), (unsigned)(mp / 128));
            hipLaunchKernelGGL(kmat_kernel<double>, grid, dim3(256), 0, s, Lz, ld, (const double*)zsD_v, mp,
                               (const double*)zsD_v, mp, d, k->kind, k->variance, (const double*)jit_v, m, m, 1, g,
                               (const double*)nullptr);
            HIPCHK(hipGetLastError());
        }
        RC(potrf_full<double>(c, Lz, ld, mp, mp, c->info_dev, m, c->scal_dev + 0));
        // G (symmetric, fp64) -> Y = G L_z⁻ᵀ -> Yᵀ -> Z = Yᵀ L_z⁻ᵀ = L_z⁻¹ G L_z⁻ᵀ -> D = Z + I
        hipLaunchKernelGGL(neg_sym_to_f64_kernel<T>, dim3((unsigned)((mp + 255) / 256), (unsigned)mp), dim3(256), 0, s,
                           (const T*)G_v, ld, Yb, ld, mp);
        HIPCHK(hipGetLastError());
        RC(trsm_rec<double>(c, s, Yb, ld, mp, Lz, ld, mp));
        hipLaunchKernelGGL(transpose_f64_kernel, dim3((unsigned)(mp / 32), (unsigned)(mp / 32)), dim3(256), 0, s, Yb, ld, Ld,
                           ld, mp);
        HIPCHK(hipGetLastError());
        RC(trsm_rec<double>(c, s, Ld, ld, mp, Lz, ld, mp));
        hipLaunchKernelGGL(diag_shift_trace_kernel, dim3(1), dim3(256), 0, s, Ld, ld, mp, m, 1.0, c->scal_dev + 2);
        HIPCHK(hipGetLastError());
        RC(potrf_full<double>(c, Ld, ld, mp, mp, c->info_dev + 0, mp, c->scal_dev + 1));
        // ---- vectors
        RC(trsv<double>(c, s, Lz, ld, mp, vec, mp, 1, true));                     // c = L_z⁻¹ v
        HIPCHK(hipMemcpyAsync(vec + mp, vec, sizeof(double) * mp, hipMemcpyDeviceToDevice, s));
        RC(trsv<double>(c, s, Ld, ld, mp, vec + mp, mp, 1, true));                // w = L_D⁻¹ c
        hipLaunchKernelGGL(rowsumsq_kernel<double>, dim3(1), dim3(256), 0, s, vec + mp, mp, mp, c->scal_dev + 3);
        HIPCHK(hipGetLastError());
        RC(trsv<double>(c, s, Ld, ld, mp, vec + mp, mp, 1, false));               // m_ε = L_D⁻ᵀ w
        HIPCHK(hipMemcpyAsync(vec + 2 * mp, vec + mp, sizeof(double) * mp, hipMemcpyDeviceToDevice, s));
        RC(trsv<double>(c, s, Lz, ld, mp, vec + 2 * mp, mp, 1, false));           // α = L_z⁻ᵀ m_ε
        HIPCHK(hipEventRecord(c->ev_phase[2], s));
        HIPCHK(hipEventRecord(c->ev_phase[3], s));
        HIPCHK(hipMemcpyAsync(&info_h, c->info_dev, sizeof(int), hipMemcpyDeviceToHost, s));
        HIPCHK(hipMemcpyAsync(scal_h, c->scal_dev, sizeof(double) * 16, hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        float ms;
        HIPCHK(hipEventElapsedTime(&ms, c->ev_phase[0], c->ev_phase[1]));
        c->tm.assemble_ms = ms;  // streamed kmat + SYRK phase
        HIPCHK(hipEventElapsedTime(&ms, c->ev_phase[1], c->ev_phase[2]));
        c->tm.potrf_ms = ms;
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
    ctx_release(c, zsT_v, zsT_b);
    ctx_release(c, cs_v, cs_b);
    ctx_release(c, t_v, cs_b);
    ctx_release(c, G_v, G_b);
    ctx_release(c, W_v, W_b);
    ctx_release(c, Y_v, L_b);
    ctx_release(c, vT_v, vT_b);
    ctx_release(c, jit_v, jit_b);
    if (rc == 0 && info_h != 0) rc = info_h;
    if (rc != 0 || !out) {
        ctx_release(c, Lz_v, L_b);
        ctx_release(c, Ld_v, L_b);
        ctx_release(c, zsD_v, zsD_b);
        ctx_release(c, vec_v, vD_b);
        if (rc != 0) return rc;
    }
    // objective: dtc = -½ (N log2π + logdet Σy + logdet Λ_ε + ‖δ_s‖² − ‖Λ_ε.U⁻ᵀ A δ_s‖²)          :302-303
    //            elbo = dtc − ½ (tr(K_ff Σy⁻¹) − ‖A‖²_F),  ‖A‖²_F = tr(D − I)                      :251
    const double logdet_lam = 2.0 * scal_h[1], trZ = scal_h[2], quad = scal_h[3];  // trZ = tr(D − I) = ‖A‖²_F
    double obj = -0.5 * ((double)n * LOG2PI + logdet_sy + logdet_lam + dd - quad);
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
                               (const double*)nullptr);
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
