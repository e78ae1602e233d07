// leaf.hpp — the register-resident 64-column leaf of the blocked Cholesky (fp64), its own translation unit (leaf.hip):
// it is compiled with -mllvm -amdgpu-mfma-vgpr-form so that the MFMA accumulators of the diagonal chain live in VGPRs — with the
// default AGPR form every column of the 16×16 factorisation paid 16 v_accvgpr_read copies to get at two registers (≈ 100 of its
// 400 cycles), and the flag is per translation unit (the GEMM kernels keep the AGPR form they were tuned with).
#pragma once
#include "kcommon.hpp"

namespace gpmi {

// ------------------------------------------------------------------------------------------------
// panel64v2 (fp64): the 64-column leaf with everything in REGISTERS — round 4's answer to the leaf chain (the round-3 leaf spent
//   32 000 of its 47 000 cycles in four serial 16×16 factorisations on 16 of 64 lanes, 6 700 in a load phase that moved the whole
//   X slab through LDS before the first pivot, and its X products read every operand from LDS).  Same contract as panel64_kernel.
//
//   Register layouts of a 16×16 block (lane = (li, lg) = (lane & 15, lane >> 4); MFMA f64 16×16×4: A operand lane (i, k), B operand
//   lane (k, j), accumulator register r of lane (li, lg) = element (row lg + 4r, col li)):
//     natural   n[s] = T[li][4lg + s]            what a lane loads / stores as ONE 32-byte piece of a row;  as a B operand of slice s
//                                                it is T's column 4lg+s, as an A operand the same — every X / D block lives like this
//     symmetric a[r] = A[lg + 4r][li]            the accumulator layout itself, for the diagonal blocks (read from the lower triangle)
//   Products (π(i) = 4(i mod 4) + i div 4; verified lane by lane against NumPy in tools/leaf_emu.py):
//     P1  T' = T' ± T·Mᵀ      acc = mfma(±M[π(li)][4lg + s], n[s], acc), s = 0..3 — the accumulator comes out in the NATURAL layout
//                              of T', so chains of left-multiplications (X Inv_jᵀ, then X_c −= X_j L_cjᵀ) never leave the registers;
//                              the small operand M (Inv_j, L_cj) is read from LDS as 4 contiguous doubles per lane
//     P2  A −= L·Lᵀ           acc = mfma(−n[s], n[s], acc) with L in natural registers serving as BOTH operands (diagonal blocks)
//     P3  16×16 factorisation one column at a time on the accumulator: pivot by v_readlane, 1/√ by v_rsq + Newton, the scaled column
//                              masked to its lane group is the K-slot operand of ONE MFMA that applies the rank-1 update to the whole
//                              block; the identity rides along transposed in a second accumulator (Wᵀ −= l·wᵀ), which leaves
//                              L⁻¹ = Inv_j without a single extra VALU instruction on the chain.  ≈ 15 VALU + 2 MFMA per column on all
//                              64 lanes instead of ≈ 60 fp64 VALU on 16 lanes.
//   Roles: wave 0 runs the diagonal chain (P3 of block j).  Waves 1–3 own the other row tiles: wave w the rows 16w..16w+15 of the
//   diagonal tile (blocks (w, c), c < w, natural; block (w, w) symmetric) and XR/16 row tiles of X spread over the three.
//   Step j: wave 0 factors block (j, j), publishes Inv_j                                 | barrier B1
//           owners: X_j ← X_j Inv_jᵀ (stored to global at once), D(t, j) ← D(t, j) Inv_jᵀ published as L(t, j), D(t, t) −= L(t, j) L(t, j)ᵀ;
//                   the owner of row tile j+1 goes first and hands block (j+1, j+1) to wave 0  | barrier B2
//           wave 0 factors block (j+1, j+1)  ∥  owners: X_c −= X_j L(c, j)ᵀ, D(t, c) −= D(t, j) L(c, j)ᵀ for c > j
//   Loads: every wave loads exactly what it owns straight into registers (32-byte pieces); wave 0 needs 4 doubles per lane before the
//   first pivot.  LDS holds only the published factor blocks (45 KB instead of 111 KB), so several leaves fit on a CU.
//   kpre > 0 (left-looking entry, as in panel64_kernel): the kpre tiles to the left are applied first, L_k staged in LDS, the left
//   tile's rows of this workgroup loaded in natural layout and used as the B operands of P1 / both operands of P2.
// ------------------------------------------------------------------------------------------------
// The body of one wave, specialised on the wave's role W (0: the pivot chain; 1..3: owners): what a wave owns is then known at compile
// time — no exec-masked branch around any MFMA (the first version tested `t < nt`, `c < w` at run time: every such test became an
// s_and_saveexec / s_cbranch pair that also fenced the scheduler, so operand reads from LDS were issued right before their use and
// 140–160 cycles went by per MFMA instead of 62–70) — and the independent accumulators of a phase are updated slice by slice,
// back to back.  All four instances execute the same barriers.
template <int XR, int W>
__device__ __forceinline__ void leaf_wave(double* __restrict__ A, long lda, int mrows, int* __restrict__ info, int col0, int n_valid,
                                          double* __restrict__ logdet_acc, int* __restrict__ ticket, int kpre, double* __restrict__ Lp,
                                          double* __restrict__ Inv, double* __restrict__ dAx, int* __restrict__ writer_s) {
    using TR = Tr<double>;
    constexpr int LDP = 66, LIP = 18;            // row pitches (doubles): 16-byte aligned rows for the 4-double operand reads
    constexpr int NXT = XR / 16;                 // row tiles of X per workgroup
    constexpr int NTW = (NXT + 2) / 3;           // X row tiles per owner wave (waves 1..3)
    constexpr int T0 = W == 0 ? 0 : (W - 1) * NTW;
    constexpr int NT = W == 0 ? 0 : ((NXT - T0) < 0 ? 0 : ((NXT - T0) < NTW ? (NXT - T0) : NTW));
    constexpr int NTA = NT > 0 ? NT : 1;         // array extent
    const int tid = threadIdx.x, lane = tid & 63;
    const int li = lane & 15, lg = lane >> 4;
#ifdef GPMI_PANEL_STAMPS
    long stamps[20];
    int nst = 0;
#define PSTAMP2() do { if (nst < 20) stamps[nst++] = (long)__builtin_readcyclecounter(); } while (0)
#else
#define PSTAMP2() do { } while (0)
#endif
    PSTAMP2();
    const int pirow = 4 * (li & 3) + (li >> 2);  // π(li)
    int xrows = mrows - (int)blockIdx.x * XR;
    xrows = xrows < 0 ? 0 : (xrows > XR ? XR : xrows);
    double* const Xg = A + (long)(64 + (long)blockIdx.x * XR) * lda;

    auto ld4 = [&](const double* p16) -> d4_t {  // 32 bytes as two 16-byte pieces (global or LDS)
        const d2_t* src = reinterpret_cast<const d2_t*>(p16);
        const d2_t lo = src[0], hi = src[1];
        d4_t v;
        v[0] = lo[0]; v[1] = lo[1]; v[2] = hi[0]; v[3] = hi[1];
        return v;
    };
    auto st4 = [&](double* p16, const d4_t& v) {
        d2_t lo, hi;
        lo[0] = v[0]; lo[1] = v[1]; hi[0] = v[2]; hi[1] = v[3];
        d2_t* dst = reinterpret_cast<d2_t*>(p16);
        dst[0] = lo;
        dst[1] = hi;
    };
    auto zero4 = [&]() -> d4_t {
        d4_t v;
        v[0] = v[1] = v[2] = v[3] = 0.0;
        return v;
    };
    auto neg4 = [&](const d4_t& v) -> d4_t {
        d4_t r;
        r[0] = -v[0]; r[1] = -v[1]; r[2] = -v[2]; r[3] = -v[3];
        return r;
    };
    // A operand of P1 for the 16×16 block at M (LDS): 4 contiguous doubles of row π(li)
    auto aop = [&](const double* M, int pitch) -> d4_t { return ld4(M + pirow * pitch + 4 * lg); };
    auto ld_x = [&](int t, long coloff) -> d4_t {  // this wave's X row tile t, 32 bytes at column coloff + 4 lg (zero past the last row)
        const int row = 16 * (T0 + t) + li;
        return row < xrows ? ld4(Xg + (long)row * lda + coloff + 4 * lg) : zero4();
    };
    auto ld_sym = [&](int t) -> d4_t {  // block (t, t) from the lower triangle, symmetric layout
        d4_t v;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int rho = lg + 4 * r, rr = rho > li ? rho : li, cc = rho > li ? li : rho;
            v[r] = A[(long)(16 * t + rr) * lda + 16 * t + cc];
        }
        return v;
    };

    // ---- what this wave owns, loaded straight into registers
    d4_t dA = ld_sym(W);  // wave 0: the block being factored; owner W: block (W, W)
    d4_t y[3];            // owner W: blocks (W, c), c < W
    d4_t x[NTA][4];       // X row tiles
#pragma unroll
    for (int c = 0; c < 3; ++c) y[c] = c < W ? ld4(A + (long)(16 * W + li) * lda + 16 * c + 4 * lg) : zero4();
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int c = 0; c < 4; ++c) x[t][c] = ld_x(t, 16 * c);

    // ---- left-looking pre-update by the kpre tiles to the left (same rows): [D; X] −= [L_k; X_k] · L_kᵀ
    for (int k = 0; k < kpre; ++k) {
        const long coff = -64L * (kpre - k);
        if (k > 0) __syncthreads();  // the previous tile's operand reads are done
#pragma unroll
        for (int i = 0; i < 4; ++i) {  // stage L_k (64×64) into Lp: 1 024 pieces of 32 bytes over 256 threads
            const int e = tid + 256 * i, row = e >> 4, pc = e & 15;
            st4(&Lp[row * LDP + 4 * pc], ld4(A + (long)row * lda + coff + 4 * pc));
        }
        d4_t xk[NTA][4];  // the left tile's rows of this wave: requested before the barrier, consumed slice by slice behind it
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) xk[t][q] = ld_x(t, coff + 16 * q);
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const d4_t dk = ld4(&Lp[(16 * W + li) * LDP + 16 * q + 4 * lg]);  // rows 16W.. of the left tile (natural), from the staged image
            d4_t na[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) na[c] = neg4(aop(&Lp[(16 * c) * LDP + 16 * q], LDP));
            const d4_t ndk = neg4(dk);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                dA = TR::mfma(ndk[s], dk[s], dA);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    if (c < W && c < 3) y[c] = TR::mfma(na[c][s], dk[s], y[c]);
#pragma unroll
                    for (int t = 0; t < NT; ++t) x[t][c] = TR::mfma(na[c][s], xk[t][q][s], x[t][c]);
                }
            }
        }
    }
    if (kpre > 0) __syncthreads();  // Lp is free for the published blocks

    int bad = 0, tk_old = -1;
    double mydiag[4] = {1.0, 1.0, 1.0, 1.0};  // wave 0, lane (c, c & 3): L_cc of column 16j + c
    PSTAMP2();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        PSTAMP2();
        if constexpr (W == 0) {
            // P3: factor block (j, j) column by column on the accumulator; the identity rides transposed (accW).
            // What the measurements say (tools/lat_probe2.hip, tools/f16_probe.hip): fp64 VALU and the fp64 MFMA share ONE pipe on a
            // SIMD (an MFMA followed by 8 independent fma takes 62 + 56 cycles) and a VALU read of an MFMA result waits ≈ 30 cycles
            // longer than a dependent MFMA, so a column costs roughly the SUM of what this wave issues — ≈ 300 cycles, against ≈ 500
            // on 16 of 64 lanes in the round-3 leaf.  The pivot of column c+1 is formed from the value before the update and the
            // multiplier (A[c+1][c+1] − L[c+1][c]², one fma), so its 1/√ chain is issued between the two MFMAs of column c instead of
            // behind them; the finished columns are kept by v_cndmask (Ls, Ws) and written once per block; a non-positive pivot turns
            // every later column into NaN, so the LAPACK info is read off the saved diagonal afterwards.
            // (Tried and measured slower IN this kernel, although faster stand-alone: the LDLᵀ form with reciprocal pivots and raw
            // accumulator registers as B operands — 380 cycles per column — and streaming the inverse's rank-1 updates to another
            // wave's SIMD — the consumer's polling and the hand-over at the block end cost more than the 62 cycles per column saved.)
            d4_t accA = dA, accW, Ls, Ws;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                accW[r] = (li == lg + 4 * r) ? 1.0 : 0.0;
                Ls[r] = 0.0;
                Ws[r] = 0.0;
            }
            double sel;
            {
                const double piv = lane_bcast<double>(accA[0], 0);
                const double ri = fast_rsqrt<double>(piv);
                sel = (lg == 0) ? ri : 0.0;
            }
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                const int k = c & 3, p = c >> 2;
                const double pa = accA[p] * sel;   // lane (i, k): L[i][c]; zero outside lane group k
                const double npa = -pa;
                double l = 0.0, dnext = 1.0;
                if (c < 15) {
                    const int c1 = c + 1, k1 = c1 & 3, p1i = c1 >> 2;
                    l = lane_bcast<double>(pa, 16 * k + c1);               // L[c+1][c]
                    dnext = lane_bcast<double>(accA[p1i], 16 * k1 + c1);  // A[c+1][c+1] before this column's update
                    accA = TR::mfma(npa, pa, accA);
                }
                __builtin_amdgcn_sched_barrier(0);
                const double pw = accW[p] * sel;  // lane (i, k): (L⁻ᵀ)[i][c] = Inv[c][i]
                Ls[p] = (lg == k) ? pa : Ls[p];
                Ws[p] = (lg == k) ? pw : Ws[p];
                if (c < 15) {
                    const int k1 = (c + 1) & 3;
                    const double piv = fma(-l, l, dnext);
                    const double ri = fast_rsqrt<double>(piv);  // wave-uniform
                    sel = (lg == k1) ? ri : 0.0;
                    __builtin_amdgcn_sched_barrier(0);
                    accW = TR::mfma(npa, pw, accW);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            // Ls[r] of lane (li, lg) = L[li][lg + 4r] (garbage above the diagonal), Ws[r] = (L⁻ᵀ)[li][lg + 4r] = Inv[lg + 4r][li]
            double dg = 1.0;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int col = lg + 4 * r;
                Lp[(16 * j + li) * LDP + 16 * j + col] = (li >= col) ? Ls[r] : 0.0;
                Inv[j * 16 * LIP + col * LIP + li] = Ws[r];
                if (col == li) dg = Ls[r];
            }
            mydiag[j] = dg;  // lanes (c, c & 3): L_cc of column 16j + c
            if (bad == 0) {
                const unsigned long notpos = __ballot((lg == (li & 3)) && !(dg > 0.0));  // bit of lane (c, c & 3)
                if (notpos) {
                    int first = 16;
#pragma unroll
                    for (int c = 15; c >= 0; --c)
                        if (notpos & (1ul << (16 * (c & 3) + c))) first = c;
                    bad = 16 * j + first + 1;
                }
            }
        }
        // the ticket below says "this workgroup has READ the input tile": a workgroup barrier does not drain vmcnt, so do it by hand
        if (j == 0 && W != 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        PSTAMP2();
        __syncthreads();  // B1: Inv_j and L(j, j) are published
        PSTAMP2();
        if (j == 0 && W == 3 && lane == 0) tk_old = atomicAdd(ticket, 1);  // every load of the input tile has landed; the reply is awaited at the end
        d4_t ai = zero4();
        if constexpr (W != 0) ai = aop(&Inv[j * 16 * LIP], LIP);
        if constexpr (W != 0) {
            if (W > j) {  // this wave's block (W, j): the owner of row tile j+1 is on the critical path (it hands block (j+1, j+1) to wave 0)
                d4_t r = zero4();
#pragma unroll
                for (int s = 0; s < 4; ++s) r = TR::mfma(ai[s], y[j < 3 ? j : 0][s], r);
                y[j < 3 ? j : 0] = r;
                st4(&Lp[(16 * W + li) * LDP + 16 * j + 4 * lg], r);
                const d4_t nr = neg4(r);
#pragma unroll
                for (int s = 0; s < 4; ++s) dA = TR::mfma(nr[s], r[s], dA);
                if (W == j + 1) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) dAx[lane * 4 + q] = dA[q];
                }
            }
        }
        PSTAMP2();
        __syncthreads();  // B2: L(t, j) of every row tile below and the next diagonal block are published
        if constexpr (W == 0) {
            if (j < 3) {
#pragma unroll
                for (int r = 0; r < 4; ++r) dA[r] = dAx[lane * 4 + r];
            }
        } else {
            // X_j ← X_j Inv_jᵀ needs only Inv_j: it runs behind B2, in the shadow of wave 0's next block; the operands of the updates are
            // requested first so that their LDS round trip hides behind these MFMAs
            d4_t na[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) na[c] = (c > j) ? neg4(aop(&Lp[(16 * c) * LDP + 16 * j], LDP)) : zero4();
            if constexpr (NT > 0) {
                d4_t r[NTA];
#pragma unroll
                for (int t = 0; t < NT; ++t) r[t] = zero4();
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int t = 0; t < NT; ++t) r[t] = TR::mfma(ai[s], x[t][j][s], r[t]);
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    x[t][j] = r[t];
                    const int row = 16 * (T0 + t) + li;
                    if (row < xrows) st4(Xg + (long)row * lda + 16 * j + 4 * lg, r[t]);  // column tile j of X is final
                }
            }
            if (j < 3) {
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int c = j + 1; c < 4; ++c) {
                        if (c < W && c < 3) y[c] = TR::mfma(na[c][s], y[j][s], y[c]);
#pragma unroll
                        for (int t = 0; t < NT; ++t) x[t][c] = TR::mfma(na[c][s], x[t][j][s], x[t][c]);
                    }
            }
        }
    }
    PSTAMP2();
    if (W == 3 && lane == 0) *writer_s = (tk_old == (int)gridDim.x - 1);
    __syncthreads();  // every published block is in Lp; writer_s is visible
    if (*writer_s) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int e = tid + 256 * i, row = e >> 6, c = e & 63;
            if (c <= row) A[(long)row * lda + c] = Lp[row * LDP + c];
        }
        if constexpr (W == 0) {
            double logd = 0.0;
            if (lg == (li & 3)) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (col0 + 16 * j + li < n_valid) logd += log(mydiag[j]);
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) logd += __shfl_xor(logd, o, 64);
            if (lane == 0) {
                if (logdet_acc) atomicAdd(logdet_acc, logd);
                if (bad && info && *info == 0) *info = col0 + bad;
                *ticket = 0;
            }
        }
    }
#ifdef GPMI_PANEL_STAMPS
    PSTAMP2();
    if (blockIdx.x == 0 && lane == 0 && W < 2 && logdet_acc) {  // wave 0 at +8, wave 1 at +32 (cycles since the wave's first stamp)
        long* dst = reinterpret_cast<long*>(logdet_acc) + 8 + 24 * W;
        for (int i = 0; i < 20; ++i) dst[i] = i < nst ? stamps[i] - stamps[0] : 0;
    }
#endif
#undef PSTAMP2
}

template <int XR>
__global__ __launch_bounds__(256) void panel64v2_kernel(double* __restrict__ A, long lda, int mrows, int* __restrict__ info, int col0,
                                                         int n_valid, double* __restrict__ logdet_acc, int* __restrict__ ticket,
                                                         int kpre) {
    __shared__ __attribute__((aligned(16))) double Lp[64 * 66];      // published blocks of L (natural rows); pre-update: the left tile
    __shared__ __attribute__((aligned(16))) double Inv[4 * 16 * 18];
    __shared__ __attribute__((aligned(16))) double dAx[64 * 4];      // hand-over of the next diagonal block (symmetric layout)
    __shared__ int writer_s;
    const int w = threadIdx.x >> 6;  // wave-uniform: each wave runs the instance of its role
    if (w == 0) leaf_wave<XR, 0>(A, lda, mrows, info, col0, n_valid, logdet_acc, ticket, kpre, Lp, Inv, dAx, &writer_s);
    else if (w == 1) leaf_wave<XR, 1>(A, lda, mrows, info, col0, n_valid, logdet_acc, ticket, kpre, Lp, Inv, dAx, &writer_s);
    else if (w == 2) leaf_wave<XR, 2>(A, lda, mrows, info, col0, n_valid, logdet_acc, ticket, kpre, Lp, Inv, dAx, &writer_s);
    else leaf_wave<XR, 3>(A, lda, mrows, info, col0, n_valid, logdet_acc, ticket, kpre, Lp, Inv, dAx, &writer_s);
}

}  // namespace gpmi
