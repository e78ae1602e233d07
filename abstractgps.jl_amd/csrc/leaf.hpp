// leaf.hpp — the register-resident 64- / 128-column leaf of the blocked Cholesky (fp64) and the in-panel update kernel in the same style
// (panel_updk_kernel, at the end), their own translation unit (leaf.hip):
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
//   Step j (ONE workgroup barrier per step):
//           wave 0 factors block (j, j) and publishes Inv_j, L(j, j); the owner of row tile j+1 has meanwhile brought its blocks (j+1, j)
//           and (j+1, j+1) up to date with the columns before j (left-looking) and leaves them RAW in LDS (yx, dAx)        | barrier B1
//           wave 0: block (j+1, j) ← raw · Inv_jᵀ, published as L(j+1, j); block (j+1, j+1) −= L(j+1, j) L(j+1, j)ᵀ; goes straight on with
//                   the factorisation of block j+1 — the pivot chain never waits for another wave's arithmetic
//           owners: X_j ← X_j Inv_jᵀ (stored to global at once), D(t, j) ← D(t, j) Inv_jᵀ published as L(t, j) for t > j+1; they meet at a
//                   counter in LDS (`pub`: 4 increments per step, wave 0's included) before the right-looking updates D(t, c) −= …, c > j,
//                   that read the blocks the others published.  X is updated left-looking (x[t][j] −= Σ_{c<j} x[t][c] L(j, c)ᵀ right
//                   before its solve), so X tiles are loaded lazily, two steps ahead of their use.
//   Loads: every wave loads exactly what it owns straight into registers (32-byte pieces); wave 0 needs 4 doubles per lane before the
//   first pivot.  LDS holds only the published factor blocks (46 KB for 64 columns instead of 111 KB; 152 KB for the 128-column leaf).
//   kpre > 0 (left-looking entry, as in panel64_kernel): the kpre tiles to the left are applied first, L_k staged in LDS, the left
//   tile's rows of this workgroup loaded in natural layout and used as the B operands of P1 / both operands of P2.
// ------------------------------------------------------------------------------------------------
// The body of one wave, specialised on the wave's role W (0: the pivot chain; 1..3: owners) and on the number of 16-column tiles NC
// of the leaf (4: a 64-column leaf; 8: a 128-column leaf): what a wave owns is then known at compile time — no exec-masked branch
// around any MFMA (the first version tested `t < nt`, `c < w` at run time: every such test became an s_and_saveexec / s_cbranch pair
// that also fenced the scheduler, operand reads from LDS were issued right before their use and 140–160 cycles went by per MFMA
// instead of 62–70) — and the independent accumulators of a phase are updated slice by slice, back to back.
// Ownership: row tile t ≥ 1 of the diagonal tile belongs to wave 1 + (t − 1) mod 3 (blocks (t, c), c < t, natural; block (t, t)
// symmetric); the XR/16 row tiles of X are spread over waves 1..3.
// Synchronisation per step j: ONE workgroup barrier B1 (Inv_j published by wave 0; the raw blocks (j+1, j), (j+1, j+1) left in LDS by their
// owner), after which wave 0 solves and updates the hand-over block itself and goes on; the owners' other blocks (t, j) are solved and
// published after B1 and the owners meet at the LDS counter `pub` (release / acquire fences around it) before the updates that read them.
// All four instances execute the same workgroup barriers.
template <int XR, int W, int NC>
__device__ __forceinline__ void leaf_wave(double* __restrict__ A, long lda, int mrows, int* __restrict__ info, int col0, int n_valid,
                                          double* __restrict__ logdet_acc, int* __restrict__ ticket, int kpre, double* __restrict__ Lp,
                                          double* __restrict__ Inv, double* __restrict__ dAx, double* __restrict__ yx,
                                          int* __restrict__ writer_s, int* __restrict__ pub) {
    using TR = Tr<double>;
    constexpr int NCOL = 16 * NC;                // columns of the leaf
    constexpr int LDP = NCOL + 2, LIP = 18;      // row pitches (doubles): 16-byte aligned rows for the 4-double operand reads
    constexpr int NXT = XR / 16;                 // row tiles of X per workgroup
    // X row tiles per owner wave.  64-column leaf: ⌈NXT/3⌉ each from wave 1 on.  128-column leaf: the rows of the diagonal tile cost the
    // owners 204 / 100 / 144 MFMAs (row tiles {1,4,7} / {2,5} / {3,6}) and an X row tile 144, so X is dealt 1-2-1 (XR = 64) or 2-3-3
    // (XR = 128) to even the waves out (with 2-2-0 wave 1 had 492 MFMAs against 144 of wave 3 and the pivot chain waited for it).
    constexpr int N1 = NC == 8 ? (NXT == 4 ? 1 : 2) : ((NXT + 2) / 3);
    constexpr int N2 = NC == 8 ? (NXT == 4 ? 2 : 3) : ((NXT - N1) < N1 ? (NXT - N1) : N1);
    constexpr int N3 = NXT - N1 - N2;
    static_assert(N3 >= 0, "X row tiles");
    constexpr int T0 = W <= 1 ? 0 : (W == 2 ? N1 : N1 + N2);
    constexpr int NT = W == 0 ? 0 : (W == 1 ? N1 : (W == 2 ? N2 : N3));
    constexpr int NTA = NT > 0 ? NT : 1;         // array extent
    constexpr int ND = W == 0 ? 0 : (NC - W + 2) / 3;  // row tiles of the diagonal tile owned by this wave: t = W + 3 s
    constexpr int NDA = ND > 0 ? ND : 1;
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
    double* const Xg = A + (long)(NCOL + (long)blockIdx.x * XR) * lda;

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
    d4_t dA = W == 0 ? ld_sym(0) : zero4();  // wave 0: the block being factored
    d4_t dAt[NDA];                           // owner: blocks (t, t), t = W + 3 s
    d4_t y[NDA][NC];                         // owner: blocks (t, c), c < t (the rest is never touched)
    d4_t x[NTA][NC];                         // X row tiles
#pragma unroll
    for (int s = 0; s < ND; ++s) {
        const int t = W + 3 * s;
        dAt[s] = ld_sym(t);
#pragma unroll
        for (int c = 0; c < NC; ++c) y[s][c] = c < t ? ld4(A + (long)(16 * t + li) * lda + 16 * c + 4 * lg) : zero4();
    }
    // the rows of the diagonal tile are requested BEFORE the X rows (the compiler may not move memory operations across the asm): the
    // ticket of step 0 needs only them, so the X rows (2·NT·NC requests per lane, the larger half of the prologue's HBM burst) stay in
    // flight behind the first block's factorisation
    asm volatile("" ::: "memory");
    // X is consumed left-looking (column tile c first at step c), so only the first XPRE column tiles are requested here and tile c + XPRE
    // at the top of step c: a wave can have 63 requests in flight — with all 2·NT·NC + 36 of a 128-column leaf in the prologue the issue
    // itself stalled on the first returns and the owners reached the first barrier after 7 400 cycles
    constexpr int XPRE = (NC == 4) ? 4 : 2;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int c = 0; c < NC; ++c) x[t][c] = c < XPRE ? ld_x(t, 16 * c) : zero4();

    // ---- left-looking pre-update by the kpre 64-column tiles to the left (same rows; 64-column leaves only): [D; X] −= [L_k; X_k] · L_kᵀ
    if constexpr (NC == 4) {
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
                    if constexpr (W == 0) dA = TR::mfma(ndk[s], dk[s], dA);
                    if constexpr (W != 0) dAt[0] = TR::mfma(ndk[s], dk[s], dAt[0]);
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        if constexpr (W != 0) {
                            if (c < W) y[0][c] = TR::mfma(na[c][s], dk[s], y[0][c]);
                        }
#pragma unroll
                        for (int t = 0; t < NT; ++t) x[t][c] = TR::mfma(na[c][s], xk[t][q][s], x[t][c]);
                    }
                }
            }
        }
        if (kpre > 0) __syncthreads();  // Lp is free for the published blocks
    }

    int bad = 0, tk_old = -1;
    double mydiag[NC];  // wave 0, lane (c, c & 3): L_cc of column 16j + c
#pragma unroll
    for (int j = 0; j < NC; ++j) mydiag[j] = 1.0;
    PSTAMP2();
#pragma unroll
    for (int j = 0; j < NC; ++j) {
        if (j < 4) PSTAMP2();
        if constexpr (W == 0) {
            // P3: factor block (j, j) column by column on the accumulator; the identity rides transposed (accW).
            // What the measurements say (tools/lat_probe2.hip, tools/f16_probe.hip): fp64 VALU and the fp64 MFMA share ONE pipe on a
            // SIMD (an MFMA followed by 8 independent fma takes 62 + 56 cycles) and a VALU read of an MFMA result waits ≈ 30 cycles
            // longer than a dependent MFMA, so a column costs roughly the SUM of what this wave issues — ≈ 290 cycles, against ≈ 500
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
        if constexpr (W != 0 && NT > 0) {
            if (j + XPRE < NC) {
#pragma unroll
                for (int t = 0; t < NT; ++t) x[t][j + XPRE] = ld_x(t, 16 * (j + XPRE));
            }
            // X is updated LEFT-looking: column tile j receives every update of the steps before it now, in the shadow of wave 0's
            // factorisation of block (j, j) — x_j −= Σ_{c<j} x_c L(j, c)ᵀ (4j MFMAs per row tile; every L(j, c) was published before the
            // owners' rendezvous of step j−1) — so that after B1 only the solve with Inv_j is left.  Together with the right-looking
            // updates of the diagonal tile's rows (most work in the first steps) an owner's work per step is nearly constant and stays
            // below the pivot chain's; right-looking X updates put 7/8 of a 128-column leaf's X work into its first steps, where wave 0
            // then waited ≈ 3 000 cycles per step at B1.
            if (j > 0) {
#pragma unroll
                for (int c = 0; c < j; ++c) {
                    const d4_t na = neg4(aop(&Lp[(16 * j) * LDP + 16 * c], LDP));
#pragma unroll
                    for (int q = 0; q < 4; ++q)
#pragma unroll
                        for (int t = 0; t < NT; ++t) x[t][j] = TR::mfma(na[q], x[t][c][q], x[t][j]);
                }
            }
        }
        if constexpr (W != 0) {
            // the owner of row tile j+1 leaves the RAW block (j+1, j) and block (j+1, j+1) — complete since the updates of step j−1 — in LDS
            // before B1: wave 0 solves that one block itself right behind its factorisation (it is the only thing between two blocks of
            // the pivot chain), so the chain never waits at a second barrier for another wave's MFMAs
            if (j + 1 < NC && (j + 1 - W) >= 0 && (j + 1 - W) % 3 == 0) {
                const int s = (j + 1 - W) / 3;
                st4(&yx[lane * 4], y[s][j]);
                st4(&dAx[lane * 4], dAt[s]);
            }
        }
        // the ticket below says "this workgroup has READ the input tile": a workgroup barrier does not drain vmcnt, so do it by hand
        // Step 0 needs, of everything requested in the prologue, only what the owner of row tile 1 hands to wave 0: its block (1, 1) and
        // block (1, 0) — the first six requests of wave 1 (requests return in order).  Everything else stays in flight behind B1(0); the
        // ticket ("this workgroup has READ the input tile": a barrier does not drain vmcnt) is taken one step later, when it has landed.
        if (j == 0 && W == 1) {
            constexpr int DLOADS = 4 * ND + 2 * (ND * W + 3 * ND * (ND - 1) / 2);  // 4 per diagonal block + 2 per block (t, c), t = W + 3 s
            constexpr int TOT = DLOADS + 2 * NT * ((NC == 4) ? 4 : 2) + (NC == 4 ? 0 : 2 * NT);  // prologue + the tile requested at the top of step 0
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(TOT - 6 > 63 ? 63 : (TOT - 6 < 0 ? 0 : TOT - 6)) : "memory");
        }
        if (j == 1 && W != 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NC == 4 ? 0 : 2 * NT) : "memory");  // all but the X tile requested at the top of this step
        if (j < 4) PSTAMP2();
        __syncthreads();  // B1: Inv_j and L(j, j) are published
        if (j < 4) PSTAMP2();
        if (j == 1 && W == 3 && lane == 0) tk_old = atomicAdd(ticket, 1);  // every load of the input tile has landed; the reply is awaited at the end
        d4_t ai = zero4();
        if constexpr (W != 0) ai = aop(&Inv[j * 16 * LIP], LIP);
        // one block (t, j) of an owned row tile: solve, publish, update the tile's diagonal block
        auto solve_block = [&](int s) {
            const int t = W + 3 * s;
            d4_t r = zero4();
#pragma unroll
            for (int q = 0; q < 4; ++q) r = TR::mfma(ai[q], y[s][j][q], r);
            y[s][j] = r;
            st4(&Lp[(16 * t + li) * LDP + 16 * j + 4 * lg], r);
            const d4_t nr = neg4(r);
#pragma unroll
            for (int q = 0; q < 4; ++q) dAt[s] = TR::mfma(nr[q], r[q], dAt[s]);
        };
        if constexpr (W == 0) {
            if (j + 1 < NC) {  // block (j+1, j): solve, publish, and the diagonal block (j+1, j+1) of the next factorisation
                const d4_t a0 = aop(&Inv[j * 16 * LIP], LIP);
                const d4_t yr = ld4(&yx[lane * 4]);
                dA = ld4(&dAx[lane * 4]);
                d4_t r = zero4();
#pragma unroll
                for (int q = 0; q < 4; ++q) r = TR::mfma(a0[q], yr[q], r);
                st4(&Lp[(16 * (j + 1) + li) * LDP + 16 * j + 4 * lg], r);
                const d4_t nr = neg4(r);
#pragma unroll
                for (int q = 0; q < 4; ++q) dA = TR::mfma(nr[q], r[q], dA);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // (the store above has long landed: no wait on the chain)
                if (lane == 0) __hip_atomic_fetch_add(pub, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        } else {
            // the other owned row tiles below j+1: solve and publish their blocks (t, j), then tell the other owners
#pragma unroll
            for (int s = 0; s < ND; ++s) {
                const int t = W + 3 * s;
                if (t > j + 1) solve_block(s);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (lane == 0) __hip_atomic_fetch_add(pub, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            // X_j ← X_j Inv_jᵀ needs only Inv_j
            if constexpr (NT > 0) {
                d4_t r[NTA];
#pragma unroll
                for (int t = 0; t < NT; ++t) r[t] = zero4();
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int t = 0; t < NT; ++t) r[t] = TR::mfma(ai[q], x[t][j][q], r[t]);
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    x[t][j] = r[t];
                    const int row = 16 * (T0 + t) + li;
                    if (row < xrows) st4(Xg + (long)row * lda + 16 * j + 4 * lg, r[t]);  // column tile j of X is final
                }
            }
            if (j + 1 < NC) {
                // every L(c, j), c > j, must be published: three owners and wave 0 (block (j+1, j)), one increment each per step
                while (__hip_atomic_load(pub, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < 4 * (j + 1)) __builtin_amdgcn_s_sleep(1);
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#pragma unroll
                for (int c = j + 1; c < NC; ++c) {
                    const d4_t na = neg4(aop(&Lp[(16 * c) * LDP + 16 * j], LDP));
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
#pragma unroll
                        for (int s = 0; s < ND; ++s)
                            if (c < W + 3 * s) y[s][c] = TR::mfma(na[q], y[s][j][q], y[s][c]);
                    }
                }
            }
        }
    }
    PSTAMP2();
    if (W == 3 && lane == 0) *writer_s = (tk_old == (int)gridDim.x - 1);
    __syncthreads();  // every published block is in Lp; writer_s is visible
    if (*writer_s) {
#pragma unroll
        for (int i = 0; i < NCOL * NCOL / 256; ++i) {
            const int e = tid + 256 * i, row = e / NCOL, c = e % NCOL;
            if (c <= row) A[(long)row * lda + c] = Lp[row * LDP + c];
        }
        if constexpr (W == 0) {
            double logd = 0.0;
            if (lg == (li & 3)) {
#pragma unroll
                for (int j = 0; j < NC; ++j)
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

// NC = 4: the 64-column leaf (same contract as panel64_kernel).  NC = 8: a 128-column leaf — mrows counts the rows below the 128×128
// tile, kpre must be 0 (what a 64-column leaf's in-leaf pre-update does for the second half of a 128-column group is here part of the
// ordinary update loop, on registers that are already loaded).
template <int XR, int NC = 4>
__global__ __launch_bounds__(256) void panel64v2_kernel(double* __restrict__ A, long lda, int mrows, int* __restrict__ info, int col0,
                                                         int n_valid, double* __restrict__ logdet_acc, int* __restrict__ ticket,
                                                         int kpre) {
    __shared__ __attribute__((aligned(16))) double Lp[16 * NC * (16 * NC + 2)];  // published blocks of L (natural rows); pre-update: the left tile
    __shared__ __attribute__((aligned(16))) double Inv[NC * 16 * 18];
    __shared__ __attribute__((aligned(16))) double dAx[64 * 4];                  // hand-over of the next diagonal block (symmetric layout)
    __shared__ __attribute__((aligned(16))) double yx[64 * 4];                   // ... and of the raw block (j+1, j) (natural layout)
    __shared__ int writer_s, pub;
    const int w = threadIdx.x >> 6;  // wave-uniform: each wave runs the instance of its role
    if (threadIdx.x == 0) pub = 0;   // (the first barrier inside orders this before any increment)
    if (w == 0) leaf_wave<XR, 0, NC>(A, lda, mrows, info, col0, n_valid, logdet_acc, ticket, kpre, Lp, Inv, dAx, yx, &writer_s, &pub);
    else if (w == 1) leaf_wave<XR, 1, NC>(A, lda, mrows, info, col0, n_valid, logdet_acc, ticket, kpre, Lp, Inv, dAx, yx, &writer_s, &pub);
    else if (w == 2) leaf_wave<XR, 2, NC>(A, lda, mrows, info, col0, n_valid, logdet_acc, ticket, kpre, Lp, Inv, dAx, yx, &writer_s, &pub);
    else leaf_wave<XR, 3, NC>(A, lda, mrows, info, col0, n_valid, logdet_acc, ticket, kpre, Lp, Inv, dAx, yx, &writer_s, &pub);
}

// ------------------------------------------------------------------------------------------------
// panel_updk: the in-panel updates of the recursion inside a panel,  C[m×N] −= P[m×K] · P[0:N, 0:K]ᵀ  with K = N = 128 / 256 / 512 (N a
//   multiple of 128, K of 32): the next N columns of every row below get the product with the first N rows of the block column just
//   factored — those rows hold the diagonal block of what is factored next.  The 128×128-tile GEMM runs these at 15 / 23 / 40 TF/s
//   (profiles/r4/gemm_dump.txt: at most N/128 · m/128 tiles for 512 workgroup slots, so every tile is cut along k and pays a
//   zero-accumulator prologue and 16 384 fp64 atomics per share; 27 µs per K = 128 launch at N = 16 384 for 4 µs of flops).  Here each
//   wave runs 16 rows through the leaf's register chain (accumulators in the natural layout, one 32-byte piece of its own rows of P per
//   16-column slice, MFMAs against the operand blocks), tiled 16·RT rows × 128 columns per workgroup: the 4 waves are RT row tiles × 4/RT
//   column groups — RT = 4 for tall launches, 2 / 1 when 64-row workgroups would leave CUs empty (the launch is latency-bound then:
//   K = 128, m = 4 096: 18.5 / 12.9 / 10.3 µs for RT = 4 / 2 / 1, tools/updk_bench.hip, profiles/r4/updk_bench.txt).  Grid
//   m/(16·RT) × N/128, 2 workgroups per CU; the k range is streamed in chunks of 32 through a double-buffered LDS image of the operand
//   rows (natural rows, stride 34): the next chunk travels global -> registers while the current one feeds the MFMAs, one barrier per
//   chunk.  Tiles entirely above the diagonal of the top block are skipped; the rest of the top block's upper triangle is scratch
//   (leaves read and write the lower one only).  (A single-stage K = N = 128 predecessor with the whole operand in LDS, panel_upd128,
//   was 1–14 µs slower per launch at every m; in the history.)
// ------------------------------------------------------------------------------------------------
#ifndef GPMI_UPDK_ABL
#define GPMI_UPDK_ABL 0  // ablation switches of tools/updk_bench.hip (timing experiments only; 0 in the product build)
#endif
template <int RT>
__global__ __launch_bounds__(256, 2) void panel_updk_kernel(double* __restrict__ C, long ldc, const double* __restrict__ P, long ldp, int m,
                                                            int K) {
    using TR = Tr<double>;
    constexpr int KCH = 32, LDB = KCH + 2;
    constexpr int CG = 4 / RT;    // column groups: the 4 waves are RT row tiles × CG column groups
    constexpr int NCW = 8 / CG;   // 16-column blocks (accumulators) per wave
    __shared__ __attribute__((aligned(16))) double Bp[2][128 * LDB];
    const int n0 = 128 * (int)blockIdx.y;
    if (n0 > 16 * RT * (int)blockIdx.x + 16 * RT - 1) return;  // above the diagonal of the top block (workgroup-uniform)
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int rt = w / CG, cg = w % CG;
    const int li = lane & 15, lg = lane >> 4;
    const int pirow = 4 * (li & 3) + (li >> 2);
    auto ld4 = [&](const double* p16) -> d4_t {
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
    const long row = (long)blockIdx.x * (16 * RT) + 16 * rt + li;  // this lane's row of C / P
    const bool ok = row < m;
    const double* const Prow = P + (ok ? row : 0) * ldp + 4 * lg;
    double* const Crow = C + (ok ? row : 0) * ldc + n0 + 16 * NCW * cg + 4 * lg;
    // operand staging map: piece e = tid + 256·i of the 128 × 32 chunk: row e >> 3, 32-byte piece e & 7
    const double* Qsrc[4];
    int qdst[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int e = tid + 256 * i, r = e >> 3, pc = e & 7;
        Qsrc[i] = P + (long)(n0 + r) * ldp + 4 * pc;
        qdst[i] = r * LDB + 4 * pc;
    }
    d4_t acc[NCW], a[2], an[2], g[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) g[i] = ld4(Qsrc[i]);
#pragma unroll
    for (int q = 0; q < 2; ++q) a[q] = ld4(Prow + 16 * q);
#pragma unroll
    for (int c = 0; c < NCW; ++c) {
        if (ok) acc[c] = ld4(Crow + 16 * c);
        else
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[c][q] = 0.0;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) st4(&Bp[0][qdst[i]], g[i]);
    // (everything requested so far is settled before the loop: a wait that is only needed by the first chunk would otherwise be paid,
    //  as vmcnt(0) behind the loads of the NEXT chunk, by every chunk)
#pragma unroll
    for (int q = 0; q < 2; ++q) asm volatile("" : "+v"(a[q][0]), "+v"(a[q][1]), "+v"(a[q][2]), "+v"(a[q][3]));
#pragma unroll
    for (int c = 0; c < NCW; ++c) asm volatile("" : "+v"(acc[c][0]), "+v"(acc[c][1]), "+v"(acc[c][2]), "+v"(acc[c][3]));
    __syncthreads();
    const int nch = K / KCH;
    const int brow = (16 * NCW * cg + pirow) * LDB + 4 * lg;
    for (int kc = 0; kc < nch; ++kc) {
        const int cur = kc & 1;
        const bool more = kc + 1 < nch;
        if (more) {
#if !(GPMI_UPDK_ABL & 2)
#pragma unroll
            for (int i = 0; i < 4; ++i) g[i] = ld4(Qsrc[i] + (kc + 1) * KCH);
#endif
#if !(GPMI_UPDK_ABL & 1)
#pragma unroll
            for (int q = 0; q < 2; ++q) an[q] = ld4(Prow + (kc + 1) * KCH + 16 * q);
#else
            an[0] = a[1]; an[1] = a[0];
#endif
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            d4_t nb[NCW];
#pragma unroll
            for (int c = 0; c < NCW; ++c) {
                const d4_t b = ld4(&Bp[cur][brow + 16 * c * LDB + 16 * q]);
                nb[c][0] = -b[0]; nb[c][1] = -b[1]; nb[c][2] = -b[2]; nb[c][3] = -b[3];
            }
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int c = 0; c < NCW; ++c) acc[c] = TR::mfma(nb[c][s], a[q][s], acc[c]);
        }
        if (more) {
#if !(GPMI_UPDK_ABL & 2)
#pragma unroll
            for (int i = 0; i < 4; ++i) st4(&Bp[cur ^ 1][qdst[i]], g[i]);
#endif
            // the rows of P for the next chunk have had the whole chunk to arrive: settle them HERE, so that no load is pending across the
            // back edge (otherwise the first MFMA of the next chunk waits with vmcnt(0) behind the loads that chunk has just issued)
#pragma unroll
            for (int q = 0; q < 2; ++q) asm volatile("" : "+v"(an[q][0]), "+v"(an[q][1]), "+v"(an[q][2]), "+v"(an[q][3]));
            a[0] = an[0];
            a[1] = an[1];
        }
#if !(GPMI_UPDK_ABL & 4)
        __syncthreads();
#endif
    }
    if (ok) {
#pragma unroll
        for (int c = 0; c < NCW; ++c) st4(Crow + 16 * c, acc[c]);
    }
}

}  // namespace gpmi
