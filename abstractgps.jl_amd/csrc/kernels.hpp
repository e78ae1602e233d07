// kernels.hpp — hand-written gfx950 (CDNA4) device kernels of libgpmi355.
//
// Storage convention for every matrix touched here: ROW-MAJOR with leading dimension ld, lower
// triangle = the factor L (K + Σy = L Lᵀ).  A row-major lower L is memory-identical to the
// column-major upper C.U that Julia's cholesky returns (reference src/finite_gp_projection.jl:308),
// so the factor can be handed to the host without a transpose.
//
// Kernels (reference call site each one replaces):
//   kmat_kernel        KernelFunctions.kernelmatrix (+ Σy on the diagonal)   src/base_gp.jl:70,74; src/finite_gp_projection.jl:133-136
//   gemm_nt_dma_kernel / gemm_nt_sk_kernel  the SYRK/GEMM trailing update inside LAPACK dpotrf/dtrsm (MFMA)  src/finite_gp_projection.jl:308
//   panel64_kernel     dpotf2 of a 64×64 diagonal tile + Σ log L_ii + X ← X L⁻ᵀ of the rows below it   :308, :310
//   trsm64_mfma_kernel X ← X L⁻ᵀ against a 64×64 tile (dtrsm)               src/util/common_covmat_ops.jl:54-60
//   trsv_*             forward / backward substitutions for vectors (dtrtrs/dpotrs)   src/exact_gpr_posterior.jl:33
//   rowsumsq_kernel    sum(abs2, ·) reductions                              src/util/common_covmat_ops.jl:64-67
//   kvec_kernel        K_*x α without materialising K_*x                    src/exact_gpr_posterior.jl:60-62
#pragma once
#include <atomic>
#include <hip/hip_runtime.h>
#include "engine.hpp"
#ifndef GPMI_GEMM_SCHED
#define GPMI_GEMM_SCHED 1  // explicit MFMA / LDS interleave of the gemm k loop (sched_group_barrier)
#endif
#ifndef GPMI_ABL
#define GPMI_ABL 0  // ablation switches of tools/gemm_ablate.hip (timing experiments only; 0 in the product build)
#endif
#include <stdint.h>
#include <type_traits>
#include "kcommon.hpp"

namespace gpmi {

// (struct GridMap lives in engine.hpp: it is shared with the host-only translation units)
// compact lower-trapezoid enumeration: block b -> (bi, bj); rows i < tri have i+dt+1 tiles, the rest tn
__device__ __forceinline__ void compact_tile(const GridMap& g, int b, int& bi, int& bj) {
    const int tri = (g.tn - g.dt) > 0 ? (g.tn - g.dt) : 0;  // rows of the triangular part (may exceed the grid's rows)
    const long tot_tri = (long)tri * (g.dt + 1) + (long)tri * (tri - 1) / 2;
    if (b < tot_tri) {
        const double q = 2.0 * g.dt + 1.0;
        int i = (int)((-q + sqrt(q * q + 8.0 * (double)b)) * 0.5);
        while ((long)i * (g.dt + 1) + (long)i * (i - 1) / 2 > b) --i;
        while ((long)(i + 1) * (g.dt + 1) + (long)(i + 1) * i / 2 <= b) ++i;
        bi = i;
        bj = b - (int)((long)i * (g.dt + 1) + (long)i * (i - 1) / 2);
    } else {
        const long r = b - tot_tri;
        bi = tri + (int)(r / g.tn);
        bj = (int)(r % g.tn);
    }
}
// XCD-aware order (modes 2, 3).  Workgroup b is dispatched to XCD b % 8, whose 32 CUs hold 64 of these workgroups at a
// time; each XCD has its own 4 MiB L2.  So the 64 consecutive workgroups of one XCD are mapped onto one 8×8 "super-tile"
// of 128×128 tiles: they share 8 A row-panels and 8 B row-panels, i.e. every operand tile fetched into that L2 is reused
// 8×, instead of the B panels being streamed once per workgroup as in a row-major order.  Super-tiles are enumerated
// row-major (mode 2) or over the lower trapezoid (mode 3, super-tile units); tiles outside the matrix return false.
__device__ __forceinline__ bool xcd_tile(const GridMap& g, int b, int& bi, int& bj) {
    const int x = b & 7, q = b >> 3;
    const int S = (q >> 6) * 8 + x, t = q & 63;
    int si, sj;
    if (g.compact == 2) {
        const int tns = (g.tn + 7) >> 3;
        si = S / tns;
        sj = S - si * tns;
    } else {
        GridMap gs = g;
        gs.tn = (g.tn + 7) >> 3;
        gs.dt = (g.dt + 7) >> 3;
        compact_tile(gs, S, si, sj);
    }
    bi = si * 8 + (t >> 3);
    bj = sj * 8 + (t & 7);
    return bi < g.tm && bj < g.tn;
}
__device__ __forceinline__ long glob_idx(long loc, long nb, int P, int p) {
    if (P == 1) return loc;
    return ((loc / nb) * P + p) * nb + (loc % nb);
}

// ------------------------------------------------------------------------------------------------
// gemm_nt_dma: C[M×N] -= A[M×K] · B[N×K]ᵀ (all row-major, k contiguous; M, N multiples of 64, K multiple of 16 f64 / 32 f32).
//   128×128 block tile, 4 waves as 2×2, each wave 64×64 = 4×4 MFMA 16×16 tiles (16 accumulators); rows past M / N
//   (M, N ≡ 64 mod 128) are over-read (allocations carry 128 slack rows) and their results are never stored.  Operand tiles
//   move global -> LDS by the LDS-DMA path (global_load_lds_dwordx4): no staging VGPRs, no ds_write pass — in the
//   register-staged predecessor (gemm_nt_sub, in the history at dbd752d) the 8 ds_write_b128 + lgkmcnt(0) + barrier tail of every
//   k step cost ≈10 % of the MFMA pipe.
//   LDS image per operand and buffer: 128 rows × 128 B, row-major (BK = 16 f64 / 32 f32 = 8 chunks of 16 B).  One
//   wave-instruction fills 1 KiB = 8 whole rows: lane l writes slot l (the DMA destination is base + 16·lane) and
//   FETCHES chunk (l&7) ^ swz(row) of global row (l>>3), swz(r) = (r>>1)&7 — the XOR lives on the source address.
//   Fragment reads use the same involution: lane (li, lg) wants chunk c = 4h+lg of row r -> slot c ^ swz(r); for each
//   16-lane ds_read_b128 group {0-3,12-15,20-27},... this visits all 16 (row parity, slot) pairs: conflict-free.
// ------------------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_void_t;

// The k loop of both MFMA GEMM kernels, software-pipelined across the step boundary (round 5; PIPE = 1).
//   The round-2 loop (PIPE = 0, kept for A/B measurements: ctx parameter "gemm_pipe") issues, after the barrier that ends step s, the
//   8 operand DMAs of step s+2 (≈ 60 scalar / address instructions), then the first 8 ds_read_b128 of step s+1, and only then its
//   first MFMA: with one wave per SIMD (fp32) the matrix pipe idles for the DMA issue + the LDS latency of four waves reading 32 KB at
//   once — ≈ 400–500 of a step's 4 096 MFMA cycles; with two workgroups per CU (fp64) whenever both sit at that point.
//   Here a step is two halves of 16·VEC MFMAs whose fragments are ALREADY in registers when the half begins:
//       half 0 (fragments h = 0 of buffer cur, read during the previous half)   ‖ ds_read of the h = 1 fragments of cur
//       s_waitcnt vmcnt(0) lgkmcnt(0); s_barrier      — buffer cur^1 (step s+1) is complete, nobody reads cur any more
//       half 1 (fragments h = 1)   ‖ the 8 DMAs of step s+2 into cur, one per group of MFMAs ‖ ds_read of the h = 0 fragments of cur^1
//   so that the only thing the matrix pipe ever waits for is the barrier itself.  Two LDS buffers as before; the DMA of step s+2 has a
//   whole step to land.  Register cost: none (a[2][4], b[2][4] were live across the step before; now a0/b0 and a1/b1 alternate).
//   dma_one(j, buf, kt): operand DMA j (0–3: A rows 8·(4j+w).., 4–7: B) of k-step kt into buffer buf.
template <typename T, typename DmaOne>
__device__ __forceinline__ void gemm_kloop_pipe(typename Tr<T>::acc_t (&acc)[4][4], const typename Tr<T>::chunk_t (*As)[128 * 8],
                                                const typename Tr<T>::chunk_t (*Bs)[128 * 8], int fa, int fb, int lg, int sw, int k0,
                                                int k1, DmaOne dma_one) {
    using TR = Tr<T>;
    using chunk_t = typename TR::chunk_t;
    constexpr int VEC = TR::VEC;
    const int sl0 = lg ^ sw, sl1 = (4 + lg) ^ sw;
    chunk_t a0[4], b0[4], a1[4], b1[4];
    {
        const int kn = (k0 + 1 < k1) ? k0 + 1 : k0;
#pragma unroll
        for (int j = 0; j < 8; ++j) dma_one(j, 1, kn);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        a0[t] = As[0][fa + t * 128 + sl0];
        b0[t] = Bs[0][fb + t * 128 + sl0];
    }
    auto mfma_group = [&](const chunk_t (&a)[4], const chunk_t (&b)[4], int g) {  // MFMAs [g·2·VEC, (g+1)·2·VEC) of a half, order (v, mt, nt)
#pragma unroll
        for (int j = 0; j < 2 * VEC; ++j) {
            const int idx = g * 2 * VEC + j, v = idx >> 4, mt = (idx >> 2) & 3, nt = idx & 3;
            acc[mt][nt] = TR::mfma(a[mt][v], b[nt][v], acc[mt][nt]);
        }
    };
    for (int kt = k0; kt < k1; ++kt) {
        const int cur = (kt - k0) & 1;
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            if (g < 4) {
                a1[g] = As[cur][fa + g * 128 + sl1];
                b1[g] = Bs[cur][fb + g * 128 + sl1];
            }
            mfma_group(a0, b0, g);
            __builtin_amdgcn_sched_barrier(0);
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const int kn = (kt + 2 < k1) ? kt + 2 : k1 - 1;  // past the end: a harmless re-fetch of the last step (keeps the loop branch-free)
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            dma_one(g, cur, kn);
            if (g < 4) {
                a0[g] = As[cur ^ 1][fa + g * 128 + sl0];
                b0[g] = Bs[cur ^ 1][fb + g * 128 + sl0];
            }
            mfma_group(a1, b1, g);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // no DMA may still be writing this workgroup's LDS when it is handed on
}

template <typename T, typename CT = T, int PIPE = 1>
__global__ __launch_bounds__(256, 2) void gemm_nt_dma_kernel(CT* C, long ldc, const T* A, long lda, const T* B, long ldb, int M,
                                                              int N, int K, GridMap g) {
    using TR = Tr<T>;
    using chunk_t = typename TR::chunk_t;
    using acc_t = typename TR::acc_t;
    constexpr int VEC = TR::VEC;
    constexpr int BK = 8 * VEC;

    int bi = blockIdx.y, bj = blockIdx.x;
    if (g.compact == 1) compact_tile(g, (int)blockIdx.x, bi, bj);
    else if (g.compact >= 2) {
        if (!xcd_tile(g, (int)blockIdx.x, bi, bj)) return;
    } else if (g.ktri == 1) {
        bi = (int)gridDim.y - 1 - bi;  // triangular k range: the long row tiles are dispatched first
    } else if (g.ktri == 3) {
        // B lower triangular: the k range grows with the COLUMN tile.  Workgroup b runs on XCD b % 8 and blockIdx.x is the fastest index, so
        // without the rotation XCD x would only ever see column tiles x, x + 8, ... (XCD 7: the longest k ranges of every row) — the
        // imbalance measured in round 4 (34 -> 57 TF/s on the explicit-inverse panel product); rotating by the row tile deals them evenly.
        bj = (bj + bi) % (int)gridDim.x;
    }
    if (g.nbatch > 1) {
        C += (long)blockIdx.z * g.cstride;
        A += (long)blockIdx.z * (g.astride ? g.astride : (long)K);  // astride / bstride = 0: split-K (the k range [z·K, (z+1)·K) of both operands)
        B += (long)blockIdx.z * (g.bstride ? g.bstride : (long)K);
    }
    const int m0 = bi * 128, n0 = bj * 128;
    long gr0 = 0, gc0 = 0;
    if (g.lower) {
        gr0 = glob_idx(g.row0 + m0, g.nb, g.P, g.p);
        gc0 = glob_idx(g.col0 + n0, g.nb, g.Q, g.q);
        if (gc0 > gr0 + 127) return;  // whole tile above the diagonal (block-uniform)
    }
    __shared__ __attribute__((aligned(1024))) chunk_t As[2][128 * 8];
    __shared__ __attribute__((aligned(1024))) chunk_t Bs[2][128 * 8];

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = w >> 1, wc = w & 1;
    bool active = (wr * 64 < M - m0) && (wc * 64 < N - n0);
    if (g.lower && (gc0 + wc * 64 > gr0 + wr * 64 + 63)) active = false;

    // DMA map: instruction i of wave w covers rows 8·(4i+w) .. +8; lane -> row 8·(4i+w) + (lane>>3), slot lane&7
    const int drow = lane >> 3;
    const T* Ag[4];
    const T* Bg[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = 8 * (4 * i + w) + drow;
        const int cch = (lane & 7) ^ ((r >> 1) & 7);
        Ag[i] = A + (long)(m0 + r) * lda + cch * VEC;
        Bg[i] = B + (long)(n0 + r) * ldb + cch * VEC;
    }
    // The DMA is issued from inline asm: hipcc (ROCm 7.2) otherwise drains vmcnt(0) in front of the next ds_read of the
    // OTHER buffer (it cannot tell the two apart), which serialises the prefetch.  The only consumer-side wait is the
    // explicit vmcnt(0) in front of the step's barrier below.
    const unsigned ldsA = (unsigned)(unsigned long)(__attribute__((address_space(3))) char*)&As[0][0];
    const unsigned ldsB = (unsigned)(unsigned long)(__attribute__((address_space(3))) char*)&Bs[0][0];
    auto dma1 = [&](const T* src, unsigned dst) {
        unsigned keep;
        const unsigned d = __builtin_amdgcn_readfirstlane(dst);
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "v"(src), "s"(d)
                     : "memory");
    };
    auto dma = [&](int buf, long kt) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            dma1(Ag[i] + kt * BK, ldsA + (unsigned)(buf * 16384 + (4 * i + w) * 1024));
            dma1(Bg[i] + kt * BK, ldsB + (unsigned)(buf * 16384 + (4 * i + w) * 1024));
        }
    };
    auto dma_wait_barrier = [&]() {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };

    const int li = lane & 15, lg = lane >> 4;
    acc_t acc[4][4];
    CT* const Cw = C + (long)(m0 + wr * 64) * ldc + n0 + wc * 64 + li;
    const CT* const Cr = active ? Cw : C + li;
    const int kt0 = (g.ktri == 2 && m0 > g.ktri_off) ? (m0 - g.ktri_off) / BK : 0;  // upper-triangular A (from row ktri_off on): leading zeros skipped
    dma(0, kt0);
    if (g.beta0) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[mt][nt][r] = T(0);
    } else {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[mt][nt][r] = -Cr[(long)(mt * 16 + TR::crow(lane, r)) * ldc + nt * 16];
    }

    // fragment slots: row wr·64 + t·16 + li, chunk 4h+lg -> slot (row·8) + ((4h+lg) ^ swz(li))
    const int sw = (li >> 1) & 7;
    const int fa = (wr * 64 + li) * 8, fb = (wc * 64 + li) * 8;
    int nk = K / BK;
    if (g.ktri == 1) nk = min(nk, (g.ktri_off + m0 + 128) / BK);
    if (g.ktri == 3) nk = min(nk, (n0 + 128) / BK);
    dma_wait_barrier();

    if constexpr (PIPE) {
        // (round 6: a light loop for waves whose 64×64 quarter is never stored — the strictly-upper quarter of a diagonal tile in lower mode: DMAs and
        //  barriers only, no fragments, no MFMA — was measured at C5 (32 of the SYRK's 528 tiles) and at C2 / C3 / C4: no effect, profiles/r6/c5_idle_ab.txt; removed)
        gemm_kloop_pipe<T>(acc, As, Bs, fa, fb, lg, sw, kt0, nk, [&](int j, int buf, long kt) {
            const int i = j & 3;
            if (j < 4) dma1(Ag[i] + kt * BK, ldsA + (unsigned)(buf * 16384 + (4 * i + w) * 1024));
            else dma1(Bg[i] + kt * BK, ldsB + (unsigned)(buf * 16384 + (4 * i + w) * 1024));
        });
    } else
    for (int kt = kt0; kt < nk; ++kt) {
        const int cur = (kt - kt0) & 1;
#if GPMI_ABL & 16  // ablation (fp32 only): no operand DMA inside the loop
        if (sizeof(T) != 4)
#endif
        dma(cur ^ 1, (kt + 1 < nk) ? kt + 1 : kt);  // the last step re-fetches its own tile into the idle buffer
        chunk_t a[2][4], b[2][4];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int sl = (4 * h + lg) ^ sw;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                a[h][t] = As[cur][fa + t * 128 + sl];
                b[h][t] = Bs[cur][fb + t * 128 + sl];
            }
        }
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int v = 0; v < VEC; ++v)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = TR::mfma(a[h][mt][v], b[h][nt][v], acc[mt][nt]);
#if GPMI_GEMM_SCHED
        __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 16 * VEC - 8, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 16 * VEC + 8, 0);
#endif
        __builtin_amdgcn_sched_barrier(0);  // keep every MFMA of this step in front of the vmcnt(0) + barrier
#if GPMI_ABL & 32  // ablation (fp32 only): no wait + barrier at the end of the step
        if (sizeof(T) != 4)
#endif
        dma_wait_barrier();
    }

#if GPMI_ABL & 8  // ablation: no epilogue stores for fp32 (timing experiment only)
    if (sizeof(T) == 4) {
        T sink = T(0);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) sink += acc[mt][nt][0] + acc[mt][nt][1] + acc[mt][nt][2] + acc[mt][nt][3];
        if (sink == T(12345.678)) Cw[0] = sink;
        return;
    }
#endif
    if (active) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) Cw[(long)(mt * 16 + TR::crow(lane, r)) * ldc + nt * 16] = -acc[mt][nt][r];
    }
}

// (Two measured-slower variants of this kernel were removed in round 3 — a three-stage operand ring with one workgroup per CU,
//  gemm_nt_dma3, and a 256×128 tile with one wave per SIMD, gemm_nt_wide: 55 vs 70 TF/s fp64, C5 94 vs 90 ms,
//  profiles/r2/gemm_wide.txt, profiles/r2/c5_ablation.txt; they live in the history at dbd752d.)

// ------------------------------------------------------------------------------------------------
// gemm_nt_sk: gemm_nt_dma with a persistent grid and a stream-K tail (single-GPU maps only: plain rectangle or the
//   compact lower trapezoid).  G workgroups (2 per CU).  Whole "rounds" of G tiles are processed tile-parallel exactly
//   as before (workgroup w takes tiles w, w+G, ...: neighbouring workgroups share operand panels in L2); the last,
//   partial round — R < G tiles, which used to leave G−R workgroup slots idle for a whole tile time — is cut along k:
//   its R·nk k-steps are dealt out evenly to G2 workgroups, each accumulating its share of one (or two) tiles from zero
//   and adding it to C with hardware fp64/fp32 atomics.  Tiles of complete rounds never see atomics.
// ------------------------------------------------------------------------------------------------
template <typename T, int PIPE = 1>
__global__ __launch_bounds__(256, 2) void gemm_nt_sk_kernel(T* C, long ldc, const T* A, long lda, const T* B, long ldb, int M, int N,
                                                             int K, GridMap g, long ntiles, int G2) {
    using TR = Tr<T>;
    using chunk_t = typename TR::chunk_t;
    using acc_t = typename TR::acc_t;
    constexpr int VEC = TR::VEC;
    constexpr int BK = 8 * VEC;
    __shared__ __attribute__((aligned(1024))) chunk_t As[2][128 * 8];
    __shared__ __attribute__((aligned(1024))) chunk_t Bs[2][128 * 8];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = w >> 1, wc = w & 1;
    const int drow = lane >> 3;
    const int li = lane & 15, lg = lane >> 4;
    const int sw = (li >> 1) & 7;
    const int fa = (wr * 64 + li) * 8, fb = (wc * 64 + li) * 8;
    const unsigned ldsA = (unsigned)(unsigned long)(__attribute__((address_space(3))) char*)&As[0][0];
    const unsigned ldsB = (unsigned)(unsigned long)(__attribute__((address_space(3))) char*)&Bs[0][0];
    auto dma1 = [&](const T* src, unsigned dst) {
        unsigned keep;
        const unsigned d = __builtin_amdgcn_readfirstlane(dst);
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "v"(src), "s"(d)
                     : "memory");
    };
    auto dma_wait_barrier = [&]() {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    const int nk = K / BK;
    const long G = gridDim.x;
    const long full = (ntiles / G) * G;          // tiles of complete rounds
    const long R = ntiles - full;                // tail tiles, cut along k
    const long tail_units = R * nk;
    long tu0 = 0, tu1 = 0;                       // my share of the tail's k-steps
    if ((long)blockIdx.x < G2) {
        tu0 = (long)blockIdx.x * tail_units / G2;
        tu1 = ((long)blockIdx.x + 1) * tail_units / G2;
    }
    long next_full = blockIdx.x;                 // next whole tile of mine
    while (true) {
        long tile;
        int k0, k1;
        if (next_full < full) {
            tile = next_full;
            k0 = 0;
            k1 = nk;
            next_full += G;
        } else if (tu0 < tu1) {
            const long t = tu0 / nk;
            tile = full + t;
            k0 = (int)(tu0 - t * nk);
            const long left = tu1 - tu0;
            k1 = (left < (long)(nk - k0)) ? k0 + (int)left : nk;
            tu0 += k1 - k0;
        } else {
            break;
        }
        const bool complete = (k0 == 0 && k1 == nk);
        int bi, bj;
        if (g.compact == 1) compact_tile(g, (int)tile, bi, bj);
        else {
            bi = (int)(tile / g.tn);
            bj = (int)(tile - (long)bi * g.tn);
        }
        const int m0 = bi * 128, n0 = bj * 128;
        const long gr0 = g.row0 + m0, gc0 = g.col0 + n0;
        if (g.lower && gc0 > gr0 + 127) continue;  // tile above the diagonal (workgroup-uniform)
        bool active = (wr * 64 < M - m0) && (wc * 64 < N - n0);
        if (g.lower && (gc0 + wc * 64 > gr0 + wr * 64 + 63)) active = false;
        const T* Ag[4];
        const T* Bg[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = 8 * (4 * i + w) + drow;
            const int cch = (lane & 7) ^ ((r >> 1) & 7);
            Ag[i] = A + (long)(m0 + r) * lda + cch * VEC;
            Bg[i] = B + (long)(n0 + r) * ldb + cch * VEC;
        }
        auto dma = [&](int buf, long kt) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                dma1(Ag[i] + kt * BK, ldsA + (unsigned)(buf * 16384 + (4 * i + w) * 1024));
                dma1(Bg[i] + kt * BK, ldsB + (unsigned)(buf * 16384 + (4 * i + w) * 1024));
            }
        };
        acc_t acc[4][4];
        T* const Cw = C + (long)(m0 + wr * 64) * ldc + n0 + wc * 64 + li;
        const T* const Cr = active ? Cw : C + li;
        dma(0, k0);
        if (complete) {
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[mt][nt][r] = -Cr[(long)(mt * 16 + TR::crow(lane, r)) * ldc + nt * 16];
        } else {
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[mt][nt][r] = T(0);
        }
        dma_wait_barrier();
        if constexpr (PIPE) {
            gemm_kloop_pipe<T>(acc, As, Bs, fa, fb, lg, sw, k0, k1, [&](int j, int buf, long kt) {
                const int i = j & 3;
                if (j < 4) dma1(Ag[i] + kt * BK, ldsA + (unsigned)(buf * 16384 + (4 * i + w) * 1024));
                else dma1(Bg[i] + kt * BK, ldsB + (unsigned)(buf * 16384 + (4 * i + w) * 1024));
            });
        } else
        for (int kt = k0; kt < k1; ++kt) {
            const int cur = (kt - k0) & 1;
            dma(cur ^ 1, (kt + 1 < k1) ? kt + 1 : kt);
            chunk_t a[2][4], b[2][4];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int sl = (4 * h + lg) ^ sw;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    a[h][t] = As[cur][fa + t * 128 + sl];
                    b[h][t] = Bs[cur][fb + t * 128 + sl];
                }
            }
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int v = 0; v < VEC; ++v)
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                        for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = TR::mfma(a[h][mt][v], b[h][nt][v], acc[mt][nt]);
#if GPMI_GEMM_SCHED
            __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 16 * VEC - 8, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 16 * VEC + 8, 0);
#endif
            __builtin_amdgcn_sched_barrier(0);
            dma_wait_barrier();
        }
        if (active) {
            if (complete) {
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) Cw[(long)(mt * 16 + TR::crow(lane, r)) * ldc + nt * 16] = -acc[mt][nt][r];
            } else {
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            unsafeAtomicAdd(&Cw[(long)(mt * 16 + TR::crow(lane, r)) * ldc + nt * 16], -acc[mt][nt][r]);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// kmat: out[r][c] = variance * κ(‖xr_r − xc_c‖) (+ noise_r on the global diagonal when sym)
//   xr / xc: pre-scaled inputs, dimension-major [d][ldx], indexed by GLOBAL point index.
//   128×128 tile per block; wave w owns rows w, w+4, ...; lane l owns columns 2l, 2l+1 so every
//   store instruction writes one contiguous 1 KiB (f64) row segment.
//   Padding (global index >= n_valid): identity when sym, zero otherwise.
//   colscale / rowscale (nullable): column j / row i of the result is multiplied by colscale[j] / rowscale[i]
//   (VFE: Σy^-1/2 K_xz).
// ------------------------------------------------------------------------------------------------
// e^x for x <= 0 (every κ below evaluates exp at a non-positive argument).  fp64: Cody–Waite reduction x = n·ln2 + r with two fma, the
// degree-13 Taylor polynomial of e^r on |r| <= ln2/2 (truncation 4e-18 relative), v_ldexp_f64 — 19 instructions against the ≈ 35 of the
// library exp with its overflow / special-case handling (underflow falls out of ldexp; NaN propagates).  kmat_kernel is VALU-bound, not
// store-bound: the same tile stores with 32 dependent fma per element in front run at 5.7 TB/s, with 48 at 4.4 TB/s, the kernel itself at
// 4.6 TB/s (tools/kmat_probe.hip, round 5).  Agreement with the library exp: <= 2 ulp (tests: |ΔK| <= 1e-14·σ² against the oracle).
template <typename T> __device__ __forceinline__ T exp_nonpos(T x) { return exp(x); }
template <> __device__ __forceinline__ double exp_nonpos<double>(double x) {
    x = (x < -800.0) ? -800.0 : x;  // e^-800 underflows to 0 already; keeps n inside the int range (a NaN stays a NaN)
    const double n = __builtin_rint(x * 1.4426950408889634074);
    double r = fma(n, -0.69314718055994528623, x);
    r = fma(n, -2.3190468138462995584e-17, r);
    double p = 1.6059043836821614599e-10;            // 1/13!
    p = fma(p, r, 2.0876756987868098979e-09);        // 1/12!
    p = fma(p, r, 2.5052108385441718775e-08);        // 1/11!
    p = fma(p, r, 2.7557319223985890653e-07);        // 1/10!
    p = fma(p, r, 2.7557319223985892511e-06);        // 1/9!
    p = fma(p, r, 2.4801587301587301566e-05);        // 1/8!
    p = fma(p, r, 1.9841269841269841253e-04);        // 1/7!
    p = fma(p, r, 1.3888888888888889419e-03);        // 1/6!
    p = fma(p, r, 8.3333333333333332177e-03);        // 1/5!
    p = fma(p, r, 4.1666666666666664354e-02);        // 1/4!
    p = fma(p, r, 1.6666666666666665741e-01);        // 1/3!
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    return ldexp(p, (int)n);
}

template <typename T> __device__ __forceinline__ T kappa(int kind, T d2) {
    if (kind == 0) return exp_nonpos<T>(T(-0.5) * d2);
    const T d = sqrt(d2);
    if (kind == 1) return exp_nonpos<T>(-d);
    if (kind == 2) {
        const T a = T(1.7320508075688772935) * d;
        return (T(1) + a) * exp_nonpos<T>(-a);
    }
    const T a = T(2.2360679774997896964) * d;
    return (T(1) + a + T(5.0 / 3.0) * d2) * exp_nonpos<T>(-a);
}

template <typename T, int KIND, int DR>  // DR = 4 / 8 / 16: D <= DR, row-by-row form; 0: any D
__device__ __forceinline__ void kmat_body(T (*xi)[128], T (*xj)[128], T* __restrict__ out, long ld, const T* __restrict__ xr, long ldxr,
                                                    const T* __restrict__ xc, long ldxc, int d, T variance,
                                                    const T* __restrict__ noise, long nr_valid, long nc_valid, int sym,
                                                    GridMap g, const T* __restrict__ colscale,
                                                    const T* __restrict__ rowscale) {
    using pair_t = typename Tr<T>::pair_t;
    constexpr int DC = 16;
    const int m0 = blockIdx.y * 128, n0 = blockIdx.x * 128;
    const long gr0 = glob_idx(g.row0 + m0, g.nb, g.P, g.p);
    const long gc0 = glob_idx(g.col0 + n0, g.nb, g.Q, g.q);
    if (g.lower && gc0 > gr0 + 127) return;

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const long gj0 = gc0 + 2 * lane, gj1 = gj0 + 1;
    // interior tiles (no padding, not on the diagonal, no row/column scaling): nothing but κ and the 1 KiB row stores
    const bool interior = gr0 + 128 <= nr_valid && gc0 + 128 <= nc_valid && !(sym && gr0 == gc0) && colscale == nullptr && rowscale == nullptr;
    auto emit_interior = [&](int rr, T a0, T a1) {
        pair_t o;
        o.x = variance * kappa<T>(KIND, a0);
        o.y = variance * kappa<T>(KIND, a1);
        *reinterpret_cast<pair_t*>(out + (long)(m0 + w + 4 * rr) * ld + n0 + 2 * lane) = o;
    };
    auto emit_edge = [&](int rr, T a0, T a1) {  // padding (identity when sym), the noise on the global diagonal, row / column scaling
        const int row = w + 4 * rr;
        const long gi = gr0 + row;
        T v0, v1;
        if (gi >= nr_valid) {
            v0 = (sym && gi == gj0) ? T(1) : T(0);
            v1 = (sym && gi == gj1) ? T(1) : T(0);
        } else {
            v0 = (gj0 < nc_valid) ? variance * kappa<T>(KIND, a0) : T(0);
            v1 = (gj1 < nc_valid) ? variance * kappa<T>(KIND, a1) : T(0);
            if (sym && noise != nullptr) {
                if (gi == gj0) v0 += noise[gi];
                if (gi == gj1) v1 += noise[gi];
            }
            if (colscale != nullptr) {
                if (gj0 < nc_valid) v0 *= colscale[gj0];
                if (gj1 < nc_valid) v1 *= colscale[gj1];
            }
            if (rowscale != nullptr) {
                const T rs = rowscale[gi];
                v0 *= rs;
                v1 *= rs;
            }
        }
        pair_t o;
        o.x = v0;
        o.y = v1;
        *reinterpret_cast<pair_t*>(out + (long)(m0 + row) * ld + n0 + 2 * lane) = o;
    };

    if constexpr (DR != 0) {
        // D <= DR (one staged chunk; one kernel instance per DR so that the register count is this path's): row by row — distance, κ, store — with
        // nothing but this lane's two columns of the inputs held in registers (85 / 101 / 133 VGPRs for DR = 4 / 8 / 16 in fp64).  The accumulate form
        // below keeps all 64 squared distances of a thread live across the dimension chunks (162 VGPRs: three waves per SIMD); its counters (round 5,
        // N = 32 768: VALU issuing 31 % of a wave's cycles, 36 % issue-stalled, 28 % parked; 35 VALU instructions per element) say the Gram kernel is bound
        // by what three waves can issue, not by its stores (the same 1 KiB row stores alone: 5.6–5.7 TB/s, tools/kmat_probe.hip).  Same-box A/B
        // ("kmat_rows"): launch_kmat below.  What holds the C4 launch at 4.6 TB/s is where it sits: it is the first heavy kernel after the previous fit's
        // latency-bound tail (tools/kmat_repeat.py: 4.04 TB/s cold, 4.95 after five back-to-back launches).
        for (int e = tid; e < DR * 128; e += 256) {  // rows d..DR-1 zero: the unrolled loops below need no predicate
            const int dd = e >> 7, i = e & 127;
            xi[dd][i] = dd < d ? xr[(long)dd * ldxr + gr0 + i] : T(0);
            xj[dd][i] = dd < d ? xc[(long)dd * ldxc + gc0 + i] : T(0);
        }
        __syncthreads();
        pair_t yv[DR];
#pragma unroll
        for (int dd = 0; dd < DR; ++dd) yv[dd] = *reinterpret_cast<const pair_t*>(&xj[dd][2 * lane]);
        auto dist = [&](int rr, T& a0, T& a1) {
            a0 = T(0);
            a1 = T(0);
#pragma unroll
            for (int dd = 0; dd < DR; ++dd) {
                const T xv = xi[dd][w + 4 * rr];
                const T t0 = xv - yv[dd].x, t1 = xv - yv[dd].y;
                a0 = fma(t0, t0, a0);
                a1 = fma(t1, t1, a1);
            }
        };
        if (interior) {
#pragma unroll 2
            for (int rr = 0; rr < 32; ++rr) {
                T a0, a1;
                dist(rr, a0, a1);
                emit_interior(rr, a0, a1);
            }
        } else {
            for (int rr = 0; rr < 32; ++rr) {
                T a0, a1;
                dist(rr, a0, a1);
                emit_edge(rr, a0, a1);
            }
        }
        return;
    } else {
    T acc0[32], acc1[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) acc0[i] = acc1[i] = T(0);

    for (int d0 = 0; d0 < d; d0 += DC) {
        const int dc = (d - d0 < DC) ? (d - d0) : DC;
        __syncthreads();
        for (int e = tid; e < dc * 128; e += 256) {
            const int dd = e >> 7, i = e & 127;
            xi[dd][i] = xr[(long)(d0 + dd) * ldxr + gr0 + i];
            xj[dd][i] = xc[(long)(d0 + dd) * ldxc + gc0 + i];
        }
        __syncthreads();
        for (int dd = 0; dd < dc; ++dd) {
            const pair_t yv = *reinterpret_cast<const pair_t*>(&xj[dd][2 * lane]);
#pragma unroll
            for (int rr = 0; rr < 32; ++rr) {
                const T xv = xi[dd][w + 4 * rr];
                const T t0 = xv - yv.x, t1 = xv - yv.y;
                acc0[rr] = fma(t0, t0, acc0[rr]);
                acc1[rr] = fma(t1, t1, acc1[rr]);
            }
        }
    }
    if (interior) {
#pragma unroll
        for (int rr = 0; rr < 32; ++rr) emit_interior(rr, acc0[rr], acc1[rr]);
        return;
    }
#pragma unroll
    for (int rr = 0; rr < 32; ++rr) emit_edge(rr, acc0[rr], acc1[rr]);
    }
}

// runtime kernel kind -> compile-time specialisation (the κ branch is hoisted out of the 64-element inner loops).  DR = 4 / 8 / 16: D <= DR, the
// row-by-row form (few registers, high occupancy: one kernel instance per DR so that each gets its own register count); 0: any D, squared
// distances accumulated over chunks of 16 dimensions.  launch_kmat picks.
template <typename T, int DR>
__global__ __launch_bounds__(256) void kmat_kernel(T* __restrict__ out, long ld, const T* __restrict__ xr, long ldxr,
                                                    const T* __restrict__ xc, long ldxc, int d, int kind, T variance,
                                                    const T* __restrict__ noise, long nr_valid, long nc_valid, int sym,
                                                    GridMap g, const T* __restrict__ colscale,
                                                    const T* __restrict__ rowscale) {
    constexpr int LR = DR != 0 ? DR : 16;  // staged dimensions per chunk
    __shared__ T xi[LR][128];
    __shared__ __attribute__((aligned(16))) T xj[LR][128];
    switch (kind) {
        case 0: kmat_body<T, 0, DR>(xi, xj, out, ld, xr, ldxr, xc, ldxc, d, variance, noise, nr_valid, nc_valid, sym, g, colscale, rowscale); break;
        case 1: kmat_body<T, 1, DR>(xi, xj, out, ld, xr, ldxr, xc, ldxc, d, variance, noise, nr_valid, nc_valid, sym, g, colscale, rowscale); break;
        case 2: kmat_body<T, 2, DR>(xi, xj, out, ld, xr, ldxr, xc, ldxc, d, variance, noise, nr_valid, nc_valid, sym, g, colscale, rowscale); break;
        default: kmat_body<T, 3, DR>(xi, xj, out, ld, xr, ldxr, xc, ldxc, d, variance, noise, nr_valid, nc_valid, sym, g, colscale, rowscale); break;
    }
}
static std::atomic<int> g_kmat_rows{1};  // PROCESS-WIDE (every ctx of the process reads it; relaxed atomic: contexts run on their own threads).  1: the row-by-row instances for D <= 16; 0: always the accumulate form (A/B switch: ctx parameter "kmat_rows")
template <typename T>
static inline void launch_kmat(dim3 grid, hipStream_t s, T* out, long ld, const T* xr, long ldxr, const T* xc, long ldxc, int d, int kind, T variance,
                               const T* noise, long nr_valid, long nc_valid, int sym, GridMap g, const T* colscale, const T* rowscale) {
#define GPMI_KMAT_LAUNCH(DR_) \
    hipLaunchKernelGGL((kmat_kernel<T, DR_>), grid, dim3(256), 0, s, out, ld, xr, ldxr, xc, ldxc, d, kind, variance, noise, nr_valid, nc_valid, sym, g, colscale, rowscale)
    // Which form: alternating A/B inside fits on one box (tools/kmat_ab.py, profiles/r5/kmat_ab.jsonl; row form vs accumulate form, ms): N = 8 192 D = 3
    // 0.128 / 0.138, C2 0.277 / 0.286, N = 32 768 D = 3 0.891 / 0.888, C3 (D = 8) 1.32 / 1.42, N = 49 152 D = 3 1.85 / 1.87, N = 65 536 D = 8 6.27 / 6.43 —
    // and C4 (N = 65 536, D = 3) 4.02 / 3.74: with the cheapest distance loop five workgroups per CU stream into more DRAM rows at once than the big
    // matrix tolerates.  So: the row form, except for D <= 4 on launches of more than ≈ 54 000² elements.
    const bool huge = (long)grid.x * (long)grid.y >= 180000;
    if (!g_kmat_rows.load(std::memory_order_relaxed) || (d <= 4 && huge)) GPMI_KMAT_LAUNCH(0);
    else if (d <= 4) GPMI_KMAT_LAUNCH(4);
    else if (d <= 8) GPMI_KMAT_LAUNCH(8);
    else if (d <= 16) GPMI_KMAT_LAUNCH(16);
    else GPMI_KMAT_LAUNCH(0);
#undef GPMI_KMAT_LAUNCH
}

// ------------------------------------------------------------------------------------------------
// kgrad: reverse-mode weights of logpdf against the kernel hyper-parameters (the pullback a ChainRules rrule of
//   logpdf(fx, y) needs; the reference differentiates the same expression by AD — test/finite_gp_projection.jl:152-178):
//     ∂logpdf/∂θ = ½ Σ_ij (α_i α_j − C⁻¹_ij) ∂C_ij/∂θ
//   One 128×128 tile of the lower triangle per workgroup (strict lower counted twice).  Outputs (fp64, atomics):
//     g[0]        ∂/∂variance           (∂C_ij = κ_ij)
//     g[1..ns]    ∂/∂scale_p            (ScaleTransform: ns = 1, ∂r² = 2 r²/s;  ARD: ∂r² = 2 (u_ip − u_jp)²/v_p)
//   with dκ/dr²: SE −κ/2 · Matern12 −κ/(2r) · Matern32 −(3/2)e^{−√3 r} · Matern52 −(5/6)(1+√5 r)e^{−√5 r}.
//   x: pre-scaled inputs u = s∘x, dimension-major.  Cinv: row-major lower, holding −C⁻¹ as the triangular product leaves it (the kernels add it: until round 6 a
//   pass over the N×N block folded the sign first — 12.5 ms at C4, and the pass whose wrapped launch was the C4 gradient bug).  NSMAX bounds the ARD dimension.
// ------------------------------------------------------------------------------------------------
template <typename T> __device__ __forceinline__ void kappa_and_dr2(int kind, T d2, T& kap, T& dk) {
    if (kind == 0) {
        kap = exp_nonpos<T>(T(-0.5) * d2);
        dk = T(-0.5) * kap;
        return;
    }
    const T d = sqrt(d2);
    if (kind == 1) {
        kap = exp_nonpos<T>(-d);
        dk = d > T(0) ? -kap / (T(2) * d) : T(0);
        return;
    }
    if (kind == 2) {
        const T a = T(1.7320508075688772935) * d, e = exp_nonpos<T>(-a);
        kap = (T(1) + a) * e;
        dk = T(-1.5) * e;
        return;
    }
    const T a = T(2.2360679774997896964) * d, e = exp_nonpos<T>(-a);
    kap = (T(1) + a + T(5.0 / 3.0) * d2) * e;
    dk = T(-5.0 / 6.0) * (T(1) + a) * e;
}

// Both gradient kernels take any input dimension D: r² needs every dimension, the per-dimension sums only the 16 of the
// launch's chunk [p0, p0 + 16) — the dimension-major tiles are staged through LDS 16 dimensions at a time, r² of the thread's
// 64 (row, column) pairs accumulates in registers, and the chunk's own tiles stay in a second LDS image for the sums.
// The host launches once per chunk (one launch for D <= 16, and for scalar / no transforms whatever D is).
template <typename T>
__device__ __forceinline__ void grad_stage_d2(T (*xi)[128], T (*xj)[128], T (*xpi)[128], T (*xpj)[128], const T* __restrict__ x, long ldx,
                                              int d, int p0, int m0, int n0, T (&d2r)[32][2]) {
    constexpr int DC = 16;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
#pragma unroll
    for (int rr = 0; rr < 32; ++rr) d2r[rr][0] = d2r[rr][1] = T(0);
    for (int dc = 0; dc < d; dc += DC) {
        __syncthreads();
        for (int e = tid; e < DC * 128; e += 256) {
            const int dd = e >> 7, i = e & 127;
            const bool in = dc + dd < d;
            const T vi = in ? x[(long)(dc + dd) * ldx + m0 + i] : T(0), vj = in ? x[(long)(dc + dd) * ldx + n0 + i] : T(0);
            xi[dd][i] = vi;
            xj[dd][i] = vj;
            if (dc == p0) {
                xpi[dd][i] = vi;
                xpj[dd][i] = vj;
            }
        }
        __syncthreads();
        const int dcnt = (d - dc < DC) ? (d - dc) : DC;  // only the dimensions that exist (D = 3 used to pay for 16: the C4 pass took 29 ms for a 7 ms read)
#pragma unroll 4
        for (int rr = 0; rr < 32; ++rr) {
            const int row = w + 4 * rr;
#pragma unroll
            for (int cc = 0; cc < 2; ++cc) {
                const int col = 2 * lane + cc;
                T acc = d2r[rr][cc];
                for (int dd = 0; dd < dcnt; ++dd) {
                    const T t = xi[dd][row] - xj[dd][col];
                    acc = fma(t, t, acc);
                }
                d2r[rr][cc] = acc;
            }
        }
    }
    __syncthreads();
}

// g layout: [0] ∂/∂variance, [1] (noise sum, written by noise_grad_kernel), [2 + p] ∂/∂scale_p
template <typename T>
__global__ __launch_bounds__(256) void kgrad_kernel(const T* __restrict__ Cinv, long ld, const T* __restrict__ x, long ldx, int d,
                                                     int kind, T variance, int nscale, const double* __restrict__ scale,
                                                     const T* __restrict__ alpha, long n, double* __restrict__ g, int p0) {
    constexpr int DC = 16;
    const int m0 = blockIdx.y * 128, n0 = blockIdx.x * 128;
    if (n0 > m0) return;
    __shared__ T xi[DC][128];
    __shared__ T xj[DC][128];
    __shared__ T xpi[DC][128];
    __shared__ T xpj[DC][128];
    __shared__ double red[4][1 + DC];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    T d2r[32][2];
    grad_stage_d2<T>(xi, xj, xpi, xpj, x, ldx, d, p0, m0, n0, d2r);
    const int np_ = nscale > 1 ? min(DC, nscale - p0) : 0;  // ARD scales of this chunk
    double acc[1 + DC];
#pragma unroll
    for (int p = 0; p <= DC; ++p) acc[p] = 0.0;
#pragma unroll 2
    for (int rr = 0; rr < 32; ++rr) {
        const int row = w + 4 * rr;
        const long gi = m0 + row;
        if (gi >= n) continue;
        const T ai = alpha[gi];
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) {
            const int col = 2 * lane + cc;
            const long gj = n0 + col;
            if (gj > gi || gj >= n) continue;
            const T d2 = d2r[rr][cc];
            T kap, dk;
            kappa_and_dr2<T>(kind, d2, kap, dk);
            const double wgt = ((double)ai * (double)alpha[gj] + (double)Cinv[gi * ld + gj]) * (gi == gj ? 0.5 : 1.0);   // Cinv holds −C⁻¹
            acc[0] += wgt * (double)kap;
            const double wk = wgt * (double)variance * (double)dk * 2.0;
            if (nscale == 1) {
                acc[1] += wk * (double)d2;
            } else if (nscale > 1) {
#pragma unroll
                for (int p = 0; p < DC; ++p)
                    if (p < np_) {
                        const T t = xpi[p][row] - xpj[p][col];
                        acc[1 + p] += wk * (double)(t * t);
                    }
            }
        }
    }
#pragma unroll
    for (int p = 0; p <= DC; ++p) {
        double v = acc[p];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        if (lane == 0) red[w][p] = v;
    }
    __syncthreads();
    const int nout = nscale == 1 ? 1 : np_;
    if (tid <= nout) {
        const double v = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
        if (tid == 0) {
            if (p0 == 0) atomicAdd(g, v);  // the variance term once
        } else {
            atomicAdd(g + 2 + p0 + tid - 1, v / scale[p0 + tid - 1]);  // the 1/s (1/v_p) factor of ∂r²
        }
    }
}
// kgrad for D <= ND (4 / 8 / 16) without scratch memory (kgrad_kernel keeps its 64 r² values in a dynamically indexed array = 512 bytes of scratch per lane; the
// same reorganisation as vgrad_fast_kernel: column inputs and α_j in registers, row inputs as LDS broadcasts, r² and the differences on the fly, the weights as one
// 16-byte load per thread and row).  C2: 2.27 -> see NOTES_r6 §9.  Same sums, same g layout, Cinv = −C⁻¹.
template <typename T, int ND>
__global__ __launch_bounds__(256) void kgrad_fast_kernel(const T* __restrict__ Cinv, long ld, const T* __restrict__ x, long ldx, int d, int kind, T variance,
                                                          int nscale, const double* __restrict__ scale, const T* __restrict__ alpha, long n,
                                                          double* __restrict__ g) {
    typedef T t2_t __attribute__((ext_vector_type(2)));
    const int m0 = blockIdx.y * 128, n0 = blockIdx.x * 128;
    if (n0 > m0) return;
    __shared__ T xi[ND][128];
    __shared__ double red[4][1 + ND];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    for (int e = tid; e < ND * 128; e += 256) {
        const int dd = e >> 7, i = e & 127;
        xi[dd][i] = dd < d ? x[(long)dd * ldx + m0 + i] : T(0);
    }
    const long gj0 = n0 + 2 * lane;
    T xj[2][ND];
#pragma unroll
    for (int dd = 0; dd < ND; ++dd) {
        xj[0][dd] = dd < d ? x[(long)dd * ldx + gj0] : T(0);
        xj[1][dd] = dd < d ? x[(long)dd * ldx + gj0 + 1] : T(0);
    }
    const double aj[2] = {gj0 < n ? (double)alpha[gj0] : 0.0, gj0 + 1 < n ? (double)alpha[gj0 + 1] : 0.0};
    const int nps = nscale > 1 ? nscale : 0;
    double acc[1 + ND];
#pragma unroll
    for (int p = 0; p <= ND; ++p) acc[p] = 0.0;
    const double var = (double)variance;
    __syncthreads();
    for (int rr = 0; rr < 32; ++rr) {
        const int row = w + 4 * rr;
        const long gi = m0 + row;
        if (gi >= n) continue;  // wave-uniform
        const double ai = (double)alpha[gi];
        const t2_t c2 = *reinterpret_cast<const t2_t*>(Cinv + gi * ld + gj0);
        T xrow[ND];
#pragma unroll
        for (int dd = 0; dd < ND; ++dd) xrow[dd] = xi[dd][row];
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) {
            const long gj = gj0 + cc;
            if (gj > gi || gj >= n) continue;
            T tp[ND];
            T d2 = T(0);
#pragma unroll
            for (int dd = 0; dd < ND; ++dd) {
                tp[dd] = xrow[dd] - xj[cc][dd];
                d2 = fma(tp[dd], tp[dd], d2);
            }
            T kap, dk;
            kappa_and_dr2<T>(kind, d2, kap, dk);
            const double wgt = (ai * aj[cc] + (double)c2[cc]) * (gi == gj ? 0.5 : 1.0);
            acc[0] += wgt * (double)kap;
            const double wk = wgt * var * (double)dk * 2.0;
            if (nscale == 1) acc[1] += wk * (double)d2;
#pragma unroll
            for (int p = 0; p < ND; ++p)
                if (p < nps) {
                    const double t = (double)tp[p];
                    acc[1 + p] += wk * t * t;
                }
        }
    }
#pragma unroll
    for (int p = 0; p <= ND; ++p) {
        double v = acc[p];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        if (lane == 0) red[w][p] = v;
    }
    __syncthreads();
    const int nout = nscale == 1 ? 1 : nps;
    if (tid <= nout) {
        const double v = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
        if (tid == 0) atomicAdd(g, v);
        else atomicAdd(g + 2 + tid - 1, v / scale[tid - 1]);
    }
}
// kgradx: ∂logpdf/∂x_ip = 2 s_p σ² Σ_j (α_i α_j − C⁻¹_ij) dκ/dr²(r²_ij) (u_ip − u_jp)   (u = s∘x; both orders of the symmetric pair
//   folded in) — the input gradient a deep-kernel model back-propagates (examples/2-deep-kernel-learning/script.jl).  One
//   128×128 tile of the FULL square per workgroup (C⁻¹ is stored lower: the mirrored entry is read for tiles above the diagonal);
//   row sums by wave shuffles, one atomicAdd per (row, p).  gx: double [d][ldg] (dimension-major like x); dimensions [p0, p0+16).
template <typename T>
__global__ __launch_bounds__(256) void kgradx_kernel(const T* __restrict__ Cinv, long ld, const T* __restrict__ x, long ldx, int d,
                                                      int kind, T variance, int nscale, const double* __restrict__ scale,
                                                      const T* __restrict__ alpha, long n, double* __restrict__ gx, long ldg, int p0) {
    constexpr int DC = 16;
    const int m0 = blockIdx.y * 128, n0 = blockIdx.x * 128;
    __shared__ T xi[DC][128];
    __shared__ T xj[DC][128];
    __shared__ T xpi[DC][128];
    __shared__ T xpj[DC][128];
    __shared__ T aj[128];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    if (tid < 128) aj[tid] = (n0 + tid < n) ? alpha[n0 + tid] : T(0);
    T d2r[32][2];
    grad_stage_d2<T>(xi, xj, xpi, xpj, x, ldx, d, p0, m0, n0, d2r);
    const int np_ = min(DC, d - p0);
#pragma unroll 2
    for (int rr = 0; rr < 32; ++rr) {
        const int row = w + 4 * rr;
        const long gi = m0 + row;
        if (gi >= n) continue;  // wave-uniform
        const T ai = alpha[gi];
        double acc[DC];
#pragma unroll
        for (int p = 0; p < DC; ++p) acc[p] = 0.0;
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) {
            const int col = 2 * lane + cc;
            const long gj = n0 + col;
            if (gj >= n || gj == gi) continue;
            T kap, dk;
            kappa_and_dr2<T>(kind, d2r[rr][cc], kap, dk);
            const T ci = gi >= gj ? Cinv[gi * ld + gj] : Cinv[gj * ld + gi];
            const double wgt = ((double)ai * (double)aj[col] + (double)ci) * (double)dk;   // ci from −C⁻¹
#pragma unroll
            for (int p = 0; p < DC; ++p)
                if (p < np_) acc[p] += wgt * (double)(xpi[p][row] - xpj[p][col]);
        }
#pragma unroll
        for (int p = 0; p < DC; ++p) {
            if (p >= np_) break;
            double v = acc[p];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
            if (lane == 0) {
                const double sp = nscale == 0 ? 1.0 : (nscale == 1 ? scale[0] : scale[p0 + p]);
                atomicAdd(gx + (long)(p0 + p) * ldg + gi, 2.0 * sp * (double)variance * v);
            }
        }
    }
}
// out[i] = ½ (α_i² − C⁻¹_ii)   (∂logpdf/∂Σy_ii; Cinv holds −C⁻¹);  sum[0] += Σ_i out[i]   (one block of 256 threads per 256 rows)
template <typename T>
__global__ __launch_bounds__(256) void noise_grad_kernel(const T* __restrict__ Cinv, long ld, const T* __restrict__ alpha, long n,
                                                          T* __restrict__ out, double* __restrict__ sum) {
    __shared__ double red[4];
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    double v = 0;
    if (i < n) {
        const double a = (double)alpha[i];
        v = 0.5 * (a * a + (double)Cinv[i * ld + i]);   // Cinv holds −C⁻¹
        out[i] = (T)v;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(sum, red[0] + red[1] + red[2] + red[3]);
}

// ------------------------------------------------------------------------------------------------
// vgrad: reverse-mode pass of the SPARSE objective (elbo / approx_log_evidence, src/sparse_approximations.jl:248-254, :282-286) over one
//   rectangular kernel block K(rows, cols) — a chunk of observations against the pseudo-inputs, or K_zz itself.  Given the weights
//   W_ij = ∂L/∂K_ij the kernel accumulates, in fp64 and with κ, dκ/dr² recomputed from the inputs (K is never read):
//     g[0]       += Σ W_ij κ_ij                                         ∂/∂variance
//     g[2 + p]   += Σ W_ij σ² dκ_ij ∂r²_ij/∂scale_p                     as kgrad_kernel
//     gz[p][j]   += zfac · Σ_i W_ij σ² dκ_ij (−2 s_p)(u_ip − w_jp)      ∂/∂(column input j)   [ZG]
//     gx[p][i]   +=        Σ_j W_ij σ² dκ_ij (+2 s_p)(u_ip − w_jp)      ∂/∂(row input i)      [XG]
//   explicit_w = 1: W = Cm (the M×M weights of K_zz; the caller passes zfac = 2 for the row role of the symmetric pair).
//   explicit_w = 0: Cm = −T̃ with T̃ = (S K)(∂L/∂ψ) from the chunk's MFMA GEMM (S = Σy^-1/2 rows), and
//       W_ij = rs_i (2 T̃_ij + b_i ν_j)            (∂L/∂K_fz = Σy⁻¹ (2 K_fz G_ψ + δ νᵀ))
//       rowq[i] += Σ_j σ²κ_ij T̃_ij,  rowp[i] += Σ_j σ²κ_ij ν_j     (the two row sums the noise / y gradients need)
//   One 128×128 tile per workgroup, thread = 32 rows × 2 adjacent columns as in kgrad_kernel; NP = per-dimension accumulators kept in registers
//   (16 per launch chunk p0; the host takes this kernel for D > 16 only — vgrad_fast_kernel below serves D <= 16).  Row sums by wave shuffles, column sums through LDS across the 4 waves, then atomics.
// ------------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void grad_stage_d2_rect(T (*xi)[128], T (*xj)[128], T (*xpi)[128], T (*xpj)[128], const T* __restrict__ xr, long ldxr,
                                                   const T* __restrict__ xc, long ldxc, int d, int p0, int m0, int n0, T (&d2r)[32][2]) {
    constexpr int DC = 16;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
#pragma unroll
    for (int rr = 0; rr < 32; ++rr) d2r[rr][0] = d2r[rr][1] = T(0);
    for (int dc = 0; dc < d; dc += DC) {
        __syncthreads();
        for (int e = tid; e < DC * 128; e += 256) {
            const int dd = e >> 7, i = e & 127;
            const bool in = dc + dd < d;
            const T vi = in ? xr[(long)(dc + dd) * ldxr + m0 + i] : T(0), vj = in ? xc[(long)(dc + dd) * ldxc + n0 + i] : T(0);
            xi[dd][i] = vi;
            xj[dd][i] = vj;
            if (dc == p0) {
                xpi[dd][i] = vi;
                xpj[dd][i] = vj;
            }
        }
        __syncthreads();
        const int dcnt = (d - dc < DC) ? (d - dc) : DC;
#pragma unroll 4
        for (int rr = 0; rr < 32; ++rr) {
            const int row = w + 4 * rr;
#pragma unroll
            for (int cc = 0; cc < 2; ++cc) {
                const int col = 2 * lane + cc;
                T acc = d2r[rr][cc];
                for (int dd = 0; dd < dcnt; ++dd) {
                    const T t = xi[dd][row] - xj[dd][col];
                    acc = fma(t, t, acc);
                }
                d2r[rr][cc] = acc;
            }
        }
    }
    __syncthreads();
}

template <typename T, int NP, bool ZG, bool XG>
__global__ __launch_bounds__(256) void vgrad_kernel(const T* __restrict__ Cm, long ldc, int explicit_w, const T* __restrict__ xr, long ldxr,
                                                     const T* __restrict__ xc, long ldxc, int d, int kind, T variance, int nscale,
                                                     const double* __restrict__ scale, const T* __restrict__ rs, const T* __restrict__ bv,
                                                     const double* __restrict__ nu, long nr, long nc, double* __restrict__ g,
                                                     double* __restrict__ gz, long ldgz, double zfac, double* __restrict__ rowq,
                                                     double* __restrict__ rowp, double* __restrict__ gx, long ldgx, int p0) {
    constexpr int DC = 16;
    const int m0 = blockIdx.y * 128, n0 = blockIdx.x * 128;
    __shared__ T xi[DC][128];
    __shared__ T xj[DC][128];
    __shared__ T xpi[DC][128];
    __shared__ T xpj[DC][128];
    __shared__ double red[4][128];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    T d2r[32][2];
    grad_stage_d2_rect<T>(xi, xj, xpi, xpj, xr, ldxr, xc, ldxc, d, p0, m0, n0, d2r);
    const int npd = min(NP, d - p0);                          // input dimensions of this launch (z / x gradients)
    const int nps = nscale > 1 ? min(NP, nscale - p0) : 0;    // ARD scales of this launch
    const bool rows_out = !explicit_w && rowq && p0 == 0;
    double acch[1 + NP];
#pragma unroll
    for (int p = 0; p <= NP; ++p) acch[p] = 0.0;
    double accz[2][NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) accz[0][p] = accz[1][p] = 0.0;
    double nuj[2] = {0.0, 0.0};
    if (!explicit_w) {
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) nuj[cc] = (n0 + 2 * lane + cc < nc) ? nu[n0 + 2 * lane + cc] : 0.0;
    }
    const double var = (double)variance;
    for (int rr = 0; rr < 32; ++rr) {
        const int row = w + 4 * rr;
        const long gi = m0 + row;
        if (gi >= nr) continue;  // wave-uniform
        const double rsi = explicit_w ? 0.0 : (double)rs[gi], bi = explicit_w ? 0.0 : (double)bv[gi];
        double accx[NP];
#pragma unroll
        for (int p = 0; p < NP; ++p) accx[p] = 0.0;
        double aq = 0.0, ap = 0.0;
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) {
            const int col = 2 * lane + cc;
            const long gj = n0 + col;
            if (gj >= nc) continue;
            const T d2 = d2r[rr][cc];
            T kap, dk;
            kappa_and_dr2<T>(kind, d2, kap, dk);
            const double cij = (double)Cm[gi * ldc + gj];
            double W;
            if (explicit_w) {
                W = cij;
            } else {
                W = rsi * (-2.0 * cij + bi * nuj[cc]);
                aq -= var * (double)kap * cij;
                ap += var * (double)kap * nuj[cc];
            }
            if (p0 == 0) acch[0] += W * (double)kap;
            const double wk = W * var * (double)dk * 2.0;
            if (nscale == 1 && p0 == 0) acch[1] += wk * (double)d2;
#pragma unroll
            for (int p = 0; p < NP; ++p)
                if (p < npd) {
                    const double t = (double)(xpi[p][row] - xpj[p][col]);
                    if (p < nps) acch[1 + p] += wk * t * t;
                    if (ZG) accz[cc][p] -= wk * t;
                    if (XG) accx[p] += wk * t;
                }
        }
        if (rows_out) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                aq += __shfl_xor(aq, o, 64);
                ap += __shfl_xor(ap, o, 64);
            }
            if (lane == 0) {
                atomicAdd(rowq + gi, aq);
                atomicAdd(rowp + gi, ap);
            }
        }
        if (XG) {
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                if (p >= npd) break;
                double v = accx[p];
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
                if (lane == 0) {
                    const double sp = nscale == 0 ? 1.0 : (nscale == 1 ? scale[0] : scale[p0 + p]);
                    atomicAdd(gx + (long)(p0 + p) * ldgx + gi, sp * v);
                }
            }
        }
    }
    // hyper-parameter sums: waves -> LDS -> one atomic per workgroup and parameter
#pragma unroll
    for (int p = 0; p <= NP; ++p) {
        double v = acch[p];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        if (lane == 0) red[w][p] = v;
    }
    __syncthreads();
    {
        const int nout = nscale == 1 ? (p0 == 0 ? 1 : 0) : nps;
        if (tid <= nout) {
            const double v = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
            if (tid == 0) {
                if (p0 == 0) atomicAdd(g, v);
            } else {
                atomicAdd(g + 2 + p0 + tid - 1, v / scale[p0 + tid - 1]);
            }
        }
    }
    if (ZG) {
        for (int p = 0; p < npd; ++p) {
            __syncthreads();
            red[w][2 * lane] = accz[0][p];
            red[w][2 * lane + 1] = accz[1][p];
            __syncthreads();
            if (tid < 128 && n0 + tid < nc) {
                const double sp = nscale == 0 ? 1.0 : (nscale == 1 ? scale[0] : scale[p0 + p]);
                atomicAdd(gz + (long)(p0 + p) * ldgz + n0 + tid, zfac * sp * (red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid]));
            }
        }
    }
}

// vgrad for D <= ND (4 / 8 / 16): the same sums as vgrad_kernel, organised so that nothing lives in scratch memory — the thread's two column inputs sit in
// registers, the row inputs are LDS broadcasts, r² and the per-dimension differences t_p are formed on the fly (vgrad_kernel keeps its 64 r² values in a
// dynamically indexed array: 512 bytes per lane of scratch traffic, ≈ 1 GB per chunk at C5 — 1.1 ms per chunk against 0.13 ms for reading C once).
// The weights are read as one 16-byte load per thread and row.
template <typename T, int ND, bool XG>
__global__ __launch_bounds__(256) void vgrad_fast_kernel(const T* __restrict__ Cm, long ldc, int explicit_w, const T* __restrict__ xr, long ldxr,
                                                          const T* __restrict__ xc, long ldxc, int d, int kind, T variance, int nscale,
                                                          const double* __restrict__ scale, const T* __restrict__ rs, const T* __restrict__ bv,
                                                          const double* __restrict__ nu, long nr, long nc, double* __restrict__ g,
                                                          double* __restrict__ gz, long ldgz, double zfac, double* __restrict__ rowq,
                                                          double* __restrict__ rowp, double* __restrict__ gx, long ldgx) {
    typedef T t2_t __attribute__((ext_vector_type(2)));
    const int m0 = blockIdx.y * 128, n0 = blockIdx.x * 128;
    __shared__ T xi[ND][128];
    __shared__ double red[4][128];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    for (int e = tid; e < ND * 128; e += 256) {
        const int dd = e >> 7, i = e & 127;
        xi[dd][i] = (dd < d && m0 + i < nr) ? xr[(long)dd * ldxr + m0 + i] : T(0);
    }
    T xj[2][ND];
    const long gj0 = n0 + 2 * lane;
#pragma unroll
    for (int dd = 0; dd < ND; ++dd) {
        xj[0][dd] = (dd < d && gj0 < nc) ? xc[(long)dd * ldxc + gj0] : T(0);
        xj[1][dd] = (dd < d && gj0 + 1 < nc) ? xc[(long)dd * ldxc + gj0 + 1] : T(0);
    }
    const int nps = nscale > 1 ? nscale : 0;   // ARD: one scale per dimension
    const bool rows_out = !explicit_w && rowq;
    double acch[1 + ND];
#pragma unroll
    for (int p = 0; p <= ND; ++p) acch[p] = 0.0;
    double accz[2][ND];
#pragma unroll
    for (int p = 0; p < ND; ++p) accz[0][p] = accz[1][p] = 0.0;
    double nuj[2] = {0.0, 0.0};
    if (!explicit_w) {
        nuj[0] = gj0 < nc ? nu[gj0] : 0.0;
        nuj[1] = gj0 + 1 < nc ? nu[gj0 + 1] : 0.0;
    }
    const double var = (double)variance;
    __syncthreads();
    for (int rr = 0; rr < 32; ++rr) {
        const int row = w + 4 * rr;
        const long gi = m0 + row;
        if (gi >= nr) continue;  // wave-uniform
        const double rsi = explicit_w ? 0.0 : (double)rs[gi], bi = explicit_w ? 0.0 : (double)bv[gi];
        const t2_t c2 = *reinterpret_cast<const t2_t*>(Cm + gi * ldc + gj0);   // columns beyond nc are padding of the same allocation
        T xrow[ND];
#pragma unroll
        for (int dd = 0; dd < ND; ++dd) xrow[dd] = xi[dd][row];
        double accx[ND];
#pragma unroll
        for (int p = 0; p < ND; ++p) accx[p] = 0.0;
        double aq = 0.0, ap = 0.0;
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) {
            if (gj0 + cc >= nc) continue;
            T tp[ND];
            T d2 = T(0);
#pragma unroll
            for (int dd = 0; dd < ND; ++dd) {
                tp[dd] = xrow[dd] - xj[cc][dd];
                d2 = fma(tp[dd], tp[dd], d2);
            }
            T kap, dk;
            kappa_and_dr2<T>(kind, d2, kap, dk);
            const double cij = (double)c2[cc];
            double W;
            if (explicit_w) {
                W = cij;
            } else {
                W = rsi * (-2.0 * cij + bi * nuj[cc]);
                aq -= var * (double)kap * cij;
                ap += var * (double)kap * nuj[cc];
            }
            acch[0] += W * (double)kap;
            const double wk = W * var * (double)dk * 2.0;
            if (nscale == 1) acch[1] += wk * (double)d2;
#pragma unroll
            for (int p = 0; p < ND; ++p) {
                const double t = (double)tp[p];
                if (p < nps) acch[1 + p] += wk * t * t;
                accz[cc][p] -= wk * t;
                if (XG) accx[p] += wk * t;
            }
        }
        if (rows_out) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                aq += __shfl_xor(aq, o, 64);
                ap += __shfl_xor(ap, o, 64);
            }
            if (lane == 0) {
                atomicAdd(rowq + gi, aq);
                atomicAdd(rowp + gi, ap);
            }
        }
        if (XG) {
#pragma unroll
            for (int p = 0; p < ND; ++p) {
                if (p >= d) break;
                double v = accx[p];
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
                if (lane == 0) {
                    const double sp = nscale == 0 ? 1.0 : (nscale == 1 ? scale[0] : scale[p]);
                    atomicAdd(gx + (long)p * ldgx + gi, sp * v);
                }
            }
        }
    }
#pragma unroll
    for (int p = 0; p <= ND; ++p) {
        double v = acch[p];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        if (lane == 0) red[w][p] = v;
    }
    __syncthreads();
    {
        const int nout = nscale == 1 ? 1 : nps;
        if (tid <= nout) {
            const double v = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
            if (tid == 0) atomicAdd(g, v);
            else atomicAdd(g + 2 + tid - 1, v / scale[tid - 1]);
        }
    }
#pragma unroll
    for (int p = 0; p < ND; ++p) {
        if (p >= d) break;
        __syncthreads();
        red[w][2 * lane] = accz[0][p];
        red[w][2 * lane + 1] = accz[1][p];
        __syncthreads();
        if (tid < 128 && n0 + tid < nc) {
            const double sp = nscale == 0 ? 1.0 : (nscale == 1 ? scale[0] : scale[p]);
            atomicAdd(gz + (long)p * ldgz + n0 + tid, zfac * sp * (red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid]));
        }
    }
}

// Per observation, from the two row sums of the streamed pass (rq_i = Σ_j σ²κ_ij T̃_ij, rp_i = Σ_j σ²κ_ij α_j = the posterior mean's K_fz α), r_i = σ_i⁻¹ and
// b_i = r_i δ_i:   ∂L/∂σ_i² = −½ r² + ½ b² r² − rq r³ − rp b r³ [+ ½ σ_k² r⁴ for VFE]      ∂L/∂y_i = −(b r − rp r²)
// written in the handle's dtype; sums[0] += Σ_i ∂L/∂σ_i², sums[1] += Σ_i r_i² (fp64, one atomic pair per block).  Keeps the N-long finishing loop and two of the
// four N-long downloads off the host (N = 2·10⁶: ≈ 20 ms of a 30 ms gradient call).
template <typename T>
__global__ __launch_bounds__(256) void vgrad_finish_kernel(const T* __restrict__ rs, const T* __restrict__ bv, const double* __restrict__ rq,
                                                            const double* __restrict__ rp, long n, double variance, int vfe, T* __restrict__ dn,
                                                            T* __restrict__ dy, double* __restrict__ sums) {
    __shared__ double red[2][4];
    double s0 = 0.0, s1 = 0.0;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const double r = (double)rs[i], b = (double)bv[i], r2 = r * r, r3 = r2 * r;
        const double v = -0.5 * r2 + 0.5 * b * b * r2 - rq[i] * r3 - rp[i] * b * r3 + (vfe ? 0.5 * variance * r2 * r2 : 0.0);
        dn[i] = (T)v;
        dy[i] = (T)(-(b * r - rp[i] * r2));
        s0 += v;
        s1 += r2;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        s0 += __shfl_xor(s0, o, 64);
        s1 += __shfl_xor(s1, o, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        red[0][threadIdx.x >> 6] = s0;
        red[1][threadIdx.x >> 6] = s1;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(sums, red[0][0] + red[0][1] + red[0][2] + red[0][3]);
        atomicAdd(sums + 1, red[1][0] + red[1][1] + red[1][2] + red[1][3]);
    }
}

// out[i][j] = sa·a[hi][lo] + sb·b[hi][lo] + dg·[i == j] + so·v_i v_j   (hi = max(i, j), lo = min(i, j); a, b lower-stored; b, v may be NULL): the symmetric
// M×M combinations of the sparse gradient (I − A⁻¹ − B Bᵀ from two lower triangles; ½ E − ½ ααᵀ).  grid (ceil(n/256), n)
__global__ __launch_bounds__(256) void sym_combine_kernel(double* __restrict__ out, long ldo, long n, const double* __restrict__ a, long lda, double sa,
                                                          const double* __restrict__ b, long ldb, double sb, double dg,
                                                          const double* __restrict__ v, double so) {
    const long j = (long)blockIdx.x * 256 + threadIdx.x, i = blockIdx.y;
    if (j >= n) return;
    const long hi = i > j ? i : j, lo = i > j ? j : i;
    double r = sa * a[hi * lda + lo];
    if (b) r += sb * b[hi * ldb + lo];
    if (i == j) r += dg;
    if (v) r += so * v[i] * v[j];
    out[i * ldo + j] = r;
}

// ------------------------------------------------------------------------------------------------
// panel64: ONE launch per 64-column leaf of the panel factorisation — Cholesky of the 64×64 diagonal tile AND
//   X ← X L⁻ᵀ for every row below it (one launch instead of a tile factorisation + a triangular solve and their dependent-launch gap).
//   info (device int32): first failing global column (1-based) if a pivot is not > 0 (LAPACK dpotrf info), untouched otherwise;
//   logdet_acc += Σ_c log L_cc over columns with global index < n_valid.
//   grid = ceil(mrows / 128) workgroups (at least 1) of 4 waves; workgroup b owns rows [64 + 128 b, +128) under the
//   tile.  EVERY workgroup factors the diagonal tile itself, redundantly and bit-identically, in LDS — the serial
//   chain costs latency, not throughput, so replicating it is free and removes the global hand-off.  The input tile
//   must stay intact until every workgroup has read it (late workgroups start after early ones retire), so each
//   workgroup ticks `ticket` after its load and the LAST one to load is the one that stores the factor, Σ log L_ii
//   and the LAPACK info (and re-arms the ticket for the next launch on the stream).
//   The tile is processed in 16-column blocks: wave 0 factors the 16×16 diagonal block in registers (lane = row, all
//   cross-lane traffic by v_readlane) and, in the same instruction stream, builds its inverse (lane = column of
//   inv(L16); forward substitution fed by the same readlanes — it fills the latency holes of the pivot chain).  Every
//   other operation is a 16×16×16 product P·Qᵀ on the fp64/fp32 MFMA with both operands read row-wise from LDS:
//     block TRSM   B_ij ← B_ij · Inv_jᵀ          (tile rows below the block and the workgroup's X rows)
//     block update B_ik ← B_ik − B_ij · B_kjᵀ     (k > j)
//   Left-looking entry (kpre > 0): the kpre 64-column tiles immediately LEFT of this leaf (same rows, already final) are
//   applied first —  [D; X] ← [D; X] − [L_k; X_k] · L_kᵀ,  k = 0..kpre−1  — as 16×16×16 MFMA products accumulated in
//   registers, operands staged through the same two LDS buffers.  That replaces the K = 64 / 128 trailing GEMMs between
//   the leaves of a 256-column group (three launches of a latency-bound kernel per group) by ≈3 µs of in-leaf work each.
// ------------------------------------------------------------------------------------------------
template <typename T, int XR = 128>  // XR: rows of X per workgroup (128; 64 keeps the LDS footprint at 75 KB)
__global__ __launch_bounds__(256) void panel64_kernel(T* __restrict__ A, long lda, int mrows, int* __restrict__ info, int col0,
                                                       int n_valid, double* __restrict__ logdet_acc, int* __restrict__ ticket,
                                                       int kpre) {
    using TR = Tr<T>;
    using chunk_t = typename TR::chunk_t;
    using acc_t = typename TR::acc_t;
    constexpr int VEC = TR::VEC;
    // row pitch: the MFMA operand reads (lane (li, lg) -> element li·LD + 4m + lg) hit distinct banks for a half-wave when the pitch is
    // ≡ 2 (f64: 8-B elements) / ≡ 4 (f32) modulo the 64 dword banks' period; 65 was two-way conflicted (SQ_LDS_BANK_CONFLICT = 16 % of
    // SQ_LDS_IDX_ACTIVE, profiles/r2/pmc_sq_summary.json).  An even pitch also keeps the 16-B staging chunks aligned.
    constexpr int LD = sizeof(T) == 8 ? 66 : 68, LI = 17;
    __shared__ __attribute__((aligned(16))) T Ds[64 * LD];
    __shared__ __attribute__((aligned(16))) T Xs[XR * LD];
    __shared__ T Inv[4][16 * LI];
    constexpr int LI2 = sizeof(T) == 8 ? 18 : 20;  // row pitch of Lr: a multiple of 16 B
    __shared__ __attribute__((aligned(16))) T Lr[16 * LI2];  // rows of the 16×16 block being factored
    __shared__ int writer_s;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
#ifdef GPMI_PANEL_STAMPS
    long stamps[16];
    int nst = 0;
#define PSTAMP() do { if (nst < 16) stamps[nst++] = (long)__builtin_readcyclecounter(); } while (0)
#else
#define PSTAMP() do { } while (0)
#endif
    PSTAMP();
    const int li = lane & 15, lg = lane >> 4;
    int xrows = mrows - (int)blockIdx.x * XR;
    xrows = xrows < 0 ? 0 : (xrows > XR ? XR : xrows);
    T* const Xg = A + (long)(64 + (long)blockIdx.x * XR) * lda;

    // ---- left-looking pre-update: wave w owns column block w of every 16-row tile (X: XR/16 tiles, D: 4 tiles)
    constexpr int NXT = XR / 16;
    acc_t dxp[NXT], ddp[4];
    if (kpre > 0) {
        constexpr int CPR = 64 / VEC;
        constexpr int ND = 64 * CPR / 256, NX = XR * CPR / 256;
#pragma unroll
        for (int i = 0; i < NXT; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) dxp[i][r] = T(0);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) ddp[i][r] = T(0);
        const int nxtp = (xrows + 15) >> 4;
        for (int k = 0; k < kpre; ++k) {
            const long coff = -64L * (kpre - k);
            chunk_t dv[ND], xv[NX];
#pragma unroll
            for (int i = 0; i < ND; ++i) {
                const int e = tid + 256 * i, row = e / CPR, cc = e % CPR;
                dv[i] = *reinterpret_cast<const chunk_t*>(A + (long)row * lda + coff + cc * VEC);
            }
#pragma unroll
            for (int i = 0; i < NX; ++i) {
                const int e = tid + 256 * i, row = e / CPR, cc = e % CPR;
#pragma unroll
                for (int q = 0; q < VEC; ++q) xv[i][q] = T(0);
                if (row < xrows) xv[i] = *reinterpret_cast<const chunk_t*>(Xg + (long)row * lda + coff + cc * VEC);
            }
            if (k > 0) __syncthreads();  // the previous tile's fragments have been read
#pragma unroll
            for (int i = 0; i < ND; ++i) {
                const int e = tid + 256 * i, row = e / CPR, cc = e % CPR;
#pragma unroll
                for (int q = 0; q < VEC; ++q) Ds[row * LD + cc * VEC + q] = dv[i][q];
            }
#pragma unroll
            for (int i = 0; i < NX; ++i) {
                const int e = tid + 256 * i, row = e / CPR, cc = e % CPR;
#pragma unroll
                for (int q = 0; q < VEC; ++q) Xs[row * LD + cc * VEC + q] = xv[i][q];
            }
            __syncthreads();
            T qf[16];  // Q fragments of block row w of L_k: shared by every product of this wave
#pragma unroll
            for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                for (int m = 0; m < 4; ++m) qf[4 * kb + m] = Ds[(16 * w + li) * LD + 16 * kb + 4 * m + lg];
#pragma unroll
            for (int i = 0; i < NXT; ++i) {
                if (i < nxtp) {
#pragma unroll
                    for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                        for (int m = 0; m < 4; ++m)
                            dxp[i] = TR::mfma(-Xs[(16 * i + li) * LD + 16 * kb + 4 * m + lg], qf[4 * kb + m], dxp[i]);
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (i >= w) {  // lower block triangle of D only
#pragma unroll
                    for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                        for (int m = 0; m < 4; ++m)
                            ddp[i] = TR::mfma(-Ds[(16 * i + li) * LD + 16 * kb + 4 * m + lg], qf[4 * kb + m], ddp[i]);
                }
            }
        }
        __syncthreads();  // LDS buffers are free again
    }

    {  // all global loads of the tile and of the X slab are issued before the first LDS store (one memory round trip)
        constexpr int CPR = 64 / VEC;              // 16-B chunks per 64-column row
        constexpr int ND = 64 * CPR / 256, NX = XR * CPR / 256;
        chunk_t dv[ND], xv[NX];
#pragma unroll
        for (int i = 0; i < ND; ++i) {
            const int e = tid + 256 * i, row = e / CPR, cc = e % CPR;
            dv[i] = *reinterpret_cast<const chunk_t*>(A + (long)row * lda + cc * VEC);
        }
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            const int e = tid + 256 * i, row = e / CPR, cc = e % CPR;
#pragma unroll
            for (int q = 0; q < VEC; ++q) xv[i][q] = T(0);
            if (row < xrows) xv[i] = *reinterpret_cast<const chunk_t*>(Xg + (long)row * lda + cc * VEC);
        }
#pragma unroll
        for (int i = 0; i < ND; ++i) {
            const int e = tid + 256 * i, row = e / CPR, cc = e % CPR;
#pragma unroll
            for (int q = 0; q < VEC; ++q) Ds[row * LD + cc * VEC + q] = dv[i][q];
        }
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            const int e = tid + 256 * i, row = e / CPR, cc = e % CPR;
#pragma unroll
            for (int q = 0; q < VEC; ++q) Xs[row * LD + cc * VEC + q] = xv[i][q];
        }
    }
    __syncthreads();  // every load of the input tile by this workgroup has completed (values are in LDS)
    PSTAMP();
    if (tid == 0) writer_s = (atomicAdd(ticket, 1) == (int)gridDim.x - 1);
    if (kpre > 0) {  // fold the pre-update in (each 16×16 block is owned by exactly one wave)
#pragma unroll
        for (int i = 0; i < NXT; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) Xs[(16 * i + TR::crow(lane, r)) * LD + 16 * w + li] += dxp[i][r];
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (i >= w) {
#pragma unroll
                for (int r = 0; r < 4; ++r) Ds[(16 * i + TR::crow(lane, r)) * LD + 16 * w + li] += ddp[i][r];
            }
        __syncthreads();
    }

    int bad = 0;
    T mydiag[4] = {T(1), T(1), T(1), T(1)};  // L_cc of column 16j + li (lanes < 16 of wave 0); log() is taken once, by the writer
    const int nxt = xrows >> 4;  // 16-row tiles of X owned by this workgroup
    // P·Qᵀ fragment: lane supplies P[li][4m + lg] and Q[li][4m + lg]
    auto mma16 = [&](acc_t d, const T* P, int ldp, const T* Q, int ldq, bool neg) -> acc_t {
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            T a = P[li * ldp + 4 * m + lg];
            const T b = Q[li * ldq + 4 * m + lg];
            if (neg) a = -a;
            d = TR::mfma(a, b, d);
        }
        return d;
    };
    // wave 0 only: factor the 16×16 diagonal block j (L16 into Ds, inv(L16) into Inv[j])
    auto factor16 = [&](int j) {
            // Four-column blocks.  The serial chain is pivot -> rsqrt -> scale -> next pivot; everything a later pivot of the SAME
            // block (and the first pivot of the next block) needs travels by v_readlane, so no LDS round trip sits on that chain.
            // After a block the four finished entries of every row are published ONCE (Lr[row][c0..c0+3], 32 B per lane) and the
            // columns further right are updated from broadcast ds_read_b128s (4 multipliers per row); the rows c0..c0+3 of L are
            // then complete in LDS and the matching rows of inv(L16) follow (lane = column of the inverse).  The one-column variant
            // paid a store -> wait -> load -> wait per column: ≈10 000 cycles per 16×16 block (tools/panel_stamps.hip).
            T a[16], x[16], rin[16];
#pragma unroll
            for (int c = 0; c < 16; ++c) a[c] = Ds[(16 * j + li) * LD + 16 * j + c];
#pragma unroll
            for (int b4 = 0; b4 < 4; ++b4) {
                constexpr int NCH = 4 / VEC;
                const int c0 = 4 * b4;
                T v[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int c = c0 + k;
                    const T piv = lane_bcast<T>(a[c], c);
                    if (!(piv > T(0)) && bad == 0) bad = 16 * j + c + 1;
                    const T ri = fast_rsqrt<T>(piv);
                    const T dd = piv * ri;
                    rin[c] = ri;
                    v[k] = (li == c) ? dd : a[c] * ri;
                    a[c] = v[k];
                    if (li == c) mydiag[j] = dd;
#pragma unroll
                    for (int t = c + 1; t <= c0 + 4; ++t)
                        if (t < 16) a[t] = fma(-v[k], lane_bcast<T>(v[k], t < 16 ? t : 15), a[t]);
                }
                {  // publish L[li][c0..c0+3]
                    chunk_t* dst = reinterpret_cast<chunk_t*>(&Lr[li * LI2 + c0]);
#pragma unroll
                    for (int q = 0; q < NCH; ++q) {
                        chunk_t cv;
#pragma unroll
                        for (int e = 0; e < VEC; ++e) cv[e] = v[q * VEC + e];
                        dst[q] = cv;
                    }
                }
#ifndef GPMI_EXP_NOUPD
#pragma unroll
                for (int t = c0 + 5; t < 16; ++t) {  // a[t] −= Σ_k L[li][c0+k] · L[t][c0+k]
                    const chunk_t* src = reinterpret_cast<const chunk_t*>(&Lr[t * LI2 + c0]);
                    T s0 = a[t], s1 = T(0);
#pragma unroll
                    for (int q = 0; q < NCH; ++q) {
                        const chunk_t m = src[q];
#pragma unroll
                        for (int e = 0; e < VEC; ++e) {
                            if ((q * VEC + e) & 1) s1 = fma(-v[q * VEC + e], m[e], s1);
                            else s0 = fma(-v[q * VEC + e], m[e], s0);
                        }
                    }
                    a[t] = s0 + s1;
                }
#endif
#ifndef GPMI_EXP_NOINV
#pragma unroll
                for (int k = 0; k < 4; ++k) {  // row c of inv(L16): x[c] = (δ − Σ_{m<c} L[c][m] x[m]) / L[c][c]
                    const int c = c0 + k;
                    T s0 = (li == c) ? T(1) : T(0), s1 = T(0);
                    const chunk_t* src = reinterpret_cast<const chunk_t*>(&Lr[c * LI2]);
#pragma unroll
                    for (int q = 0; q * VEC < c; ++q) {
                        const chunk_t m = src[q];
#pragma unroll
                        for (int e = 0; e < VEC; ++e) {
                            const int kk = q * VEC + e;
                            if (kk < c) {
                                if (kk & 1) s1 = fma(-m[e], x[kk], s1);
                                else s0 = fma(-m[e], x[kk], s0);
                            }
                        }
                    }
                    x[c] = (s0 + s1) * rin[c];
                }
#else
#pragma unroll
                for (int k = 0; k < 4; ++k) x[c0 + k] = rin[c0 + k];
#endif
            }
            if (lane < 16) {
#pragma unroll
                for (int c = 0; c < 16; ++c) {
                    if (c <= li) Ds[(16 * j + li) * LD + 16 * j + c] = a[c];
                    Inv[j][c * LI + li] = x[c];
                }
            }
    };
    // one 16×16 task each: block TRSM  P ← P · Inv_jᵀ   and block update  C ← C − P · Qᵀ
    auto trsm_blk = [&](T* P, int j) {
        acc_t d;
#pragma unroll
        for (int r = 0; r < 4; ++r) d[r] = T(0);
        d = mma16(d, P, LD, &Inv[j][0], LI, false);
#pragma unroll
        for (int r = 0; r < 4; ++r) P[TR::crow(lane, r) * LD + li] = d[r];
    };
    auto upd_blk = [&](T* Cb, const T* P, const T* Q) {
        acc_t d;
#pragma unroll
        for (int r = 0; r < 4; ++r) d[r] = Cb[TR::crow(lane, r) * LD + li];
        d = mma16(d, P, LD, Q, LD, true);
#pragma unroll
        for (int r = 0; r < 4; ++r) Cb[TR::crow(lane, r) * LD + li] = d[r];
    };
    // Schedule per 16-column block j (three barriers, as before), with the X-row products of block j running in the shadow of
    // the factorisation of block j+1 — the 16×16 factorisation is a serial chain on ONE wave (≈2.7 µs) and used to idle the
    // other three:
    //   1. tile TRSM      D(k, j) ← D(k, j) Inv_jᵀ,  k > j              (≤ 3 tasks, one per wave)
    //   2. tile updates   D(i, k) −= D(i, j) D(k, j)ᵀ,  j < k ≤ i        (≤ 6 tasks)
    //   3. wave 0: factor block j+1   ∥   waves 1–3: for each of their X row tiles  X(·, j) ← X(·, j) Inv_jᵀ  and then
    //      X(·, k) −= X(·, j) D(k, j)ᵀ, k > j  (row-tile local: no barrier between the two)
    if (w == 0) factor16(0);
    __syncthreads();
    PSTAMP();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (j < 3) {
            if (w < 3 - j) trsm_blk(&Ds[(16 * (j + 1 + w)) * LD + 16 * j], j);
            __syncthreads();
            const int nk = 3 - j, npair = nk * (nk + 1) / 2;
            for (int q = w; q < npair; q += 4) {
                int i = 0, rem = q;  // enumerate (i, k): i = j+1..3, k = j+1..i
                while (rem > i) {
                    rem -= i + 1;
                    ++i;
                }
                const int bi = j + 1 + i, bk = j + 1 + rem;
                upd_blk(&Ds[(16 * bi) * LD + 16 * bk], &Ds[(16 * bi) * LD + 16 * j], &Ds[(16 * bk) * LD + 16 * j]);
            }
            __syncthreads();
            PSTAMP();
        }
        if (j < 3 && w == 0) {
            factor16(j + 1);
        } else {
            const int nw = (j < 3) ? 3 : 4, w0 = (j < 3) ? w - 1 : w;
            for (int rt = w0; rt < nxt; rt += nw) {
                T* const Xr = &Xs[(16 * rt) * LD];
                trsm_blk(Xr + 16 * j, j);
#pragma unroll
                for (int k = j + 1; k < 4; ++k) upd_blk(Xr + 16 * k, Xr + 16 * j, &Ds[(16 * k) * LD + 16 * j]);
            }
        }
        __syncthreads();
        PSTAMP();
    }

    PSTAMP();
    // write back
#pragma unroll 8
    for (int e = tid; e < xrows * 64; e += 256) {
        const int row = e >> 6, c = e & 63;
        Xg[(long)row * lda + c] = Xs[row * LD + c];
    }
    if (writer_s) {  // (the barriers of the block loop made writer_s visible)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int e = tid + 256 * i, row = e >> 6, c = e & 63;
            if (c <= row) A[(long)row * lda + c] = Ds[row * LD + c];
        }
        if (w == 0) {
            double logd = 0.0;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (lane < 16 && col0 + 16 * j + lane < n_valid) logd += log((double)mydiag[j]);
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) logd += __shfl_xor(logd, o, 64);
            if (lane == 0) {
                if (logdet_acc) atomicAdd(logdet_acc, logd);
                if (bad && info && *info == 0) *info = col0 + bad;
                *ticket = 0;
            }
        }
    }
#ifdef GPMI_PANEL_STAMPS
    PSTAMP();
    if (blockIdx.x == 0 && tid == 0 && logdet_acc) {
        long* dst = reinterpret_cast<long*>(logdet_acc) + 8;
        for (int i = 0; i < 16; ++i) dst[i] = i < nst ? stamps[i] - stamps[0] : 0;
    }
#endif
#undef PSTAMP
}

// ------------------------------------------------------------------------------------------------
// trsm64_mfma: X[M×64] ← X · L⁻ᵀ against a GIVEN 64×64 lower tile, on the matrix pipe (the leaf of every blocked TRSM:
//   predictive variances, sequential updates, triangular inverses, the multi-GPU rows-below solve).  One workgroup per
//   128 rows.  The four 16×16 diagonal blocks of L are inverted concurrently, one per wave (lane = column of the
//   inverse, forward substitution against LDS broadcasts); then the same block TRSM / block update products P·Qᵀ as
//   in panel64_kernel.  M multiple of 64.
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void trsm64_mfma_kernel(T* __restrict__ X, long ldx, int M, const T* __restrict__ L, long ldl, long xstride = 0,
                                                           long lstride = 0) {  // blockIdx.y = b: an independent solve on X + b·xstride against L + b·lstride
    X += (long)blockIdx.y * xstride;
    L += (long)blockIdx.y * lstride;
    using TR = Tr<T>;
    using chunk_t = typename TR::chunk_t;
    using acc_t = typename TR::acc_t;
    constexpr int VEC = TR::VEC;
    constexpr int LD = 65, LI = 17;
    __shared__ T Ds[64 * LD];
    __shared__ T Xs[128 * LD];
    __shared__ T Inv[4][16 * LI];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int li = lane & 15, lg = lane >> 4;
    int xrows = M - (int)blockIdx.x * 128;
    xrows = xrows > 128 ? 128 : xrows;
    T* const Xg = X + (long)blockIdx.x * 128 * ldx;
    {  // all global loads first, then the LDS stores (one memory round trip)
        constexpr int CPR = 64 / VEC;
        constexpr int ND = 64 * CPR / 256, NX = 128 * CPR / 256;
        chunk_t dv[ND], xv[NX];
#pragma unroll
        for (int i = 0; i < ND; ++i) {
            const int e = tid + 256 * i, row = e / CPR, cc = e % CPR;
            dv[i] = *reinterpret_cast<const chunk_t*>(L + (long)row * ldl + cc * VEC);
        }
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            const int e = tid + 256 * i, row = e / CPR, cc = e % CPR;
#pragma unroll
            for (int q = 0; q < VEC; ++q) xv[i][q] = T(0);
            if (row < xrows) xv[i] = *reinterpret_cast<const chunk_t*>(Xg + (long)row * ldx + cc * VEC);
        }
#pragma unroll
        for (int i = 0; i < ND; ++i) {
            const int e = tid + 256 * i, row = e / CPR, cc = e % CPR;
#pragma unroll
            for (int q = 0; q < VEC; ++q) Ds[row * LD + cc * VEC + q] = dv[i][q];
        }
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            const int e = tid + 256 * i, row = e / CPR, cc = e % CPR;
#pragma unroll
            for (int q = 0; q < VEC; ++q) Xs[row * LD + cc * VEC + q] = xv[i][q];
        }
    }
    __syncthreads();
    {  // wave w inverts diagonal block w: lane c (< 16) owns column c of inv(L16); x[r] = Inv[r][c]
        const T* Lb = &Ds[(16 * w) * LD + 16 * w];
        T x[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            T sacc = (li == r) ? T(1) : T(0);
#pragma unroll
            for (int k = 0; k < r; ++k) sacc = fma(-Lb[r * LD + k], x[k], sacc);
            x[r] = sacc / Lb[r * LD + r];
        }
        if (lane < 16) {
#pragma unroll
            for (int r = 0; r < 16; ++r) Inv[w][r * LI + li] = x[r];
        }
    }
    __syncthreads();
    const int nxt = xrows >> 4;
    auto mma16 = [&](acc_t d, const T* P, int ldp, const T* Q, int ldq, bool neg) -> acc_t {
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            T a = P[li * ldp + 4 * m + lg];
            const T b = Q[li * ldq + 4 * m + lg];
            if (neg) a = -a;
            d = TR::mfma(a, b, d);
        }
        return d;
    };
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        for (int q = w; q < nxt; q += 4) {
            T* P = &Xs[(16 * q) * LD + 16 * j];
            acc_t d;
#pragma unroll
            for (int r = 0; r < 4; ++r) d[r] = T(0);
            d = mma16(d, P, LD, &Inv[j][0], LI, false);
#pragma unroll
            for (int r = 0; r < 4; ++r) P[TR::crow(lane, r) * LD + li] = d[r];
        }
        __syncthreads();
        if (j < 3) {
            const int nk = 3 - j;
            for (int q = w; q < nxt * nk; q += 4) {
                const int rt = q / nk, bk = j + 1 + q % nk;
                T* Cb = &Xs[(16 * rt) * LD + 16 * bk];
                acc_t d;
#pragma unroll
                for (int r = 0; r < 4; ++r) d[r] = Cb[TR::crow(lane, r) * LD + li];
                d = mma16(d, &Xs[(16 * rt) * LD + 16 * j], LD, &Ds[(16 * bk) * LD + 16 * j], LD, true);
#pragma unroll
                for (int r = 0; r < 4; ++r) Cb[TR::crow(lane, r) * LD + li] = d[r];
            }
            __syncthreads();
        }
    }
#pragma unroll 8
    for (int e = tid; e < xrows * 64; e += 256) {
        const int row = e >> 6, c = e & 63;
        Xg[(long)row * ldx + c] = Xs[row * LD + c];
    }
}

// ------------------------------------------------------------------------------------------------
// trtri_64 (batched): W_j = I − inv(L_jj) for the 64×64 diagonal tiles j of a lower factor (one wave per tile).
//   With it the 64-wide triangular solve becomes an in-place MFMA update  X_j ← X_j − X_j W_jᵀ = X_j L_jj⁻ᵀ
//   — what the vector solves' 64-wide steps use (trsv_diag*).  Lane i runs the forward substitution on row e_i, i.e. ends up
//   holding column i of inv(L).
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(64) void trtri_64_kernel(const T* __restrict__ L, long ldl, T* __restrict__ W) {
    using chunk_t = typename Tr<T>::chunk_t;
    constexpr int VEC = Tr<T>::VEC;
    __shared__ __attribute__((aligned(16))) T Lt[64][64];
    const int tid = threadIdx.x;
    const T* Lj = L + (long)blockIdx.x * 64 * ldl + (long)blockIdx.x * 64;
    const chunk_t* lrow = reinterpret_cast<const chunk_t*>(Lj + (long)tid * ldl);
#pragma unroll
    for (int t = 0; t < 64 / VEC; ++t) {
        const chunk_t v = lrow[t];
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            const int c = VEC * t + e;
            Lt[c][tid] = (c == tid) ? T(1) / v[e] : v[e];
        }
    }
    __syncthreads();
    T x[64];
#pragma unroll
    for (int t = 0; t < 64; ++t) x[t] = (t == tid) ? T(1) : T(0);
#pragma unroll
    for (int c = 0; c < 64; ++c) {
        const T v = x[c] * Lt[c][c];
        x[c] = v;
#pragma unroll
        for (int t = c + 1; t < 64; ++t) x[t] = fma(-v, Lt[c][t], x[t]);
    }
    T* Wj = W + (long)blockIdx.x * 64 * 64;
#pragma unroll
    for (int t = 0; t < 64; ++t) Wj[t * 64 + tid] = ((t == tid) ? T(1) : T(0)) - ((t >= tid) ? x[t] : T(0));
}

// trtri_64_neg (batched): the base of the level-wise inverse of a lower block (gpmi355.hip eng_inv_lower): for every 64×64 diagonal tile j, −inv(L_jj) into the
// pitched matrix W (lower, zeros above the diagonal of the tile) and its transpose into WT (upper).  Same substitution as trtri_64_kernel.
template <typename T>
__global__ __launch_bounds__(64) void trtri_64_neg_kernel(const T* __restrict__ L, long ldl, T* __restrict__ W, T* __restrict__ WT, long ldw) {
    using chunk_t = typename Tr<T>::chunk_t;
    constexpr int VEC = Tr<T>::VEC;
    __shared__ __attribute__((aligned(16))) T Lt[64][64];
    const int tid = threadIdx.x;
    const long j0 = (long)blockIdx.x * 64;
    const T* Lj = L + j0 * ldl + j0;
    const chunk_t* lrow = reinterpret_cast<const chunk_t*>(Lj + (long)tid * ldl);
#pragma unroll
    for (int t = 0; t < 64 / VEC; ++t) {
        const chunk_t v = lrow[t];
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            const int c = VEC * t + e;
            Lt[c][tid] = (c == tid) ? T(1) / v[e] : v[e];
        }
    }
    __syncthreads();
    T x[64];
#pragma unroll
    for (int t = 0; t < 64; ++t) x[t] = (t == tid) ? T(1) : T(0);
#pragma unroll
    for (int c = 0; c < 64; ++c) {
        const T v = x[c] * Lt[c][c];
        x[c] = v;
#pragma unroll
        for (int t = c + 1; t < 64; ++t) x[t] = fma(-v, Lt[c][t], x[t]);
    }
    // lane tid holds column tid of inv(L_jj): x[t] = inv[t][tid] for t >= tid
    T* Wj = W + j0 * ldw + j0;
    T* WTj = WT + j0 * ldw + j0;
#pragma unroll
    for (int t = 0; t < 64; ++t) {
        const T v = (t >= tid) ? -x[t] : T(0);
        Wj[(long)t * ldw + tid] = v;
        WTj[(long)tid * ldw + t] = v;
    }
}

// ------------------------------------------------------------------------------------------------
// Vector triangular solves, blocked by NBV = 1024 with 64-wide inner steps.  nrhs right-hand sides
// are stored as rows: R[s*ldr + i].
//   trsv_diag<FWD>: one workgroup solves the nbv×nbv diagonal block at b0 for all nrhs.
//   trsv_upd_fwd : r[i] -= Σ_j L[i][b0+j] z[b0+j]   rows i >= b0+nbv          (one wave per row)
//   trsv_upd_bwd : r[j] -= Σ_i L[b0+i][j] a[b0+i]   columns j < b0            (atomics over row chunks)
// ------------------------------------------------------------------------------------------------
// The 64-wide steps use the precomputed tiles W_j = I − inv(L_jj) (trtri_64, one batched launch per solve): the step is
// a 64×64 GEMV  x = r − W r  (forward) / x = r − Wᵀ r  (backward) spread over the 16 waves — no 64-step substitution
// chain on the critical path.
template <typename T, bool FWD>
__global__ __launch_bounds__(1024) void trsv_diag_kernel(const T* __restrict__ L, long ldl, long b0, int nbv,
                                                          T* __restrict__ R, long ldr, int nrhs,
                                                          const T* __restrict__ W) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T* rv = reinterpret_cast<T*>(smem_raw);  // [nbv] current rhs / solution
    T* Ws = rv + nbv;                        // [64][65] W tile of the current step
    T* red = Ws + 64 * 65;                   // [16][64] partial sums
    const int tid = threadIdx.x;
    const int t = tid & 63, part = tid >> 6;
    const int ns = nbv / 64;
    for (int s = 0; s < nrhs; ++s) {
        T* r = R + (long)s * ldr + b0;
        for (int i = tid; i < nbv; i += 1024) rv[i] = r[i];
        T wreg[4];  // this thread's share of the NEXT step's W tile: its load is in flight during the current step
        {
            const T* W0 = W + ((b0 >> 6) + (FWD ? 0 : ns - 1)) * 4096;
#pragma unroll
            for (int i = 0; i < 4; ++i) wreg[i] = W0[tid + 1024 * i];
        }
        __syncthreads();
        for (int ss = 0; ss < ns; ++ss) {
            const int sb = FWD ? ss : (ns - 1 - ss);
            const int s0 = sb * 64;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int e = tid + 1024 * i;
                Ws[(e >> 6) * 65 + (e & 63)] = wreg[i];
            }
            if (ss + 1 < ns) {
                const T* Wn = W + ((b0 >> 6) + (FWD ? sb + 1 : sb - 1)) * 4096;
#pragma unroll
                for (int i = 0; i < 4; ++i) wreg[i] = Wn[tid + 1024 * i];
            }
            __syncthreads();
            {
                T acc = 0;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int c = 4 * part + i;
                    acc = fma(FWD ? Ws[t * 65 + c] : Ws[c * 65 + t], rv[s0 + c], acc);
                }
                red[part * 64 + t] = acc;
            }
            __syncthreads();
            if (tid < 64) {
                T acc = 0;
#pragma unroll
                for (int q = 0; q < 16; ++q) acc += red[q * 64 + tid];
                rv[s0 + tid] -= acc;
            }
            __syncthreads();
            // update the not-yet-solved part of this diagonal block
            if (FWD) {
                for (int i = s0 + 64 + tid; i < nbv; i += 1024) {
                    const T* lrow = L + (b0 + i) * ldl + b0 + s0;
                    T acc0 = 0, acc1 = 0;
#pragma unroll
                    for (int c = 0; c < 64; c += 2) {  // all 64 loads of the row segment in flight at once
                        acc0 = fma(lrow[c], rv[s0 + c], acc0);
                        acc1 = fma(lrow[c + 1], rv[s0 + c + 1], acc1);
                    }
                    rv[i] -= acc0 + acc1;
                }
            } else {
                for (int j = tid; j < s0; j += 1024) {
                    T acc0 = 0, acc1 = 0;
#pragma unroll
                    for (int c = 0; c < 64; c += 2) {  // 64 independent (coalesced) loads in flight: one memory round trip per step
                        acc0 = fma(L[(b0 + s0 + c) * ldl + b0 + j], rv[s0 + c], acc0);
                        acc1 = fma(L[(b0 + s0 + c + 1) * ldl + b0 + j], rv[s0 + c + 1], acc1);
                    }
                    rv[j] -= acc0 + acc1;
                }
            }
            __syncthreads();
        }
        for (int i = tid; i < nbv; i += 1024) r[i] = rv[i];
        __syncthreads();
    }
}

// trsv_diag2: the same block solve for nbv <= 256 with every memory round trip of a 64-wide step issued AHEAD of its use: besides
// the W tile, the L elements of the in-block update (64 × up to 192, spread over all 1 024 threads: target index j = tid & 255,
// 16 of the step's 64 source rows per thread quarter) are loaded into registers right after the previous update and are in flight
// during the GEMV phases of the step, so a step costs its LDS traffic and barriers instead of a global round trip behind them
// (round 2: 24 µs per 256-wide block, ~6 µs per step, 64 loads per thread on <= 192 active threads; this kernel: ≈11 µs —
// C4's backward sweep 10.2 -> 6.7 ms, C2's 2.05 -> 1.24 ms, profiles/r3/sweep_trsv.jsonl).  A two-stream variant (this chain on
// one stream, the bulk of every update beside it on another) was built and measured SLOWER — 3.1 / 14.0 ms: two event records and
// two stream waits per 256-column block cost more host time than the overlap saves — and removed.
// One nbv-wide (<= 256) diagonal block of a vector solve by the 1 024 threads of a workgroup: r (global, nbv entries at the block's offset) <- L_bb⁻¹ r (FWD)
// or L_bb⁻ᵀ r; smem: the dynamic LDS of trsv_diag2_kernel.  Every thread of the workgroup must call it (workgroup barriers inside).
template <typename T, bool FWD>
__device__ __forceinline__ void trsv_diag2_block(const T* __restrict__ L, long ldl, long b0, int nbv, T* __restrict__ r, const T* __restrict__ W,
                                                 unsigned char* smem_raw) {
    T* rv = reinterpret_cast<T*>(smem_raw);  // [256] current rhs / solution
    T* Ws = rv + 256;                        // [64][65] W tile of the current step
    T* red = Ws + 64 * 65;                   // [16][64] GEMV partial sums
    T* prt = red + 16 * 64;                  // [4][256] update partial sums
    const int tid = threadIdx.x;
    const int t = tid & 63, part = tid >> 6;
    const int j = tid & 255, q = tid >> 8;
    const int ns = nbv / 64;
    auto load_l = [&](int sb, T (&lr)[16]) {  // L elements of the update that follows step sb
        const int s0 = sb * 64;
        if (FWD) {
            const int i = s0 + 64 + j;       // target row
            if (i < nbv) {
                const T* src = L + (b0 + i) * ldl + b0 + s0 + 16 * q;
#pragma unroll
                for (int c = 0; c < 16; ++c) lr[c] = src[c];
            }
        } else if (j < s0) {                 // target column j
            const T* src = L + (b0 + s0 + 16 * q) * ldl + b0 + j;
#pragma unroll
            for (int c = 0; c < 16; ++c) lr[c] = src[(long)c * ldl];
        }
    };
    T wreg[4], lreg[16];
    {
        const T* W0 = W + ((b0 >> 6) + (FWD ? 0 : ns - 1)) * 4096;
#pragma unroll
        for (int i = 0; i < 4; ++i) wreg[i] = W0[tid + 1024 * i];
    }
    if (tid < nbv) rv[tid] = r[tid];
    load_l(FWD ? 0 : ns - 1, lreg);  // in flight during the first step's GEMV
    __syncthreads();
    for (int ss = 0; ss < ns; ++ss) {
        const int sb = FWD ? ss : (ns - 1 - ss);
        const int s0 = sb * 64;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = tid + 1024 * i;
            Ws[(e >> 6) * 65 + (e & 63)] = wreg[i];
        }
        if (ss + 1 < ns) {  // next step's W tile: in flight during this step
            const T* Wn = W + ((b0 >> 6) + (FWD ? sb + 1 : sb - 1)) * 4096;
#pragma unroll
            for (int i = 0; i < 4; ++i) wreg[i] = Wn[tid + 1024 * i];
        }
        __syncthreads();
        {
            T acc = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int c = 4 * part + i;
                acc = fma(FWD ? Ws[t * 65 + c] : Ws[c * 65 + t], rv[s0 + c], acc);
            }
            red[part * 64 + t] = acc;
        }
        __syncthreads();
        if (tid < 64) {
            T acc = 0;
#pragma unroll
            for (int qq = 0; qq < 16; ++qq) acc += red[qq * 64 + tid];
            rv[s0 + tid] -= acc;
        }
        __syncthreads();
        // update the not-yet-solved part of this diagonal block from the registers loaded one step ago
        const int tgt = FWD ? s0 + 64 + j : j;
        const bool active = FWD ? (tgt < nbv) : (j < s0);
        {
            T acc = 0;
            if (active) {
#pragma unroll
                for (int c = 0; c < 16; ++c) acc = fma(lreg[c], rv[s0 + 16 * q + c], acc);
            }
            prt[q * 256 + j] = acc;
        }
        if (ss + 1 < ns) load_l(FWD ? sb + 1 : sb - 1, lreg);  // the next update's elements: in flight during the next step's GEMV
        __syncthreads();
        if (q == 0 && active) rv[tgt] -= prt[j] + prt[256 + j] + prt[512 + j] + prt[768 + j];
        __syncthreads();
    }
    if (tid < nbv) r[tid] = rv[tid];
    __syncthreads();
}

template <typename T, bool FWD>
__global__ __launch_bounds__(1024) void trsv_diag2_kernel(const T* __restrict__ L, long ldl, long b0, int nbv,
                                                           T* __restrict__ R, long ldr, int nrhs,
                                                           const T* __restrict__ W) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    for (int s = 0; s < nrhs; ++s) trsv_diag2_block<T, FWD>(L, ldl, b0, nbv, R + (long)s * ldr + b0, W, smem_raw);
}

// (Round 6 built the whole sweep as ONE persistent launch — a solver workgroup taking the diagonal blocks in order, one workgroup per slice of every later block
//  applying the published blocks, in-kernel progress counters with agent-scope release / acquire — verified it against this path and measured it SLOWER at
//  every size: N = 4 096 pair 2.20 -> 2.41 ms, C2 28.8 -> 31.5, N = 32 768 185.3 -> 196.4, C4 1 383 -> 1 445, C5 76.8 -> 77.5 (profiles/r6/trsv_persist_ab.jsonl):
//  a device-scope release is a write-back of the XCD's whole L2 (buffer_wbl2 sc1), an acquire its invalidation (buffer_inv sc1), and the two hand-overs per
//  block cost ≈ 59 µs where the two launches they replace take ≈ 18 (3 µs of dispatch each, the rest their own work).  Removed; in the history at f29d505: "Vector solves as one persistent launch".)
// rows [row_lo, row_hi): r[s][i] -= Σ_{j<nbv} L[i][b0+j] z[s][b0+j]; one wave per row, 4 rows per block.
template <typename T>
__global__ __launch_bounds__(256) void trsv_upd_fwd_kernel(const T* __restrict__ L, long ldl, long b0, int nbv,
                                                            long row_lo, long row_hi, T* __restrict__ R, long ldr,
                                                            int nrhs) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const long i = row_lo + (long)blockIdx.x * 4 + w;
    if (i >= row_hi) return;
    const T* lrow = L + i * ldl + b0;
    for (int s = 0; s < nrhs; ++s) {
        const T* z = R + (long)s * ldr + b0;
        T acc = 0;
        for (int j = lane; j < nbv; j += 64) acc = fma(lrow[j], z[j], acc);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
        if (lane == 0) R[(long)s * ldr + i] -= acc;
    }
}

// columns j < b0: r[s][j] -= Σ_{i<nbv} L[b0+i][j] a[s][b0+i].  grid (ceil(b0/256), nbv/64): each block
// takes 64 rows × 256 columns, one column per thread, then one atomicAdd per column.
template <typename T>
__global__ __launch_bounds__(256) void trsv_upd_bwd_kernel(const T* __restrict__ L, long ldl, long b0, int nbv,
                                                            T* __restrict__ R, long ldr, int nrhs) {
    const long j = (long)blockIdx.x * 256 + threadIdx.x;
    if (j >= b0) return;
    if (gridDim.y == 1) {  // "deterministic": one thread owns its column, the 64-row pieces are summed in a fixed order, no atomics
        for (int s = 0; s < nrhs; ++s) {
            const T* a = R + (long)s * ldr;
            T tot = 0;
            for (int ch = 0; ch < nbv / 64; ++ch) {
                const long i0 = b0 + 64L * ch;
                T acc = 0;
#pragma unroll 8
                for (int i = 0; i < 64; ++i) acc = fma(L[(i0 + i) * ldl + j], a[i0 + i], acc);
                tot += acc;
            }
            R[(long)s * ldr + j] -= tot;
        }
        return;
    }
    const long i0 = b0 + (long)blockIdx.y * 64;
    for (int s = 0; s < nrhs; ++s) {
        const T* a = R + (long)s * ldr;
        T acc = 0;
#pragma unroll 8
        for (int i = 0; i < 64; ++i) acc = fma(L[(i0 + i) * ldl + j], a[i0 + i], acc);
        atomicAdd(R + (long)s * ldr + j, -acc);
    }
}

// out[row] = Σ_{c<ncols} X[row][c]²  (one block per row)
template <typename T>
__global__ __launch_bounds__(256) void rowsumsq_kernel(const T* __restrict__ X, long ldx, long ncols,
                                                        double* __restrict__ out) {
    __shared__ double red[4];
    const T* x = X + (long)blockIdx.x * ldx;
    double acc = 0;
    for (long c = threadIdx.x; c < ncols; c += 256) {
        const double v = (double)x[c];
        acc = fma(v, v, acc);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// out[0] += Σ_{row<nrows, c<ncols} X[row][c]²  accumulated in fp64 whatever T is (one block per row, one atomic each)
template <typename T>
__global__ __launch_bounds__(256) void sumsq_accum_kernel(const T* __restrict__ X, long ldx, long ncols,
                                                           double* __restrict__ out) {
    __shared__ double red[4];
    const T* x = X + (long)blockIdx.x * ldx;
    double acc = 0;
    for (long c = threadIdx.x; c < ncols; c += 256) {
        const double v = (double)x[c];
        acc = fma(v, v, acc);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, red[0] + red[1] + red[2] + red[3]);
}

// acc[row] -= Σ_{c<ncols} X[row][c] · b[c]   (fp64 accumulation; one block per row)
template <typename T>
__global__ __launch_bounds__(256) void rowdot_sub_kernel(const T* __restrict__ X, long ldx, long ncols, const T* __restrict__ b,
                                                          double* __restrict__ acc_out) {
    __shared__ double red[4];
    const T* x = X + (long)blockIdx.x * ldx;
    double acc = 0;
    for (long c = threadIdx.x; c < ncols; c += 256) acc = fma((double)x[c], (double)b[c], acc);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) acc_out[blockIdx.x] -= red[0] + red[1] + red[2] + red[3];
}

// out[s][i] = Σ_{j<=i} L[i][j] ξ[s][j]   (C.U' * ξ for the row-major lower factor; one block per row i, ξ / out stored
// as rows of length ldv)                                                       src/finite_gp_projection.jl:233-237
template <typename T>
__global__ __launch_bounds__(256) void trmv_lower_kernel(const T* __restrict__ L, long ldl, const T* __restrict__ xi, long ldv,
                                                          int nrhs, T* __restrict__ out) {
    __shared__ double red[4];
    const long i = blockIdx.x;
    const T* l = L + i * ldl;
    for (int s = 0; s < nrhs; ++s) {
        const T* x = xi + (long)s * ldv;
        double acc = 0;
        for (long j = threadIdx.x; j <= i; j += 256) acc = fma((double)l[j], (double)x[j], acc);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
        __syncthreads();
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
        __syncthreads();
        if (threadIdx.x == 0) out[(long)s * ldv + i] = (T)(red[0] + red[1] + red[2] + red[3]);
    }
}
// rows [n, np) of the np×np row-major matrix become identity rows (padding of a factor)
template <typename T>
__global__ __launch_bounds__(256) void pad_identity_kernel(T* __restrict__ A, long lda, long n, long np) {
    const long i = n + blockIdx.y;
    const long j = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < np && j < np) A[i * lda + j] = (i == j) ? T(1) : T(0);
}

// out[s] = Σ_i variance κ(‖xs_s − x_i‖) α_i   (K_*x α fused with the kernel evaluation; one block per s)
template <typename T>
__global__ __launch_bounds__(256) void kvec_kernel(const T* __restrict__ xs, long ldxs, const T* __restrict__ x,
                                                    long ldx, int d, int kind, T variance, long n,
                                                    const T* __restrict__ alpha, T* __restrict__ out) {
    __shared__ double red[4];
    const long s = blockIdx.x;
    double acc = 0;
    for (long i = threadIdx.x; i < n; i += 256) {
        T d2 = 0;
        for (int dd = 0; dd < d; ++dd) {
            const T t = xs[(long)dd * ldxs + s] - x[(long)dd * ldx + i];
            d2 = fma(t, t, d2);
        }
        acc += (double)(variance * kappa<T>(kind, d2) * alpha[i]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) out[s] = (T)(red[0] + red[1] + red[2] + red[3]);
}

// ---- small M×M helpers of the VFE path ------------------------------------------------------------
// dst[i][j] = -(double) src[max(i,j)][min(i,j)]   (the SYRK accumulator holds -G in its lower triangle)
template <typename T>
__global__ __launch_bounds__(256) void neg_sym_to_f64_kernel(const T* __restrict__ src, long lds, double* __restrict__ dst,
                                                              long ldd, long n) {
    const long j = (long)blockIdx.x * 256 + threadIdx.x, i = blockIdx.y;
    if (j >= n) return;
    const long a = i > j ? i : j, b = i > j ? j : i;
    dst[i * ldd + j] = -(double)src[a * lds + b];
}
// dst[i][j] += (double) src[i][j] for j <= i  (fp64 accumulation of an fp32 partial SYRK result, lower triangle)
template <typename T>
__global__ __launch_bounds__(256) void add_lower_to_f64_kernel(const T* __restrict__ src, long lds, double* __restrict__ dst,
                                                                long ldd, long n) {
    const long j = (long)blockIdx.x * 256 + threadIdx.x, i = blockIdx.y;
    if (j <= i && j < n) dst[i * ldd + j] += (double)src[i * lds + j];
}
// dst = srcᵀ (n×n, 32×32 LDS tiles)
// dst[j][i] = scale · src[i][j] for an n×n block (n multiple of 32): 32×32 tiles through LDS; the inverse diagonal blocks of a factor
// (L_bb⁻ᵀ upper, row-major -> −L_bb⁻¹ lower, row-major: the B operand of the triangular-k GEMM that replaces a TRSM leaf)
template <typename T>
__global__ __launch_bounds__(256) void transpose_scale_kernel(const T* __restrict__ src, long lds_, T* __restrict__ dst, long ldd, long n, T scale,
                                                               long sstride = 0, long dstride = 0) {  // blockIdx.z = b: block b of a batch
    src += (long)blockIdx.z * sstride;
    dst += (long)blockIdx.z * dstride;
    __shared__ T tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const long i0 = (long)blockIdx.y * 32, j0 = (long)blockIdx.x * 32;
#pragma unroll
    for (int r = ty; r < 32; r += 8) tile[r][tx] = (i0 + r < n && j0 + tx < n) ? src[(i0 + r) * lds_ + j0 + tx] : T(0);
    __syncthreads();
#pragma unroll
    for (int r = ty; r < 32; r += 8)
        if (j0 + r < n && i0 + tx < n) dst[(j0 + r) * ldd + i0 + tx] = scale * tile[tx][r];
}
__global__ __launch_bounds__(256) void transpose_f64_kernel(const double* __restrict__ src, long lds, double* __restrict__ dst,
                                                            long ldd, long n) {
    __shared__ double t[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const long r0 = (long)blockIdx.y * 32, c0 = (long)blockIdx.x * 32;
    for (int k = ty; k < 32; k += 8) t[k][tx] = src[(r0 + k) * lds + c0 + tx];
    __syncthreads();
    for (int k = ty; k < 32; k += 8) dst[(c0 + k) * ldd + r0 + tx] = t[tx][k];
}
// A[i][i] += v for i < n; out[0] = Σ_{i<n_valid} A[i][i] BEFORE the shift (one block)
__global__ __launch_bounds__(256) void diag_shift_trace_kernel(double* __restrict__ A, long lda, long n, long n_valid, double v,
                                                               double* __restrict__ out) {
    __shared__ double red[4];
    double acc = 0;
    for (long i = threadIdx.x; i < n; i += 256) {
        const double d = A[i * lda + i];
        if (i < n_valid) acc += d;
        A[i * lda + i] = d + v;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0 && out) out[0] = red[0] + red[1] + red[2] + red[3];
}
// A = I (n×n, leading dimension lda)
// out[0] = min_i |A[i][i]|, out[1] = max_i |A[i][i]| over i < n (one workgroup): max / min of a Cholesky factor's diagonal is a lower bound of its
// condition number — the guard of the inverse-diagonal-block solves (gpmi355.hip trsm_post)
template <typename T>
__global__ __launch_bounds__(1024) void diag_minmax_kernel(const T* __restrict__ A, long lda, long n, double* __restrict__ out) {
    __shared__ double smin[16], smax[16];
    double mn = 1e300, mx = 0.0;
    for (long i = threadIdx.x; i < n; i += 1024) {
        const double v = fabs((double)A[i * lda + i]);
        mn = v < mn ? v : mn;
        mx = v > mx ? v : mx;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const double a = __shfl_xor(mn, o, 64), b = __shfl_xor(mx, o, 64);
        mn = a < mn ? a : mn;
        mx = b > mx ? b : mx;
    }
    if ((threadIdx.x & 63) == 0) {
        smin[threadIdx.x >> 6] = mn;
        smax[threadIdx.x >> 6] = mx;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 16; ++w) {
            mn = smin[w] < mn ? smin[w] : mn;
            mx = smax[w] > mx ? smax[w] : mx;
        }
        out[0] = mn;
        out[1] = mx;
    }
}
// A[b·stride + i·lda + i] = 1 for i < n, b < gridDim.y (the diagonals of a batch of n×n blocks in a zeroed buffer)
template <typename T>
__global__ __launch_bounds__(256) void diag_ones_kernel(T* __restrict__ A, long lda, long n, long stride) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) A[(long)blockIdx.y * stride + i * lda + i] = T(1);
}
template <typename T>
__global__ __launch_bounds__(256) void identity_kernel(T* __restrict__ A, long lda, long n) {
    const long j = (long)blockIdx.x * 256 + threadIdx.x, i = blockIdx.y;
    if (j < n) A[i * lda + j] = (i == j) ? T(1) : T(0);
}
template <typename TS, typename TD>
__global__ __launch_bounds__(256) void convert_kernel(const TS* __restrict__ src, TD* __restrict__ dst, long n, double scale) {
    // grid-stride: a launch may not carry 2³² work-items per dimension (the runtime wraps the count WITHOUT an error — round 6: the gradient's sign fold
    // over np·ld = 65 536 · 65 568 > 2³² elements touched the first 2 M only; launches go through convert_grid below)
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) dst[i] = (TD)(scale * (double)src[i]);
}
static inline dim3 convert_grid(long n) { return dim3((unsigned)std::min<long>((n + 255) / 256, 1L << 20)); }
// r[j] -= Σ_{i<nrows} L[i][j] a[i]  for j < ncols.  grid (ceil(ncols/256), ceil(nrows/64)); one atomic per (block, column)
template <typename T, typename RT = T>
__global__ __launch_bounds__(256) void gemv_t_kernel(const T* __restrict__ L, long ldl, long nrows, long ncols,
                                                      const T* __restrict__ a, RT* __restrict__ r) {
    const long j = (long)blockIdx.x * 256 + threadIdx.x;
    const long i0 = (long)blockIdx.y * 64;
    if (j >= ncols) return;
    const long i1 = (i0 + 64 < nrows) ? i0 + 64 : nrows;
    RT acc = 0;
    for (long i = i0; i < i1; ++i) acc = fma((RT)L[i * ldl + j], (RT)a[i], acc);
    atomicAdd(r + j, -acc);
}

// dst[i][j] += Σ_b (double) S[b·cstride + i·lds + j]  for row_lo <= i < n, j <= i: the split-K partial products of one chunk's
// fp32 SYRK summed into the fp64 accumulator in ONE pass (was one launch per partial)
template <typename T>
__global__ __launch_bounds__(256) void add_lower_batched_kernel(const T* __restrict__ S, long cstride, int nbatch, long lds,
                                                                 double* __restrict__ dst, long ldd, long n, long row_lo) {
    // four consecutive columns per thread (16-byte streaming loads of the partials: they are read once and must not push the GEMM operands out of
    // the caches they share; rows start 16-byte aligned: lds and ldd are multiples of 4) — a quarter of the wave-instructions of the one-column form,
    // so the launch holds the issue slots it shares with the chunk GEMMs for a quarter of the time
    const long j0 = ((long)blockIdx.x * 256 + threadIdx.x) * 4, i = row_lo + blockIdx.y;
    if (j0 > i || j0 >= n) return;
    double acc[4] = {0, 0, 0, 0};
    if constexpr (sizeof(T) == 4) {
        typedef float f4_t __attribute__((ext_vector_type(4)));
        for (int b = 0; b < nbatch; ++b) {
            const f4_t v = __builtin_nontemporal_load(reinterpret_cast<const f4_t*>(S + (long)b * cstride + i * lds + j0));
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] += (double)v[e];
        }
    } else {
        for (int b = 0; b < nbatch; ++b)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] += (double)S[(long)b * cstride + i * lds + j0 + e];
    }
    double* d = dst + i * ldd + j0;
#pragma unroll
    for (int e = 0; e < 4; ++e)
        if (j0 + e <= i && j0 + e < n) d[e] += acc[e];
}
// One pass over row (row_lo + blockIdx.x) of Y = −B_c:  rowss[row] += Σ_c Y²  (‖B‖²_F per inducing row, fp64) and
// cacc[row] −= Σ_c Y·b  (c = B b_y)                                         src/sparse_approximations.jl:66-71, 251
template <typename T>
__global__ __launch_bounds__(256) void ystats_kernel(const T* __restrict__ Y, long ldy, long ncols, const T* __restrict__ b,
                                                      long row_lo, double* __restrict__ cacc, double* __restrict__ rowss) {
    __shared__ double red[2][4];
    const long row = row_lo + blockIdx.x;
    const T* y = Y + row * ldy;
    double ss = 0, dot = 0;
    if constexpr (sizeof(T) == 4) {  // 16-byte streaming loads, four columns per thread and pass (rows of Y and the chunk's piece of b start 16-byte aligned)
        typedef float f4_t __attribute__((ext_vector_type(4)));
        const long nc4 = ncols & ~3L;
        for (long c = 4L * threadIdx.x; c < nc4; c += 1024) {
            const f4_t v = __builtin_nontemporal_load(reinterpret_cast<const f4_t*>(y + c));
            const f4_t w = *reinterpret_cast<const f4_t*>(b + c);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const double vv = (double)v[e];
                ss = fma(vv, vv, ss);
                dot = fma(vv, (double)w[e], dot);
            }
        }
        for (long c = nc4 + threadIdx.x; c < ncols; c += 256) {
            const double v = (double)y[c];
            ss = fma(v, v, ss);
            dot = fma(v, (double)b[c], dot);
        }
    } else {  // fp64: the same 16-byte streaming loads, two columns each, two loads in flight per thread (the scalar form ran at 155 GB/s beside the fp64 GEMMs)
        typedef double d2v_t __attribute__((ext_vector_type(2)));
        const long nc4 = ncols & ~3L;
        for (long c = 4L * threadIdx.x; c < nc4; c += 1024) {
            const d2v_t v0 = __builtin_nontemporal_load(reinterpret_cast<const d2v_t*>(y + c));
            const d2v_t v1 = __builtin_nontemporal_load(reinterpret_cast<const d2v_t*>(y + c + 2));
            const d2v_t w0 = *reinterpret_cast<const d2v_t*>(b + c);
            const d2v_t w1 = *reinterpret_cast<const d2v_t*>(b + c + 2);
            ss = fma(v0[0], v0[0], ss); dot = fma(v0[0], w0[0], dot);
            ss = fma(v0[1], v0[1], ss); dot = fma(v0[1], w0[1], dot);
            ss = fma(v1[0], v1[0], ss); dot = fma(v1[0], w1[0], dot);
            ss = fma(v1[1], v1[1], ss); dot = fma(v1[1], w1[1], dot);
        }
        for (long c = nc4 + threadIdx.x; c < ncols; c += 256) {
            const double v = (double)y[c];
            ss = fma(v, v, ss);
            dot = fma(v, (double)b[c], dot);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        ss += __shfl_xor(ss, o, 64);
        dot += __shfl_xor(dot, o, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        red[0][threadIdx.x >> 6] = ss;
        red[1][threadIdx.x >> 6] = dot;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        rowss[row] += red[0][0] + red[0][1] + red[0][2] + red[0][3];
        cacc[row] -= red[1][0] + red[1][1] + red[1][2] + red[1][3];
    }
}
// dst[r][c] = src[r][c] for r < rows, c < cols (cols multiple of 2; 16-B accesses; grid-stride over rows): the 2-D block copy of
// the multi-device driver when source and destination are directly addressable (same device, or a peer over xGMI)
__global__ __launch_bounds__(256) void copy2d_kernel(double* __restrict__ dst, long dld, const double* __restrict__ src, long sld,
                                                     long rows, long cols) {
    const long c2 = cols >> 1;
    for (long r = blockIdx.x; r < rows; r += gridDim.x) {
        const d2_t* s2 = reinterpret_cast<const d2_t*>(src + r * sld);
        d2_t* d2 = reinterpret_cast<d2_t*>(dst + r * dld);
        for (long c = threadIdx.x; c < c2; c += 256) d2[c] = s2[c];
    }
}
// dst[i] += src[i]
__global__ __launch_bounds__(256) void axpy_kernel(double* __restrict__ dst, const double* __restrict__ src, long n) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] += src[i];
}
// A[i][i] += v[i] for i < n   (Σy* on the diagonal of a predictive covariance, src/finite_gp_projection.jl:133-136)
template <typename T>
__global__ __launch_bounds__(256) void diag_add_kernel(T* __restrict__ A, long lda, const T* __restrict__ v, long n) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) A[i * lda + i] += v[i];
}
// out[s][i] += m[i] for i < n  (rows of length ldv)
template <typename T>
__global__ __launch_bounds__(256) void add_rowvec_kernel(T* __restrict__ out, long ldv, const T* __restrict__ m, long n) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[(long)blockIdx.y * ldv + i] += m[i];
}

// MFMA layout / rate probe: D = A·B for one 16×16×4 tile (A row-major 16×4, B row-major 4×16).
__global__ void mfma_probe_f64_kernel(const double* A, const double* B, double* D) {
    const int l = threadIdx.x;
    d4_t c = (d4_t)(0.0);
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(A[(l & 15) * 4 + (l >> 4)], B[(l >> 4) * 16 + (l & 15)], c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[((l >> 4) + 4 * r) * 16 + (l & 15)] = c[r];
}
// Peak-rate microbenchmark: one 1024-thread block per CU (a large dynamic-LDS request pins residency to
// 1 block/CU), i.e. 4 waves per SIMD, 4 independent accumulators per wave, iters back-to-back MFMAs.
__global__ __launch_bounds__(1024) void mfma_rate_f64_kernel(double* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char rate_smem[];
    d4_t acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = (d4_t)((double)i);
    double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 12345.678) {
        out[0] = s;  // keep live
        rate_smem[threadIdx.x] = 1;
    }
}

// fp32 counterparts of mfma_rate_f64_kernel: back-to-back v_mfma_f32_16x16x4_f32 (VAR 0) or v_mfma_f32_32x32x2_f32 (VAR 1)
typedef float f16_t __attribute__((ext_vector_type(16)));
template <int VAR> __global__ __launch_bounds__(1024) void mfma_rate_f32_kernel(float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char rate_smem32[];
    float a = 1.0f + threadIdx.x * 1e-6f, b = 1.0f - threadIdx.x * 1e-6f;
    float s = 0;
    if constexpr (VAR == 0) {
        f4_t acc[4];
        for (int i = 0; i < 4; ++i) acc[i] = (f4_t)((float)i);
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
        }
        for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    } else {
        f16_t acc[4];
        for (int i = 0; i < 4; ++i) acc[i] = (f16_t)((float)i);
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
        }
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 16; ++j) s += acc[i][j];
    }
    if (s == 12345.678f) {
        out[0] = s;
        rate_smem32[threadIdx.x] = 1;
    }
}

}  // namespace gpmi
