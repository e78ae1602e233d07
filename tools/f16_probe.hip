// Where do the cycles of the 16-column diagonal chain go?  The factor16 loop of leaf.hpp stand-alone (one wave), with parts removed:
//   V=0 full | 1 no W-update MFMA | 2 no MFMAs at all | 3 no 1/sqrt chain (sel constant) | 4 no broadcasts (l, dnext constants) | 5 only the 2 MFMAs
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form tools/f16_probe.hip -o tools/bin/f16_probe
#include "../abstractgps.jl_amd/csrc/kcommon.hpp"
#include <cstdio>
using namespace gpmi;

typedef __attribute__((address_space(3))) void lds_void_p_t;
__device__ __forceinline__ double rcp_full(double x) {
    const double r = __builtin_amdgcn_rcp(x);
    const double e = fma(-x, r, 1.0);
    const double q = fma(e, e, e);
    return fma(r, q, r);
}
__device__ __forceinline__ unsigned lds_addr(const void* p) { return (unsigned)(unsigned long)(lds_void_p_t*)p; }
__device__ __forceinline__ void lds_put_f64(unsigned addr, double v) { asm volatile("ds_write_b64 %0, %1" ::"v"(addr), "v"(v) : "memory"); }
__device__ __forceinline__ void lds_put_i32(unsigned addr, int v) { asm volatile("ds_write_b32 %0, %1" ::"v"(addr), "v"(v) : "memory"); }
template <int V> __global__ void f16(double* out, long* cyc, const double* in) {
    using TR = Tr<double>;
    const int lane = threadIdx.x & 63, li = lane & 15, lg = lane >> 4;
    d4_t accA, accW, Ls, Ws;
    for (int r = 0; r < 4; ++r) { accA[r] = in[lane * 4 + r]; accW[r] = (li == lg + 4 * r) ? 1.0 : 0.0; Ls[r] = 0; Ws[r] = 0; }
    double sel = (lg == 0) ? fast_rsqrt<double>(lane_bcast<double>(accA[0], 0)) : 0.0;
    long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    asm volatile("" : "+v"(sel));
#pragma unroll
    for (int rep = 0; rep < 4; ++rep) {
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            const int k = c & 3, p = c >> 2;
            const double pa = accA[p] * sel, npa = -pa;
            double l = 0.1, dnext = 2.0;
            const int c1 = (c + 1) & 15, k1 = c1 & 3, p1i = c1 >> 2;
            if (V != 4 && V != 5) {
                l = lane_bcast<double>(pa, 16 * k + c1);
                dnext = lane_bcast<double>(accA[p1i], 16 * k1 + c1);
            }
            if (V != 2) accA = TR::mfma(npa, pa, accA);
            __builtin_amdgcn_sched_barrier(0);
            const double pw = accW[p] * sel;
            if (V != 5) { Ls[p] = (lg == k) ? pa : Ls[p]; Ws[p] = (lg == k) ? pw : Ws[p]; }
            if (V != 3 && V != 5) {
                const double piv = fma(-l, l, dnext);
                const double ri = fast_rsqrt<double>(piv);
                sel = (lg == k1) ? ri : 0.0;
            } else {
                sel = (lg == k1) ? (0.7 + 1e-9 * l) : 0.0;
            }
            __builtin_amdgcn_sched_barrier(0);
            if (V != 1 && V != 2) accW = TR::mfma(npa, pw, accW);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    asm volatile("" : "+v"(sel));
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    double s = sel;
    for (int r = 0; r < 4; ++r) s += accA[r] + accW[r] + Ls[r] + Ws[r];
    out[lane] = s;
    if (lane == 0) cyc[V] = t1 - t0;
}

// LDL form: unscaled columns, reciprocal pivot (v_rcp + one cubic step) on the chain, B operands are the raw accumulator registers
template <int V> __global__ void f16l(double* out, long* cyc, const double* in) {
    using TR = Tr<double>;
    const int lane = threadIdx.x & 63, li = lane & 15, lg = lane >> 4;
    d4_t accA, accW, Ls, Ws;
    for (int r = 0; r < 4; ++r) { accA[r] = in[lane * 4 + r]; accW[r] = (li == lg + 4 * r) ? 1.0 : 0.0; Ls[r] = 0; Ws[r] = 0; }
    double nm[4];
    for (int k = 0; k < 4; ++k) nm[k] = (lg == k) ? -1.0 : 0.0;
    double rcp = rcp_full(lane_bcast<double>(accA[0], 0));
    double selr = rcp * nm[0];
    long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    asm volatile("" : "+v"(rcp));
#pragma unroll
    for (int rep = 0; rep < 4; ++rep) {
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            const int k = c & 3, p = c >> 2;
            const int c1 = (c + 1) & 15, k1 = c1 & 3, p1i = c1 >> 2;
            const double a = accA[p] * selr;
            const double braw = accA[p], wraw = accW[p], dcol = accA[p1i];
            if (V != 2) accA = TR::mfma(a, braw, accA);
            __builtin_amdgcn_sched_barrier(0);
            const double u1 = lane_bcast<double>(braw, 16 * k + c1);
            const double dnext = lane_bcast<double>(dcol, 16 * k1 + c1);
            const double t = u1 * rcp;
            const double piv = fma(-t, u1, dnext);
            rcp = rcp_full(piv);
            selr = rcp * nm[k1];
            if (V != 3) { Ls[p] = (lg == k) ? braw : Ls[p]; Ws[p] = (lg == k) ? wraw : Ws[p]; }
            __builtin_amdgcn_sched_barrier(0);
            if (V != 1 && V != 2) accW = TR::mfma(a, wraw, accW);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    asm volatile("" : "+v"(rcp));
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    double s = rcp;
    for (int r = 0; r < 4; ++r) s += accA[r] + accW[r] + Ls[r] + Ws[r];
    out[lane] = s;
    if (lane == 0) cyc[8 + V] = t1 - t0;
}
// the kernel's own column loop (leaf.hpp, wave 0), V: 0 as in the kernel | 1 no LDS stream | 2 no saves (Ls, dsave) | 3 neither
template <int V> __global__ void f16k(double* out, long* cyc, const double* in) {
    using TR = Tr<double>;
    __shared__ double abuf[16][64];
    __shared__ int aseq;
    const int lane = threadIdx.x & 63, li = lane & 15, lg = lane >> 4;
    d4_t accA, Ls;
    for (int r = 0; r < 4; ++r) { accA[r] = in[lane * 4 + r]; Ls[r] = 0; }
    double nm[4];
    for (int k = 0; k < 4; ++k) nm[k] = (lg == k) ? -1.0 : 0.0;
    double dcur = lane_bcast<double>(accA[0], 0);
    double rcp = rcp_full(dcur), dsave = 1.0;
    const unsigned abuf_a = lds_addr(&abuf[0][lane]), aseq_a = lds_addr(&aseq);
    long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    asm volatile("" : "+v"(rcp));
#pragma unroll
    for (int rep = 0; rep < 4; ++rep) {
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            const int k = c & 3, p = c >> 2;
            const double a = accA[p] * (rcp * nm[k]);
            if (V == 0 || V == 2) { lds_put_f64(abuf_a + c * 64 * 8, a); lds_put_i32(aseq_a, 16 * rep + c + 1); }
            if (V == 0 || V == 1) { Ls[p] = (lg == k) ? accA[p] : Ls[p]; dsave = (lane == c) ? dcur : dsave; }
            const int c1 = (c + 1) & 15, k1 = c1 & 3, p1i = c1 >> 2;
            const double u1 = lane_bcast<double>(accA[p], 16 * k + c1);
            const double dnext = lane_bcast<double>(accA[p1i], 16 * k1 + c1);
            __builtin_amdgcn_sched_barrier(0);
            accA = TR::mfma(a, accA[p], accA);
            __builtin_amdgcn_sched_barrier(0);
            const double t = u1 * rcp;
            dcur = fma(-t, u1, dnext);
            rcp = rcp_full(dcur);
        }
    }
    asm volatile("" : "+v"(rcp));
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    double s = rcp + dsave;
    for (int r = 0; r < 4; ++r) s += accA[r] + Ls[r];
    out[lane] = s + abuf[3][lane];
    if (lane == 0) cyc[12 + V] = t1 - t0;
}
// the same column loop as f16<0>, executed 16 times by a REAL loop (the 16 columns unrolled once): is the straight-line version fetch bound?
__global__ void f16loop(double* out, long* cyc, const double* in) {
    using TR = Tr<double>;
    const int lane = threadIdx.x & 63, li = lane & 15, lg = lane >> 4;
    d4_t accA, accW, Ls, Ws;
    for (int r = 0; r < 4; ++r) { accA[r] = in[lane * 4 + r]; accW[r] = (li == lg + 4 * r) ? 1.0 : 0.0; Ls[r] = 0; Ws[r] = 0; }
    double sel = (lg == 0) ? fast_rsqrt<double>(lane_bcast<double>(accA[0], 0)) : 0.0;
    long t0, t1, tm = 0;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    asm volatile("" : "+v"(sel));
#pragma unroll 1
    for (int rep = 0; rep < 16; ++rep) {
        if (rep == 1) { asm volatile("" : "+v"(sel)); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tm) :: "memory"); asm volatile("" : "+v"(sel)); }
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            const int k = c & 3, p = c >> 2;
            const double pa = accA[p] * sel, npa = -pa;
            const int c1 = (c + 1) & 15, k1 = c1 & 3, p1i = c1 >> 2;
            const double l = lane_bcast<double>(pa, 16 * k + c1);
            const double dnext = lane_bcast<double>(accA[p1i], 16 * k1 + c1);
            accA = TR::mfma(npa, pa, accA);
            __builtin_amdgcn_sched_barrier(0);
            const double pw = accW[p] * sel;
            Ls[p] = (lg == k) ? pa : Ls[p]; Ws[p] = (lg == k) ? pw : Ws[p];
            const double piv = fma(-l, l, dnext);
            const double ri = fast_rsqrt<double>(piv);
            sel = (lg == k1) ? ri : 0.0;
            __builtin_amdgcn_sched_barrier(0);
            accW = TR::mfma(npa, pw, accW);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    asm volatile("" : "+v"(sel));
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    double s = sel;
    for (int r = 0; r < 4; ++r) s += accA[r] + accW[r] + Ls[r] + Ws[r];
    out[lane] = s;
    if (lane == 0) { cyc[6] = tm - t0; cyc[7] = t1 - tm; }
}
int main() {
    double *out, *in; long* cyc;
    hipMalloc(&out, 8 * 64); hipMalloc(&in, 8 * 256); hipMalloc(&cyc, 8 * 16); hipMemset(cyc, 0, 8 * 16);
    double h[256];
    for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) { int row = (l >> 4) + 4 * r, col = l & 15; h[l * 4 + r] = (row == col ? 40.0 : 0.0) + 0.01 * ((row * 7 + col * 13) % 5); }
    hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(f16<0>, dim3(1), dim3(64), 0, 0, out, cyc, in); hipLaunchKernelGGL(f16<1>, dim3(1), dim3(64), 0, 0, out, cyc, in);
        hipLaunchKernelGGL(f16<2>, dim3(1), dim3(64), 0, 0, out, cyc, in); hipLaunchKernelGGL(f16<3>, dim3(1), dim3(64), 0, 0, out, cyc, in);
        hipLaunchKernelGGL(f16<4>, dim3(1), dim3(64), 0, 0, out, cyc, in); hipLaunchKernelGGL(f16<5>, dim3(1), dim3(64), 0, 0, out, cyc, in);
    }
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(f16l<0>, dim3(1), dim3(64), 0, 0, out, cyc, in); hipLaunchKernelGGL(f16l<1>, dim3(1), dim3(64), 0, 0, out, cyc, in);
        hipLaunchKernelGGL(f16l<2>, dim3(1), dim3(64), 0, 0, out, cyc, in); hipLaunchKernelGGL(f16l<3>, dim3(1), dim3(64), 0, 0, out, cyc, in);
    }
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(f16k<0>, dim3(1), dim3(64), 0, 0, out, cyc, in); hipLaunchKernelGGL(f16k<1>, dim3(1), dim3(64), 0, 0, out, cyc, in);
        hipLaunchKernelGGL(f16k<2>, dim3(1), dim3(64), 0, 0, out, cyc, in); hipLaunchKernelGGL(f16k<3>, dim3(1), dim3(64), 0, 0, out, cyc, in);
    }
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(f16loop, dim3(1), dim3(64), 0, 0, out, cyc, in);
    hipDeviceSynchronize();
    long c[16]; hipMemcpy(c, cyc, sizeof(c), hipMemcpyDeviceToHost);
    const char* nm[] = {"full", "no W-update MFMA", "no MFMA at all", "no 1/sqrt chain", "no broadcasts", "only the two MFMAs + 2 mul"};
    for (int v = 0; v < 6; ++v) printf("%-28s %7.1f cycles per column\n", nm[v], (double)c[v] / 64.0);
    printf("%-28s %7.1f cycles per column (first pass of the loop body)   %7.1f (passes 2..16, code resident)\n", "full, as a real loop", (double)c[6] / 16.0, (double)c[7] / 240.0);
    const char* nl[] = {"LDL full", "LDL no W-update MFMA", "LDL no MFMA at all", "LDL no saves"};
    for (int v = 0; v < 4; ++v) printf("%-28s %7.1f cycles per column\n", nl[v], (double)c[8 + v] / 64.0);
    const char* nk[] = {"kernel loop", "kernel loop, no LDS stream", "kernel loop, no saves", "kernel loop, neither"};
    for (int v = 0; v < 4; ++v) printf("%-28s %7.1f cycles per column\n", nk[v], (double)c[12 + v] / 64.0);
    return 0;
}
