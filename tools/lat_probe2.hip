// Does VALU work overlap a running fp64 MFMA on the same wave?  (one wave alone)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4_t __attribute__((ext_vector_type(4)));
#define N 32
#define STAMP(v) do { asm volatile("" : "+v"(v)); long tt_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tt_) :: "memory"); t[n++] = tt_; asm volatile("" : "+v"(v)); } while (0)
__global__ void probe(double* out, long* cyc, double seed) {
    double x = seed + threadIdx.x * 1e-9, y = 1.000001, z = 0.5, w = 0.25;
    d4_t acc = {x, x, x, x}, acc2 = {y, y, y, y};
    long t[12]; int n = 0;
    STAMP(x);
#pragma unroll
    for (int i = 0; i < N; ++i) { acc = __builtin_amdgcn_mfma_f64_16x16x4f64(y, z, acc, 0, 0, 0); }        // 0: dependent MFMA chain alone
    { double a0 = acc[0]; STAMP(a0); acc[0] = a0; }
#pragma unroll
    for (int i = 0; i < N; ++i) {                                                                             // 1: + 8 dependent fma on x (independent of the MFMA)
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(y, z, acc, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < 8; ++q) x = __builtin_fma(x, w, z);
        __builtin_amdgcn_sched_barrier(0);
    }
    { double a0 = acc[0] + x; STAMP(a0); acc[0] = a0; }
#pragma unroll
    for (int i = 0; i < N; ++i) {                                                                             // 2: + 16 dependent fma
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(y, z, acc, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < 16; ++q) x = __builtin_fma(x, w, z);
        __builtin_amdgcn_sched_barrier(0);
    }
    { double a0 = acc[0] + x; STAMP(a0); acc[0] = a0; }
#pragma unroll
    for (int i = 0; i < N; ++i) {                                                                             // 3: MFMA -> VALU on its result (1 mul) -> MFMA operand
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(y, z, acc, 0, 0, 0);
        y = acc[1] * w;
    }
    { double a0 = acc[0] + y; STAMP(a0); acc[0] = a0; }
#pragma unroll
    for (int i = 0; i < N; ++i) {                                                                             // 4: as 3 with 8 independent fma between
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(y, z, acc, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < 8; ++q) x = __builtin_fma(x, w, z);
        __builtin_amdgcn_sched_barrier(0);
        y = acc[1] * w;
    }
    { double a0 = acc[0] + y + x; STAMP(a0); acc[0] = a0; }
#pragma unroll
    for (int i = 0; i < N; ++i) {                                                                             // 5: two independent MFMA chains interleaved
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(y, z, acc, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, z, acc2, 0, 0, 0);
    }
    { double a0 = acc[0] + acc2[0]; STAMP(a0); acc[0] = a0; }
#pragma unroll
    for (int i = 0; i < N; ++i) {                                                                             // 6: MFMA -> readlane of the result -> (sgpr) mul -> MFMA operand
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(y, z, acc, 0, 0, 0);
        long b = __builtin_bit_cast(long, acc[1]); int lo = __builtin_amdgcn_readlane((int)b, 5), hi = __builtin_amdgcn_readlane((int)(b >> 32), 5);
        y = w * __builtin_bit_cast(double, ((long)hi << 32) | (unsigned)lo);
    }
    { double a0 = acc[0] + y; STAMP(a0); acc[0] = a0; }
    out[threadIdx.x] = x + acc[0] + acc[1] + acc[2] + acc[3] + y + acc2[1];
    if (threadIdx.x == 0) for (int i = 0; i + 1 < n; ++i) cyc[i] = t[i + 1] - t[i];
}
int main() {
    double* out; long* cyc;
    hipMalloc(&out, 8 * 64); hipMalloc(&cyc, 8 * 16);
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, out, cyc, 1.5);
    hipDeviceSynchronize();
    long h[16]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    const char* names[] = {"dependent MFMA chain", "MFMA + 8 dep fma (independent)", "MFMA + 16 dep fma (independent)", "MFMA -> mul(result) -> MFMA operand",
                           "same + 8 independent fma between", "two independent MFMA chains", "MFMA -> readlane(result) -> mul -> MFMA operand"};
    for (int i = 0; i < 7; ++i) printf("%-48s %6.1f cycles/iter\n", names[i], (double)h[i] / N);
    return 0;
}
