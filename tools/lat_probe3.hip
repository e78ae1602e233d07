// What do the 16-lane row moves of gfx950 (v_permlane16_swap / v_permlane32_swap) cost on a dependent chain, alone and between fp64 MFMAs?
// One wave alone, cycles per iteration by s_memtime (as tools/lat_probe2.hip).  For the rank-4 variant of the 16×16 factorisation.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form tools/lat_probe3.hip -o tools/bin/lat_probe3
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4_t __attribute__((ext_vector_type(4)));
typedef unsigned u2_t __attribute__((ext_vector_type(2)));
#define N 32
// every stamp pins ALL live values (x, y and the accumulator), so no segment's arithmetic can move across it
#define PIN() asm volatile("" : "+v"(x), "+v"(y), "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]))
#define STAMP() do { PIN(); long tt_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tt_) :: "memory"); t[n++] = tt_; PIN(); } while (0)

// a 64-bit value of lane group 0 into lane groups 1..3 as well: two swaps per dword
__device__ __forceinline__ double row0_to_all(double v) {
    unsigned long b = __builtin_bit_cast(unsigned long, v);
    unsigned lo = (unsigned)b, hi = (unsigned)(b >> 32);
    u2_t a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);   // rows 1, 3 of the first <-> rows 0, 2 of the second
    u2_t c = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    u2_t e = __builtin_amdgcn_permlane32_swap(a[0], a[0], false, false);  // rows 2, 3 of the first <-> rows 0, 1 of the second
    u2_t f = __builtin_amdgcn_permlane32_swap(c[0], c[0], false, false);
    return __builtin_bit_cast(double, ((unsigned long)f[0] << 32) | e[0]);
}

__global__ void probe(double* out, long* cyc, double seed) {
    double x = seed + threadIdx.x * 1e-9, y = 1.000001, z = 0.5, w = 0.25;
    d4_t acc = {x, x, x, x};
    long t[8]; int n = 0;
    STAMP();
#pragma unroll
    for (int i = 0; i < N; ++i) x = row0_to_all(x) * w;                                                      // 0: move + mul, dependent
    STAMP();
#pragma unroll
    for (int i = 0; i < N; ++i) x = __builtin_fma(x, w, z);                                                  // 1: one dependent fma (reference)
    STAMP();
#pragma unroll
    for (int i = 0; i < N; ++i) {                                                                             // 2: MFMA -> move of its result -> MFMA operand
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(y, z, acc, 0, 0, 0);
        y = row0_to_all(acc[1]) * w;
    }
    STAMP();
#pragma unroll
    for (int i = 0; i < N; ++i) {                                                                             // 3: MFMA -> mul of its result -> MFMA operand (reference: 94)
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(y, z, acc, 0, 0, 0);
        y = acc[1] * w;
    }
    STAMP();
#pragma unroll
    for (int i = 0; i < N; ++i) {                                                                             // 4: one column of the rank-4 chain: readlane pivot -> rsq + Newton -> scale -> move -> readlane scalar -> fma
        long b = __builtin_bit_cast(long, x); int lo = __builtin_amdgcn_readlane((int)b, 5), hi = __builtin_amdgcn_readlane((int)(b >> 32), 5);
        double d = __builtin_bit_cast(double, ((long)hi << 32) | (unsigned)lo);
        double r = __builtin_amdgcn_rsq(d);
        r = __builtin_fma(__builtin_fma(-d * r, r, 1.0), 0.5 * r, r);
        double l = x * r;
        double la = row0_to_all(l);
        long b2 = __builtin_bit_cast(long, la); int lo2 = __builtin_amdgcn_readlane((int)b2, 21), hi2 = __builtin_amdgcn_readlane((int)(b2 >> 32), 21);
        double m = __builtin_bit_cast(double, ((long)hi2 << 32) | (unsigned)lo2);
        x = __builtin_fma(-m, la, x + 3.0);
    }
    STAMP();
    out[threadIdx.x] = x + acc[0] + acc[1] + y;
    if (threadIdx.x == 0) for (int i = 0; i + 1 < n; ++i) cyc[i] = (t[i + 1] - t[i]);
}

int main() {
    double* out; long* cyc;
    hipMalloc(&out, 64 * 8); hipMalloc(&cyc, 8 * 8);
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, out, cyc, 1.5); hipDeviceSynchronize(); }
    long h[8]; hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
    const char* what[] = {"row move (2x permlane16_swap + 2x permlane32_swap) + mul, dependent", "dependent v_fma_f64 (reference)",
                          "MFMA -> row move of its result -> mul -> MFMA operand", "MFMA -> mul of its result -> MFMA operand (reference)",
                          "one column of the rank-4 chain: readlane pivot, rsq + Newton, scale, row move, readlane scalar, fma"};
    for (int i = 0; i < 5; ++i) printf("%-100s %7.1f cycles per iteration (s_memtime ticks x clock ratio not applied)\n", what[i], (double)h[i] / N);
    return 0;
}
