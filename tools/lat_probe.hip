// Dependent-chain latencies (cycles per op, one wave alone on a SIMD) of the instructions on the pivot chain of the leaf:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form tools/lat_probe.hip -o tools/bin/lat_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4_t __attribute__((ext_vector_type(4)));
#define N 64
#define STAMP(v) do { asm volatile("" : "+v"(v)); long tt_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tt_) :: "memory"); t[n++] = tt_; asm volatile("" : "+v"(v)); } while (0)
#define REP(...) _Pragma("unroll") for (int i = 0; i < N; ++i) { __VA_ARGS__; }
__global__ void probe(double* out, long* cyc, double seed) {
    double x = seed + threadIdx.x * 1e-9, y = 1.000001, z = 0.5;
    long t[12];
    int n = 0;
    STAMP(x);
    REP(x = __builtin_fma(x, y, z));                               // 0: v_fma_f64 dependent
    STAMP(x);
    REP(x = x * y);                                                // 1: v_mul_f64 dependent
    STAMP(x);
    REP(x = __builtin_amdgcn_rsq(x + 2.0));                        // 2: v_add_f64 + v_rsq_f64 dependent (subtract row 0-ish for the add)
    STAMP(x);
    REP(x = __builtin_amdgcn_rcp(x + 2.0));                        // 3: v_add + v_rcp_f64
    STAMP(x);
    float f = (float)x;
    REP(f = __builtin_amdgcn_rsqf(f + 2.0f));                      // 4: v_add_f32 + v_rsq_f32
    STAMP(f);
    x += (double)f;
    REP({ long b = __builtin_bit_cast(long, x); int lo = __builtin_amdgcn_readlane((int)b, 5), hi = __builtin_amdgcn_readlane((int)(b >> 32), 5);
          x = x * __builtin_bit_cast(double, ((long)hi << 32) | (unsigned)lo); });   // 5: 2 readlane + v_mul_f64 (SGPR operand)
    STAMP(x);
    d4_t acc = {x, x, x, x};
    REP(acc = __builtin_amdgcn_mfma_f64_16x16x4f64(y, z, acc, 0, 0, 0));             // 6: dependent-accumulator MFMA
    { double a0_ = acc[0]; STAMP(a0_); acc[0] = a0_; }
    REP({ acc = __builtin_amdgcn_mfma_f64_16x16x4f64(y, z, acc, 0, 0, 0); y = acc[0] * z; });  // 7: MFMA -> VALU read -> MFMA operand
    STAMP(y);
    REP({ x = (threadIdx.x & 16) ? x : y; x = x * 1.0000001; });                    // 8: 2 cndmask + mul
    STAMP(x);
    REP({ float g = (float)x; g = __builtin_amdgcn_rsqf(g); x = (double)g + 1.0; });  // 9: cvt + rsq_f32 + cvt + add
    STAMP(x);
    out[threadIdx.x] = x + acc[0] + acc[1] + acc[2] + acc[3] + y;
    if (threadIdx.x == 0) for (int i = 0; i + 1 < n; ++i) cyc[i] = t[i + 1] - t[i];
}
int main() {
    double* out; long* cyc;
    hipMalloc(&out, 8 * 64); hipMalloc(&cyc, 8 * 16);
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, out, cyc, 1.5);
    hipDeviceSynchronize();
    long h[16]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    const char* names[] = {"v_fma_f64 dep", "v_mul_f64 dep", "v_add_f64+v_rsq_f64", "v_add_f64+v_rcp_f64", "v_add_f32+v_rsq_f32", "2 readlane + mul(sgpr)",
                           "mfma f64 16x16x4 dep acc", "mfma -> v_mul -> mfma operand", "2 cndmask + mul", "cvt+rsq_f32+cvt+add_f64"};
    for (int i = 0; i < 10; ++i) printf("%-32s %6.1f cycles/iter\n", names[i], (double)h[i] / N);
    return 0;
}
