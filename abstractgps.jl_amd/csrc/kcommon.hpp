// kcommon.hpp — what every device translation unit shares: vector typedefs, the MFMA traits (lane maps), 1/√x and the lane broadcast.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gpmi {

typedef double d2_t __attribute__((ext_vector_type(2)));
typedef double d4_t __attribute__((ext_vector_type(4)));
typedef float f2_t __attribute__((ext_vector_type(2)));
typedef float f4_t __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------------
// MFMA traits.  A operand: lane l supplies A[i = l&15][k = l>>4]; B operand: B[k = l>>4][j = l&15]
// (same for f64 16x16x4 and f32 16x16x4).  C/D: col = l&15 for both; row differs:
//   f64: row = (l>>4) + 4*r      f32: row = 4*(l>>4) + r          (cdna_hip_programming.md §3)
// ------------------------------------------------------------------------------------------------
template <typename T> struct Tr;
template <> struct Tr<double> {
    typedef d2_t chunk_t;  // 16 B = 2 k-values
    typedef d2_t pair_t;   // 2 output columns per lane in kmat
    typedef d4_t acc_t;
    static constexpr int VEC = 2;
    __device__ static inline acc_t mfma(double a, double b, acc_t c) {
        return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    }
    __device__ static inline int crow(int lane, int r) { return (lane >> 4) + 4 * r; }
};
template <> struct Tr<float> {
    typedef f4_t chunk_t;  // 16 B = 4 k-values
    typedef f2_t pair_t;
    typedef f4_t acc_t;
    static constexpr int VEC = 4;
    __device__ static inline acc_t mfma(float a, float b, acc_t c) {
        return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    }
    __device__ static inline int crow(int lane, int r) { return 4 * (lane >> 4) + r; }
};

// 1/√x at full precision (v_rsq estimate + Newton) and a lane broadcast by v_readlane — the serial chain of the tile factorisation
template <typename T> __device__ __forceinline__ T fast_rsqrt(T x);
template <> __device__ __forceinline__ double fast_rsqrt<double>(double x) {
    double r = __builtin_amdgcn_rsq(x);       // v_rsq_f64 estimate
    r = r * fma(-0.5 * x * r, r, 1.5);        // two Newton steps -> full fp64 precision
    r = r * fma(-0.5 * x * r, r, 1.5);
    return r;
}
template <> __device__ __forceinline__ float fast_rsqrt<float>(float x) {
    float r = __builtin_amdgcn_rsqf(x);
    r = r * fmaf(-0.5f * x * r, r, 1.5f);
    return r;
}

template <typename T> __device__ __forceinline__ T lane_bcast(T v, int srclane);  // srclane must be a compile-time constant
template <> __device__ __forceinline__ double lane_bcast<double>(double v, int srclane) {
    const long bits = __builtin_bit_cast(long, v);
    const int lo = __builtin_amdgcn_readlane((int)bits, srclane);
    const int hi = __builtin_amdgcn_readlane((int)(bits >> 32), srclane);
    return __builtin_bit_cast(double, ((long)hi << 32) | (long)(unsigned)lo);
}
template <> __device__ __forceinline__ float lane_bcast<float>(float v, int srclane) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), srclane));
}

}  // namespace gpmi
