// What bounds kmat_kernel's 4.6 TB/s (round 5, the review's item 7): the store pattern, the store kind, or the VALU work in front of it?
// Fills the lower 128×128 tiles of an N×N fp64 matrix (row-major, ld = N + 32: the engine's layout) in four ways and times each:
//   0  kmat's pattern (wave w owns rows w, w+4, ..; one 1 KiB row segment per store instruction), plain global_store_dwordx4
//   1  the same with nontemporal stores (__builtin_nontemporal_store -> global_store ... nt)
//   2  pattern 0 with V dependent fp64 fma per element in front of the store (V = 16 / 32 / 48: brackets kmat's ≈ 35–45 VALU instructions
//      per element for D = 3 SE: 6 for the distance, the rest exp) — the point where the VALU time shows through the stores
//   3  whole rows: workgroup b writes rows of the lower triangle as long contiguous bursts (one wave = 1 KiB pieces of ONE row)
// and hipMemsetAsync over the same bytes for scale.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/kmat_probe.hip -o tools/bin/kmat_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("HIP error %s at %d: %s\n", #e, __LINE__, hipGetErrorString(r_)); return 1; } } while (0)
typedef double d2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void tile_of(int b, int& bi, int& bj) {  // lower-triangle enumeration, row by row
    int i = (int)((-1.0 + sqrt(1.0 + 8.0 * (double)b)) * 0.5);
    while ((long)i * (i + 1) / 2 > b) --i;
    while ((long)(i + 1) * (i + 2) / 2 <= b) ++i;
    bi = i;
    bj = b - (int)((long)i * (i + 1) / 2);
}

template <int MODE, int V>
__global__ __launch_bounds__(256) void fill_tiles(double* __restrict__ out, long ld, double seed) {
    int bi, bj;
    tile_of((int)blockIdx.x, bi, bj);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const long m0 = (long)bi * 128, n0 = (long)bj * 128;
    double a = seed + lane * 1e-3, b = seed - w * 1e-3;
#pragma unroll 4
    for (int rr = 0; rr < 32; ++rr) {
        d2_t o;
        o.x = a + rr;
        o.y = b - rr;
        if (MODE == 2) {
#pragma unroll
            for (int v = 0; v < V; ++v) {
                o.x = fma(o.x, 0.999999, 1e-7);
                o.y = fma(o.y, 0.999999, 1e-7);
            }
        }
        d2_t* p = reinterpret_cast<d2_t*>(out + (m0 + w + 4 * rr) * ld + n0 + 2 * lane);
        if (MODE == 1) __builtin_nontemporal_store(o, p);
        else *p = o;
    }
}


// other tile shapes with the same 16 384 elements per workgroup: TR rows × TC columns (TC a multiple of 128); wave w owns rows w, w+4, ..; a row of
// the tile is TC/128 store instructions of 1 KiB each (contiguous) — fewer distinct rows (DRAM pages, TLB entries) per workgroup, longer bursts
template <int TR, int TC, int V>
__global__ __launch_bounds__(256) void fill_shape(double* __restrict__ out, long ld, long n, double seed) {
    const long tcols = n / TC;
    // lower trapezoid in units of this shape: tile (bi, bj) is written when its first column is <= its last row
    const long bi = blockIdx.y, bj = blockIdx.x;
    const long m0 = bi * TR, n0 = bj * TC;
    if (n0 > m0 + TR - 1 || bj >= tcols) return;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    double a = seed + lane * 1e-3, b = seed - w * 1e-3;
    for (int rr = 0; rr < TR / 4; ++rr) {
#pragma unroll
        for (int cs = 0; cs < TC / 128; ++cs) {
            d2_t o;
            o.x = a + rr + cs;
            o.y = b - rr - cs;
#pragma unroll
            for (int v = 0; v < V; ++v) {
                o.x = fma(o.x, 0.999999, 1e-7);
                o.y = fma(o.y, 0.999999, 1e-7);
            }
            *reinterpret_cast<d2_t*>(out + (m0 + w + 4 * rr) * ld + n0 + 128 * cs + 2 * lane) = o;
        }
    }
}

// whole rows of the lower triangle: row r has r + 1 elements, rounded up to its 128-column tile boundary; one workgroup per 4 rows
__global__ __launch_bounds__(256) void fill_rows(double* __restrict__ out, long ld, long n, double seed, int nt) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const long r = (long)blockIdx.x * 4 + w;
    if (r >= n) return;
    const long len = (r / 128 + 1) * 128;
    d2_t o;
    o.x = seed + lane;
    o.y = seed - lane;
    for (long c = 2 * lane; c < len; c += 128) {
        d2_t* p = reinterpret_cast<d2_t*>(out + r * ld + c);
        if (nt) __builtin_nontemporal_store(o, p);
        else *p = o;
    }
}

int main(int argc, char** argv) {
    const long n = argc > 1 ? atol(argv[1]) : 32768;
    const long ld = n + 32;
    const long tiles = (n / 128) * (n / 128 + 1) / 2;
    const double bytes = (double)tiles * 128 * 128 * 8;
    double launch_bytes = 0;  // != 0: bytes of the shape being timed
    double* A;
    CK(hipMalloc(&A, sizeof(double) * (size_t)n * ld));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto time_it = [&](const char* name, auto launch) {
        launch();
        (void)hipDeviceSynchronize();
        float best = 1e30f;
        for (int rep = 0; rep < 5; ++rep) {
            (void)hipEventRecord(e0, 0);
            launch();
            (void)hipEventRecord(e1, 0);
            (void)hipEventSynchronize(e1);
            float ms;
            (void)hipEventElapsedTime(&ms, e0, e1);
            best = ms < best ? ms : best;
        }
        const double bb = launch_bytes > 0 ? launch_bytes : bytes;
        printf("{\"n\": %ld, \"case\": \"%s\", \"ms\": %.4f, \"GBps\": %.1f}\n", n, name, best, bb / (best * 1e-3) / 1e9);
    };
    const dim3 g((unsigned)tiles), blk(256);
    time_it("tiles_plain", [&] { hipLaunchKernelGGL((fill_tiles<0, 0>), g, blk, 0, 0, A, ld, 1.0); });
    time_it("tiles_nontemporal", [&] { hipLaunchKernelGGL((fill_tiles<1, 0>), g, blk, 0, 0, A, ld, 1.0); });
    time_it("tiles_plain_valu16", [&] { hipLaunchKernelGGL((fill_tiles<2, 16>), g, blk, 0, 0, A, ld, 1.0); });
    time_it("tiles_plain_valu32", [&] { hipLaunchKernelGGL((fill_tiles<2, 32>), g, blk, 0, 0, A, ld, 1.0); });
    time_it("tiles_plain_valu48", [&] { hipLaunchKernelGGL((fill_tiles<2, 48>), g, blk, 0, 0, A, ld, 1.0); });
    {
        auto shape = [&](const char* name, auto kern, int TR, int TC) {
            const dim3 gs((unsigned)(n / TC), (unsigned)(n / TR));
            // bytes actually written by this shape (tiles crossing the diagonal are written whole)
            double by = 0;
            for (long bi = 0; bi < n / TR; ++bi) by += (double)((bi * TR + TR - 1) / TC + 1) * TR * TC * 8.0;
            launch_bytes = by;
            time_it(name, [&] { hipLaunchKernelGGL(kern, gs, blk, 0, 0, A, ld, n, 1.0); });
            launch_bytes = 0;
        };
        shape("shape_64x256_valu32", fill_shape<64, 256, 32>, 64, 256);
        shape("shape_32x512_valu32", fill_shape<32, 512, 32>, 32, 512);
        shape("shape_16x1024_valu32", fill_shape<16, 1024, 32>, 16, 1024);
        shape("shape_32x512_valu0", fill_shape<32, 512, 0>, 32, 512);
    }
    time_it("rows_plain", [&] { hipLaunchKernelGGL(fill_rows, dim3((unsigned)((n + 3) / 4)), blk, 0, 0, A, ld, n, 1.0, 0); });
    time_it("rows_nontemporal", [&] { hipLaunchKernelGGL(fill_rows, dim3((unsigned)((n + 3) / 4)), blk, 0, 0, A, ld, n, 1.0, 1); });
    {
        // hipMemsetAsync over the same number of bytes (contiguous)
        const size_t nb = (size_t)bytes;
        time_it("hipMemsetAsync_same_bytes", [&] { (void)hipMemsetAsync(A, 0, nb, 0); });
    }
    (void)hipFree(A);
    return 0;
}
