// The in-panel update kernel of leaf.hpp in isolation: panel_updk_kernel<RT> (K, N multiples of 32 / 128; RT = 4, 2, 1 row tiles per workgroup),
// C[m×N] −= P[m×K] · P[0:N, 0:K]ᵀ, checked against a host loop on small cases, then timed (isolated launches with an
// event pair each, and 50 launches back to back) for the shapes a C2 / C3 factorisation issues.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form tools/updk_bench.hip -o tools/bin/updk_bench
#include "../abstractgps.jl_amd/csrc/kcommon.hpp"
#include "../abstractgps.jl_amd/csrc/leaf.hpp"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
using namespace gpmi;
#define CK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("HIP error %s at %d: %s\n", #e, __LINE__, hipGetErrorString(r_)); return 1; } } while (0)

static int g_rt = 4;
static void run_updk(double* C, long ldc, const double* P, long ldp, long m, long N, long K) {
    const dim3 grid((unsigned)((m + 16 * g_rt - 1) / (16 * g_rt)), (unsigned)(N / 128));
    if (g_rt == 4) hipLaunchKernelGGL(panel_updk_kernel<4>, grid, dim3(256), 0, 0, C, ldc, P, ldp, (int)m, (int)K);
    else if (g_rt == 2) hipLaunchKernelGGL(panel_updk_kernel<2>, grid, dim3(256), 0, 0, C, ldc, P, ldp, (int)m, (int)K);
    else hipLaunchKernelGGL(panel_updk_kernel<1>, grid, dim3(256), 0, 0, C, ldc, P, ldp, (int)m, (int)K);
}

int main() {
    // one matrix: rows 0..mmax, columns [0, K) = P, columns [K, K+N) = C (as in the factorisation: the panel and the block right of it)
    const long mmax = 32768, KN = 2048, ld = KN + 32;
    std::vector<double> h((size_t)(mmax + 128) * ld);
    unsigned long st = 99;
    auto rnd = [&]() { st = st * 6364136223846793005ULL + 1442695040888963407ULL; return (double)((st >> 11) & 0xfffff) / 1048576.0 - 0.5; };
    for (auto& v : h) v = rnd();
    double *A, *B;
    CK(hipMalloc(&A, sizeof(double) * h.size())); CK(hipMalloc(&B, sizeof(double) * h.size()));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    int fail = 0;
    // --- correctness: host loop on row counts that are not multiples of 64, lower part only, every tile shape
    const long checks[][3] = {{328, 256, 256}, {1000, 128, 128}, {531, 512, 512}};
    for (g_rt = 1; g_rt <= 4; g_rt *= 2)
        for (auto& ck : checks) {
            const long m = ck[0], N = ck[1], K = ck[2];
            CK(hipMemcpy(A, h.data(), sizeof(double) * h.size(), hipMemcpyHostToDevice));
            run_updk(A + K, ld, A, ld, m, N, K);
            std::vector<double> o((size_t)(m + 1) * ld);
            CK(hipMemcpy(o.data(), A, sizeof(double) * o.size(), hipMemcpyDeviceToHost));
            double worst = 0;
            for (long r = 0; r < m; ++r)
                for (long c = 0; c < N && c <= r; ++c) {
                    double s = h[r * ld + K + c];
                    for (long k = 0; k < K; ++k) s -= h[r * ld + k] * h[c * ld + k];
                    worst = std::max(worst, std::fabs(s - o[r * ld + K + c]));
                }
            double untouched = 0;
            for (long c = 0; c < N; ++c) untouched = std::max(untouched, std::fabs(o[m * ld + K + c] - h[m * ld + K + c]));
            printf("check updk<%d> m=%ld K=N=%ld: max |err| (lower part) = %.3e, row m untouched: %.1e\n", g_rt, m, K, worst, untouched);
            if (!(worst < 1e-11) || untouched != 0) fail = 1;
        }
    // --- timing
    const long shapes[][3] = {{1024, 128, 128}, {4096, 128, 128}, {8192, 128, 128}, {16384, 128, 128}, {32768, 128, 128}, {4096, 256, 256}, {8192, 256, 256},
                              {16384, 256, 256}, {32768, 256, 256}, {4096, 512, 512}, {8192, 512, 512}, {16384, 512, 512}, {32768, 512, 512}, {8192, 1024, 1024}, {16384, 1024, 1024}};
    for (auto& sh : shapes) {
        const long m = sh[0], N = sh[1], K = sh[2];
        for (int which = 0; which < 3; ++which) {
            g_rt = which == 0 ? 4 : (which == 1 ? 2 : 1);
            float iso = 1e9f, b2b = 0;
            for (int rep = 0; rep < 6; ++rep) {
                float ms;
                CK(hipEventRecord(e0, 0));
                run_updk(A + K, ld, A, ld, m, N, K);
                CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
                CK(hipEventElapsedTime(&ms, e0, e1)); iso = std::min(iso, ms);
            }
            CK(hipEventRecord(e0, 0));
            for (int rep = 0; rep < 50; ++rep) { run_updk(A + K, ld, A, ld, m, N, K); }
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&b2b, e0, e1)); b2b /= 50;
            const double fl = 2.0 * (double)m * N * K;
            printf("%-7s m=%6ld N=%5ld K=%5ld  isolated %7.1f us  back-to-back %7.1f us  %6.1f TF/s (rectangle flops / back-to-back)\n", (g_rt == 4 ? "updk<4>" : (g_rt == 2 ? "updk<2>" : "updk<1>")), m, N, K,
                   iso * 1e3, b2b * 1e3, fl / (b2b * 1e-3) / 1e12);
        }
    }
    printf(fail ? "FAILED\n" : "ok\n");
    return fail;
}
