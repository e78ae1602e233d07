// Can the panel leaf chain run beside the trailing-update GEMM if the GEMM leaves room (1 workgroup per CU) and the leaf's
// LDS footprint is reduced (64-row X slabs: 75 KB)?  Times: leaf chain alone, GEMM alone (1 and 2 WG/CU), both together.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/overlap_probe.hip -o tools/overlap.bin
#include "../abstractgps.jl_amd/csrc/kernels.hpp"
#include <cstdio>
#include <cstdlib>
#include <vector>
using namespace gpmi;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
int main(int argc, char** argv) {
    const long M = argc > 1 ? atol(argv[1]) : 16384, K = 2048, NL = argc > 2 ? atol(argv[2]) : 128;
    const int XR = 64;
    const long lda = K + 32, ldc = M + 32, ldp = 64 + 32;
    double *A, *C, *P, *logdet;
    int *info, *ticket;
    CK(hipMalloc(&A, sizeof(double) * (M + 128) * lda));
    CK(hipMalloc(&C, sizeof(double) * (M + 128) * ldc));
    CK(hipMalloc(&P, sizeof(double) * (M + 256) * ldp));
    CK(hipMalloc(&logdet, 8)); CK(hipMalloc(&info, 4)); CK(hipMalloc(&ticket, 256));
    CK(hipMemset(ticket, 0, 256)); CK(hipMemset(info, 0, 4)); CK(hipMemset(logdet, 0, 8));
    CK(hipMemset(C, 0, sizeof(double) * (M + 128) * ldc));
    {
        std::vector<double> h((size_t)(M + 128) * lda);
        for (size_t i = 0; i < h.size(); ++i) h[i] = (double)((i * 2654435761u) % 2001) / 1000.0 - 1.0;
        CK(hipMemcpy(A, h.data(), sizeof(double) * h.size(), hipMemcpyHostToDevice));
        std::vector<double> p((size_t)(M + 256) * ldp, 0.0);
        for (long r = 0; r < M + 64; ++r)
            for (long c = 0; c < 64; ++c) p[r * ldp + c] = (r == c ? 3.0 : 0.01 * (double)((r * 7 + c * 13) % 17) / 17.0);
        CK(hipMemcpy(P, p.data(), sizeof(double) * p.size(), hipMemcpyHostToDevice));
    }
    GridMap g{};
    g.lower = 1; g.P = 1; g.Q = 1; g.nb = 128; g.compact = 1; g.nbatch = 1;
    const long tm = M / 128;
    g.tn = (int)tm; g.dt = 0; g.tm = (int)tm;
    const long total = tm * (tm + 1) / 2;
    hipStream_t s1, s2;
    int lo, hi;
    CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    CK(hipStreamCreateWithPriority(&s2, hipStreamNonBlocking, hi));
    hipEvent_t e0, e1, f0, f1;
    hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreate(&f0); hipEventCreate(&f1);
    CK(hipFuncSetAttribute((const void*)gemm_nt_dma_kernel<double, double>, hipFuncAttributeMaxDynamicSharedMemorySize, 20480));
    auto gemm = [&](size_t dyn) {
        hipLaunchKernelGGL((gemm_nt_dma_kernel<double, double>), dim3((unsigned)total), dim3(256), dyn, s1, C, ldc, A, lda, A, lda, (int)M,
                           (int)M, (int)K, g);
    };
    auto chain = [&]() {
        for (int i = 0; i < NL; ++i)  // the same (already positive definite) tile again and again: timing only
            hipLaunchKernelGGL((panel64_kernel<double, XR>), dim3((unsigned)(M / XR)), dim3(256), 0, s2, P, ldp, (int)M, info, 0, 64, logdet,
                               ticket, 0);
    };
    float ms;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(f0, s2); chain(); hipEventRecord(f1, s2); hipEventSynchronize(f1);
        hipEventElapsedTime(&ms, f0, f1);
        printf("leaf chain alone (%ld leaves, M=%ld): %.3f ms = %.1f us/leaf\n", NL, M, ms, ms * 1e3 / NL);
        for (size_t dyn : {(size_t)0, (size_t)20480}) {
            hipEventRecord(e0, s1); gemm(dyn); hipEventRecord(e1, s1); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
            printf("gemm alone, dyn LDS %zu (%s WG/CU): %.3f ms\n", dyn, dyn ? "1" : "2", ms);
        }
        for (size_t dyn : {(size_t)0, (size_t)20480}) {
            hipEventRecord(e0, s1); gemm(dyn); hipEventRecord(e1, s1);
            hipEventRecord(f0, s2); chain(); hipEventRecord(f1, s2);
            hipEventSynchronize(e1); hipEventSynchronize(f1);
            float mg, mc;
            hipEventElapsedTime(&mg, e0, e1); hipEventElapsedTime(&mc, f0, f1);
            printf("together, gemm dyn LDS %zu: gemm %.3f ms, leaf chain %.3f ms (%.1f us/leaf)\n", dyn, mg, mc, mc * 1e3 / NL);
        }
    }
    return 0;
}
