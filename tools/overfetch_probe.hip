// The review's item 6, decided by measurement: the trailing update reads 7.4x its algorithmic bytes from the fabric (operand panels re-read by every
// XCD's L2: 22.6 GB of FETCH + WRITE per launch against 3.03 GB, ~2.5 TB/s while it runs).  On one GPU that is harmless — does it stay harmless when
// the same fabric carries the panel exchange of a multi-device fit?  The exchange of one block step is emulated on ONE GPU by what it costs this
// device's memory system: a stream of device-to-device copies of panel size (an inbound peer write lands in HBM through the same fabric / memory
// controllers; the outbound read likewise), sized 2-4 GB per bulk-update launch, next to the update kernel itself.
//   (1) update alone  (2) copy stream alone  (3) both, copy stream on a separate lowest-priority stream: slowdown of each.
// Shapes: the C4 trailing update after the first panels of an 8x1 grid is 1/8 of the rows: M = 7 168 local rows x N = 57 344, K = 1 024 (NB); here
// the single-GPU launch of the bench (lower trapezoid, M = N = 49 152, K = 2 048) and that rank-sized one.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/overfetch_probe.hip -o tools/bin/overfetch_probe
#include "../abstractgps.jl_amd/csrc/kernels.hpp"
#include <cstdio>
#include <cstdlib>
#include <vector>
using namespace gpmi;
#define CK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("HIP error %s at %d: %s\n", #e, __LINE__, hipGetErrorString(r_)); return 1; } } while (0)

int main() {
    const long MMAX = 57344 + 128, KMAX = 2048;
    const long lda = KMAX + 32, ldc = MMAX + 32;
    double *A, *C, *S, *D;
    const size_t copy_bytes = (size_t)512 << 20;  // one "panel" message: 512 MB (C4 step 0 ships 512 MB of panel in all)
    CK(hipMalloc(&A, sizeof(double) * (size_t)MMAX * lda));
    CK(hipMalloc(&C, sizeof(double) * (size_t)(49152 + 128) * ldc));
    CK(hipMalloc(&S, copy_bytes));
    CK(hipMalloc(&D, copy_bytes));
    {
        std::vector<double> h((size_t)MMAX * lda);
        for (size_t i = 0; i < h.size(); ++i) h[i] = (double)((i * 2654435761u) % 2001) / 1000.0 - 1.0;
        CK(hipMemcpy(A, h.data(), sizeof(double) * h.size(), hipMemcpyHostToDevice));
    }
    CK(hipMemset(C, 0, sizeof(double) * (size_t)(49152 + 128) * ldc));
    CK(hipMemset(S, 1, copy_bytes));
    hipStream_t sg, sc;
    int lo, hi;
    CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    CK(hipStreamCreateWithPriority(&sg, hipStreamNonBlocking, hi));
    CK(hipStreamCreateWithPriority(&sc, hipStreamNonBlocking, lo));
    hipEvent_t g0, g1, c0, c1;
    CK(hipEventCreate(&g0)); CK(hipEventCreate(&g1)); CK(hipEventCreate(&c0)); CK(hipEventCreate(&c1));

    struct Shape { long M, N, K; int lower; const char* what; };
    const Shape shapes[] = {{49152, 49152, 2048, 1, "single-GPU trailing update of the bench (lower trapezoid)"},
                            {7168, 57344, 1024, 0, "one rank's share on an 8x1 grid, NB = 1024 (rectangle)"}};
    for (const Shape& sh : shapes) {
        GridMap g{};
        g.P = 1; g.Q = 1; g.nb = 128; g.nbatch = 1;
        dim3 grid;
        double flops;
        if (sh.lower) {
            g.lower = 1; g.compact = 1;
            const long tm = sh.M / 128;
            g.tn = (int)tm; g.dt = 0; g.tm = (int)tm;
            grid = dim3((unsigned)(tm * (tm + 1) / 2));
            flops = 2.0 * sh.K * (double)(tm * (tm + 1) / 2) * 128 * 128;
        } else {
            grid = dim3((unsigned)(sh.N / 128), (unsigned)(sh.M / 128));
            g.tn = (int)(sh.N / 128); g.tm = (int)(sh.M / 128);
            flops = 2.0 * sh.K * (double)sh.M * (double)sh.N;
        }
        auto gemm = [&]() {
            hipLaunchKernelGGL((gemm_nt_dma_kernel<double, double, 1>), grid, dim3(256), 0, sg, C, ldc, A, lda, A, lda, (int)sh.M, (int)sh.N, (int)sh.K, g);
        };
        auto copies = [&](int n) {
            for (int i = 0; i < n; ++i) (void)hipMemcpyAsync(D, S, copy_bytes, hipMemcpyDeviceToDevice, sc);
        };
        // warm
        for (int i = 0; i < 3; ++i) gemm();
        copies(2);
        CK(hipDeviceSynchronize());
        // (1) update alone
        float t_g = 0, t_c = 0, t_g2 = 0, t_c2 = 0;
        CK(hipEventRecord(g0, sg));
        for (int i = 0; i < 4; ++i) gemm();
        CK(hipEventRecord(g1, sg));
        CK(hipEventSynchronize(g1));
        CK(hipEventElapsedTime(&t_g, g0, g1));
        t_g /= 4;
        // (2) copies alone: as many 512 MB messages as fit the update's duration at 2, 4, 8 GB per update launch
        for (int per : {4, 8, 16}) {  // messages per update launch: 2 / 4 / 8 GB
            CK(hipEventRecord(c0, sc));
            copies(per * 4);
            CK(hipEventRecord(c1, sc));
            CK(hipEventSynchronize(c1));
            CK(hipEventElapsedTime(&t_c, c0, c1));
            t_c /= 4;
            // (3) both
            CK(hipEventRecord(g0, sg));
            CK(hipEventRecord(c0, sc));
            for (int i = 0; i < 4; ++i) gemm();
            copies(per * 4);
            CK(hipEventRecord(g1, sg));
            CK(hipEventRecord(c1, sc));
            CK(hipEventSynchronize(g1));
            CK(hipEventSynchronize(c1));
            CK(hipEventElapsedTime(&t_g2, g0, g1));
            CK(hipEventElapsedTime(&t_c2, c0, c1));
            t_g2 /= 4; t_c2 /= 4;
            printf("{\"shape\": \"%ldx%ldx%ld lower=%d\", \"what\": \"%s\", \"copy_GB_per_update\": %.1f, \"update_alone_ms\": %.3f, \"update_alone_tflops\": %.2f, "
                   "\"copies_alone_ms\": %.3f, \"copies_alone_GBps\": %.0f, \"update_beside_copies_ms\": %.3f, \"update_slowdown\": %.4f, "
                   "\"copies_beside_update_ms\": %.3f, \"copies_beside_GBps\": %.0f}\n",
                   sh.M, sh.N, sh.K, sh.lower, sh.what, per * 0.5, t_g, flops / t_g / 1e9, t_c, per * 0.512 * 1.048576 / t_c * 1e3, t_g2, t_g2 / t_g, t_c2,
                   per * 0.512 * 1.048576 / t_c2 * 1e3);
            fflush(stdout);
        }
    }
    return 0;
}
