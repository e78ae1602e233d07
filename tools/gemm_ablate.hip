// Timing-only ablation of the MFMA trailing-update kernel (gemm_nt_dma): build with -DGPMI_ABL=<mask> (fp32 instantiation: 8 no
// epilogue stores, 16 no operand DMA inside the k loop, 32 no wait + barrier at the end of a step) to see which part of the loop
// the MFMA pipe waits for.  Results are wrong by construction for mask != 0.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DGPMI_ABL=16 tools/gemm_ablate.hip -o /tmp/abl16
// (Masks 1 / 2 / 4 belonged to the register-staged predecessor of the kernel, removed in round 3: history at dbd752d.)
#include "../abstractgps.jl_amd/csrc/kernels.hpp"
#include <cstdio>
#include <vector>
#define GPMI_DMA 1
#include <cstdlib>
using namespace gpmi;
int main(int argc, char** argv) {
    const long M = argc > 1 ? atol(argv[1]) : 32768, K = argc > 2 ? atol(argv[2]) : 2048;
    const long lda = K + 32, ldc = M + 32;
    double *A, *C;
    hipMalloc(&A, sizeof(double) * (M + 128) * lda);
    hipMalloc(&C, sizeof(double) * (M + 128) * ldc);
    hipMemset(A, 0, sizeof(double) * (M + 128) * lda);
    hipMemset(C, 0, sizeof(double) * (M + 128) * ldc);
    // non-trivial data (DVFS depends on operand toggling)
    {
        std::vector<double> h((size_t)(M + 128) * lda);
        for (size_t i = 0; i < h.size(); ++i) h[i] = (double)((i * 2654435761u) % 2001) / 1000.0 - 1.0;
        hipMemcpy(A, h.data(), sizeof(double) * h.size(), hipMemcpyHostToDevice);
    }
    GridMap g{};
    g.lower = 1; g.P = 1; g.Q = 1; g.nb = 128; g.compact = 1;
    const long tm = M / 128;
    g.tn = (int)tm; g.dt = 0; g.tm = (int)tm;
    const long total = tm * (tm + 1) / 2;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((gemm_nt_dma_kernel<double, double>), dim3((unsigned)total), dim3(256), 0, 0, C, ldc, A, lda, A, lda,
                           (int)M, (int)M, (int)K, g);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double fl = 2.0 * K * (double)total * 128 * 128;
        printf("DMA=%d ABL=%d M=%ld K=%ld rep %d: %.3f ms  %.2f TF/s\n", GPMI_DMA, GPMI_ABL, M, K, rep, ms, fl / ms / 1e9);
    }
    return 0;
}
