// Cycle stamps inside one panel64v2 launch (workgroup 0; wave 0 = the diagonal chain, wave 1 = an owner wave).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form -DGPMI_PANEL_STAMPS tools/leaf_stamps.hip -o tools/bin/leaf_stamps
// Stamp order: 0 entry | 1 after the pre-update | per step j: top, before B1, after B1, before B2 | before the final barrier | exit
#include "../abstractgps.jl_amd/csrc/leaf.hpp"
#include <cstdio>
#include <cstdlib>
#include <vector>
using namespace gpmi;
int main(int argc, char** argv) {
    const long M = argc > 1 ? atol(argv[1]) : 16384;
    const int kpre = argc > 2 ? atoi(argv[2]) : 0;
    const int xr = argc > 3 ? atoi(argv[3]) : 128;
    const int nc = argc > 4 ? atoi(argv[4]) : 4;
    const long tw = 16 * nc;
    const long cols = 64 * kpre + tw, ldp = cols + 32;
    double *P, *logdet;
    int *info, *ticket;
    hipMalloc(&P, sizeof(double) * (M + 448) * ldp);
    hipMalloc(&logdet, 8 * 128); hipMalloc(&info, 4); hipMalloc(&ticket, 256);
    hipMemset(ticket, 0, 256); hipMemset(info, 0, 4); hipMemset(logdet, 0, 8 * 128);
    std::vector<double> p((size_t)(M + 448) * ldp, 0.0);
    for (long r = 0; r < M + tw; ++r)
        for (long c = 0; c < cols; ++c) p[r * ldp + c] = 0.01 * (double)((r * 7 + c * 13) % 17) / 17.0;
    for (long r = 0; r < tw; ++r) p[r * ldp + 64 * kpre + r] = 3.0;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipMemcpy(P, p.data(), sizeof(double) * p.size(), hipMemcpyHostToDevice);
        hipEventRecord(e0, 0);
        if (nc == 8 && xr == 64)
            hipLaunchKernelGGL((panel64v2_kernel<64, 8>), dim3((unsigned)std::max(1L, (M + 63) / 64)), dim3(256), 0, 0, P, ldp, (int)M, info, 0, 128, logdet, ticket, 0);
        else if (nc == 8)
            hipLaunchKernelGGL((panel64v2_kernel<128, 8>), dim3((unsigned)std::max(1L, (M + 127) / 128)), dim3(256), 0, 0, P, ldp, (int)M, info, 0, 128, logdet, ticket, 0);
        else if (xr == 64)
            hipLaunchKernelGGL(panel64v2_kernel<64>, dim3((unsigned)std::max(1L, (M + 63) / 64)), dim3(256), 0, 0, P + 64 * kpre, ldp, (int)M, info, 0, 64, logdet, ticket, kpre);
        else
            hipLaunchKernelGGL(panel64v2_kernel<128>, dim3((unsigned)std::max(1L, (M + 127) / 128)), dim3(256), 0, 0, P + 64 * kpre, ldp, (int)M, info, 0, 64, logdet, ticket, kpre);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        long st[56];
        hipMemcpy(st, (long*)logdet + 8, sizeof(st), hipMemcpyDeviceToHost);
        printf("M=%ld kpre=%d XR=%d NC=%d rep %d: %.1f us\n  wave0:", M, kpre, xr, nc, rep, ms * 1e3);
        for (int i = 0; i < 20; ++i) printf(" %ld", st[i]);
        printf("\n  wave1:");
        for (int i = 0; i < 20; ++i) printf(" %ld", st[24 + i]);
        printf("\n");
    }
    return 0;
}
