// Cycle stamps inside one panel64 launch (workgroup 0, thread 0): where the 29 µs of a leaf go.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DGPMI_PANEL_STAMPS tools/panel_stamps.hip -o tools/stamps.bin
#include "../abstractgps.jl_amd/csrc/kernels.hpp"
#include <cstdio>
#include <cstdlib>
#include <vector>
using namespace gpmi;
int main(int argc, char** argv) {
    const long M = argc > 1 ? atol(argv[1]) : 16384;
    const long ldp = 96;
    double *P, *logdet;
    int *info, *ticket;
    hipMalloc(&P, sizeof(double) * (M + 256) * ldp);
    hipMalloc(&logdet, 8 * 32); hipMalloc(&info, 4); hipMalloc(&ticket, 256);
    hipMemset(ticket, 0, 256); hipMemset(info, 0, 4); hipMemset(logdet, 0, 8 * 32);
    std::vector<double> p((size_t)(M + 256) * ldp, 0.0);
    for (long r = 0; r < M + 64; ++r)
        for (long c = 0; c < 64; ++c) p[r * ldp + c] = (r == c ? 3.0 : 0.01 * (double)((r * 7 + c * 13) % 17) / 17.0);
    hipMemcpy(P, p.data(), sizeof(double) * p.size(), hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipMemcpy(P, p.data(), sizeof(double) * p.size(), hipMemcpyHostToDevice);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((panel64_kernel<double, 128>), dim3((unsigned)((M + 127) / 128 > 0 ? (M + 127) / 128 : 1)), dim3(256), 0, 0, P, ldp,
                           (int)M, info, 0, 64, logdet, ticket, 0);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        long st[24];
        hipMemcpy(st, logdet, sizeof(st), hipMemcpyDeviceToHost);
        printf("M=%ld rep %d: %.1f us; stamps (cycles since start):", M, rep, ms * 1e3);
        for (int i = 8; i < 24; ++i) printf(" %ld", st[i]);
        printf("\n");
    }
    return 0;
}
