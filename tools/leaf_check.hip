// The register-resident leaf (panel64v2_kernel) against the round-3 leaf (panel64_kernel) on the same random SPD tile + X rows:
// max |difference| of the factor tile, of X L^-T, of Σ log L_ii, and the launch times of both (isolated launches, M rows under the tile).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form tools/leaf_check.hip -o tools/bin/leaf_check;  tools/bin/leaf_check [M ...]
#include "../abstractgps.jl_amd/csrc/kernels.hpp"
#include "../abstractgps.jl_amd/csrc/leaf.hpp"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
using namespace gpmi;
#define CK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("HIP error %s at %d: %s\n", #e, __LINE__, hipGetErrorString(r_)); return 1; } } while (0)

template <int XR> static void launch_v2(double* P, long ld, long M, int* info, double* logdet, int* ticket, int kpre) {
    const unsigned nb = (unsigned)std::max(1L, (M + XR - 1) / XR);
    hipLaunchKernelGGL(panel64v2_kernel<XR>, dim3(nb), dim3(256), 0, 0, P, ld, (int)M, info, 0, 64, logdet, ticket, kpre);
}
static void launch_v1(double* P, long ld, long M, int* info, double* logdet, int* ticket, int kpre) {
    const unsigned nb = (unsigned)std::max(1L, (M + 127) / 128);
    hipLaunchKernelGGL((panel64_kernel<double, 128>), dim3(nb), dim3(256), 0, 0, P, ld, (int)M, info, 0, 64, logdet, ticket, kpre);
}

int main(int argc, char** argv) {
    std::vector<long> Ms;
    for (int i = 1; i < argc; ++i) Ms.push_back(atol(argv[i]));
    if (Ms.empty()) Ms = {0, 16, 64, 128, 208, 1024, 16384, 65536};
    int worst_fail = 0;
    for (long M : Ms)
        for (int kpre = 0; kpre <= 2; ++kpre) {
            const long cols = 64 * (kpre + 1), ld = cols + 32, rows = M + 64 + 256;
            std::vector<double> h((size_t)rows * ld, 0.0);
            // left tiles: a plausible already-final panel (small entries); the current tile: SPD after the pre-update as well
            unsigned long st = 12345 + 977 * (unsigned long)M + kpre;
            auto rnd = [&]() { st = st * 6364136223846793005ULL + 1442695040888963407ULL; return (double)((st >> 11) & 0xfffff) / 1048576.0 - 0.5; };
            for (long r = 0; r < M + 64; ++r)
                for (long c = 0; c < cols; ++c) h[r * ld + c] = 0.3 * rnd();
            for (long r = 0; r < 64; ++r)
                for (long c = 0; c < 64; ++c) {
                    double v = 0;  // G Gᵀ-like symmetric part built from a smooth kernel + strong diagonal
                    v = exp(-0.05 * (double)((r - c) * (r - c))) + (r == c ? 6.0 + 2.0 * kpre : 0.0);
                    h[r * ld + 64 * kpre + c] = v;
                }
            double *P1, *P2, *P3, *ldv;
            int *info, *ticket;
            CK(hipMalloc(&P1, sizeof(double) * h.size())); CK(hipMalloc(&P2, sizeof(double) * h.size())); CK(hipMalloc(&P3, sizeof(double) * h.size()));
            CK(hipMalloc(&ldv, 8 * 64)); CK(hipMalloc(&info, 16)); CK(hipMalloc(&ticket, 256));
            CK(hipMemset(ticket, 0, 256)); CK(hipMemset(info, 0, 16)); CK(hipMemset(ldv, 0, 8 * 64));
            hipEvent_t e0, e1;
            CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            float t1 = 1e9f, t2 = 1e9f, t3 = 1e9f;
            for (int rep = 0; rep < 4; ++rep) {
                float ms;
                CK(hipMemcpy(P1, h.data(), sizeof(double) * h.size(), hipMemcpyHostToDevice));
                CK(hipMemcpy(P2, h.data(), sizeof(double) * h.size(), hipMemcpyHostToDevice));
                CK(hipMemcpy(P3, h.data(), sizeof(double) * h.size(), hipMemcpyHostToDevice));
                CK(hipMemset(ldv, 0, 8 * 64));
                CK(hipEventRecord(e0, 0)); launch_v1(P1 + 64 * kpre, ld, M, info, ldv + 0, ticket, kpre); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
                CK(hipEventElapsedTime(&ms, e0, e1)); t1 = std::min(t1, ms);
                CK(hipEventRecord(e0, 0)); launch_v2<128>(P2 + 64 * kpre, ld, M, info + 1, ldv + 1, ticket + 8, kpre); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
                CK(hipEventElapsedTime(&ms, e0, e1)); t2 = std::min(t2, ms);
                CK(hipEventRecord(e0, 0)); launch_v2<64>(P3 + 64 * kpre, ld, M, info + 2, ldv + 2, ticket + 16, kpre); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
                CK(hipEventElapsedTime(&ms, e0, e1)); t3 = std::min(t3, ms);
                CK(hipGetLastError());
            }
            std::vector<double> a(h.size()), b(h.size()), c3(h.size());
            double lds[3]; int infos[3];
            CK(hipMemcpy(a.data(), P1, sizeof(double) * h.size(), hipMemcpyDeviceToHost));
            CK(hipMemcpy(b.data(), P2, sizeof(double) * h.size(), hipMemcpyDeviceToHost));
            CK(hipMemcpy(c3.data(), P3, sizeof(double) * h.size(), hipMemcpyDeviceToHost));
            CK(hipMemcpy(lds, ldv, sizeof(lds), hipMemcpyDeviceToHost)); CK(hipMemcpy(infos, info, sizeof(infos), hipMemcpyDeviceToHost));
            double eL = 0, eX = 0, eL3 = 0, eX3 = 0, eOther = 0;
            for (long r = 0; r < rows; ++r)
                for (long c = 0; c < ld; ++c) {
                    const bool in_tile_cols = c >= 64 * kpre && c < 64 * kpre + 64;
                    const double d2 = fabs(a[r * ld + c] - b[r * ld + c]), d3 = fabs(a[r * ld + c] - c3[r * ld + c]);
                    if (in_tile_cols && r < 64 && c - 64 * kpre <= r) { eL = fmax(eL, d2); eL3 = fmax(eL3, d3); }
                    else if (in_tile_cols && r >= 64 && r < 64 + M) { eX = fmax(eX, d2); eX3 = fmax(eX3, d3); }
                    else if (!(in_tile_cols && r < 64)) eOther = fmax(eOther, fmax(d2, d3));  // nothing else may change (upper part of the tile: unspecified)
                }
            const bool ok = eL < 1e-12 && eX < 1e-12 && eL3 < 1e-12 && eX3 < 1e-12 && eOther == 0 && fabs(lds[0] - lds[1]) < 1e-11 && fabs(lds[0] - lds[2]) < 1e-11 &&
                            infos[0] == infos[1] && infos[0] == infos[2] && eL == eL && eX == eX;
            if (!ok) worst_fail = 1;
            printf("M=%6ld kpre=%d: v1 %7.1f us  v2<128> %7.1f us  v2<64> %7.1f us | dL %.1e %.1e  dX %.1e %.1e  other %.1e  logdet %.12g %.12g %.12g info %d %d %d  %s\n",
                   M, kpre, t1 * 1e3, t2 * 1e3, t3 * 1e3, eL, eL3, eX, eX3, eOther, lds[0], lds[1], lds[2], infos[0], infos[1], infos[2], ok ? "OK" : "MISMATCH");
            fflush(stdout);
            (void)hipFree(P1); (void)hipFree(P2); (void)hipFree(P3); (void)hipFree(ldv); (void)hipFree(info); (void)hipFree(ticket);
        }
    // a non-positive-definite tile: both must report the same LAPACK info
    {
        const long ld = 96, M = 128, rows = M + 64 + 256;
        std::vector<double> h((size_t)rows * ld, 0.0);
        for (long r = 0; r < 64; ++r) for (long c = 0; c < 64; ++c) h[r * ld + c] = (r == c) ? (r == 37 ? -1.0 : 4.0) : 0.01;
        double *P1, *P2, *ldv; int *info, *ticket;
        CK(hipMalloc(&P1, sizeof(double) * h.size())); CK(hipMalloc(&P2, sizeof(double) * h.size()));
        CK(hipMalloc(&ldv, 64)); CK(hipMalloc(&info, 16)); CK(hipMalloc(&ticket, 256));
        CK(hipMemset(ticket, 0, 256)); CK(hipMemset(info, 0, 16)); CK(hipMemset(ldv, 0, 64));
        CK(hipMemcpy(P1, h.data(), sizeof(double) * h.size(), hipMemcpyHostToDevice)); CK(hipMemcpy(P2, h.data(), sizeof(double) * h.size(), hipMemcpyHostToDevice));
        launch_v1(P1, ld, M, info, ldv, ticket, 0); launch_v2<128>(P2, ld, M, info + 1, ldv + 1, ticket + 8, 0);
        CK(hipDeviceSynchronize());
        int infos[2]; CK(hipMemcpy(infos, info, sizeof(infos), hipMemcpyDeviceToHost));
        printf("non-PD tile: info v1 %d v2 %d %s\n", infos[0], infos[1], infos[0] == 38 && infos[1] == 38 ? "OK" : "MISMATCH");
        if (!(infos[0] == 38 && infos[1] == 38)) worst_fail = 1;
    }
    // ---- the 128-column leaf (NC = 8) against two round-3 leaves (the second applies the first in-leaf: kpre = 1)
    for (long M : Ms) {
        const long ld = 128 + 32, rows = M + 128 + 256;
        std::vector<double> h((size_t)rows * ld, 0.0);
        unsigned long st = 777 + 31 * (unsigned long)M;
        auto rnd = [&]() { st = st * 6364136223846793005ULL + 1442695040888963407ULL; return (double)((st >> 11) & 0xfffff) / 1048576.0 - 0.5; };
        for (long r = 0; r < M + 128; ++r)
            for (long c = 0; c < 128; ++c) h[r * ld + c] = 0.3 * rnd();
        for (long r = 0; r < 128; ++r)
            for (long c = 0; c < 128; ++c) h[r * ld + c] = exp(-0.05 * (double)((r - c) * (r - c))) + (r == c ? 6.0 : 0.0);
        double *P1, *P2, *P3, *ldv;
        int *info, *ticket;
        CK(hipMalloc(&P1, sizeof(double) * h.size())); CK(hipMalloc(&P2, sizeof(double) * h.size())); CK(hipMalloc(&P3, sizeof(double) * h.size()));
        CK(hipMalloc(&ldv, 8 * 64)); CK(hipMalloc(&info, 16)); CK(hipMalloc(&ticket, 256));
        CK(hipMemset(ticket, 0, 256)); CK(hipMemset(info, 0, 16));
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        float t1 = 1e9f, t2 = 1e9f, t3 = 1e9f;
        for (int rep = 0; rep < 4; ++rep) {
            float ms;
            CK(hipMemcpy(P1, h.data(), sizeof(double) * h.size(), hipMemcpyHostToDevice));
            CK(hipMemcpy(P2, h.data(), sizeof(double) * h.size(), hipMemcpyHostToDevice));
            CK(hipMemcpy(P3, h.data(), sizeof(double) * h.size(), hipMemcpyHostToDevice));
            CK(hipMemset(ldv, 0, 8 * 64));
            CK(hipEventRecord(e0, 0));
            launch_v1(P1, ld, M + 64, info, ldv + 0, ticket, 0);
            launch_v1(P1 + 64 * ld + 64, ld, M, info, ldv + 0, ticket, 1);
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1)); t1 = std::min(t1, ms);
            CK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL((panel64v2_kernel<64, 8>), dim3((unsigned)std::max(1L, (M + 63) / 64)), dim3(256), 0, 0, P2, ld, (int)M, info + 1, 0, 128, ldv + 1, ticket + 8, 0);
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1)); t2 = std::min(t2, ms);
            CK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL((panel64v2_kernel<128, 8>), dim3((unsigned)std::max(1L, (M + 127) / 128)), dim3(256), 0, 0, P3, ld, (int)M, info + 2, 0, 128, ldv + 2, ticket + 16, 0);
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1)); t3 = std::min(t3, ms);
            CK(hipGetLastError());
        }
        std::vector<double> a(h.size()), b(h.size()), c3(h.size());
        double lds[3]; int infos[3];
        CK(hipMemcpy(a.data(), P1, sizeof(double) * h.size(), hipMemcpyDeviceToHost));
        CK(hipMemcpy(b.data(), P2, sizeof(double) * h.size(), hipMemcpyDeviceToHost));
        CK(hipMemcpy(c3.data(), P3, sizeof(double) * h.size(), hipMemcpyDeviceToHost));
        CK(hipMemcpy(lds, ldv, sizeof(lds), hipMemcpyDeviceToHost)); CK(hipMemcpy(infos, info, sizeof(infos), hipMemcpyDeviceToHost));
        // the two-leaf reference counts n_valid = 64 per launch at col0 = 0: its second launch adds the first 64 columns of ITS tile — both 64-column halves are counted
        double eL = 0, eX = 0, eL3 = 0, eX3 = 0;
        for (long r = 0; r < M + 128; ++r)
            for (long c = 0; c < 128; ++c) {
                const double d2 = fabs(a[r * ld + c] - b[r * ld + c]), d3 = fabs(a[r * ld + c] - c3[r * ld + c]);
                if (r < 128) { if (c <= r) { eL = fmax(eL, d2); eL3 = fmax(eL3, d3); } }
                else { eX = fmax(eX, d2); eX3 = fmax(eX3, d3); }
            }
        const bool ok = eL < 1e-12 && eX < 1e-12 && eL3 < 1e-12 && eX3 < 1e-12 && fabs(lds[0] - lds[1]) < 1e-10 && fabs(lds[0] - lds[2]) < 1e-10 && infos[0] == infos[1] &&
                        infos[0] == infos[2] && eL == eL && eX == eX;
        if (!ok) worst_fail = 1;
        printf("128 columns, M=%6ld: two v1 leaves %7.1f us  v2<64,8> %7.1f us  v2<128,8> %7.1f us | dL %.1e %.1e  dX %.1e %.1e  logdet %.12g %.12g %.12g info %d %d %d  %s\n",
               M, t1 * 1e3, t2 * 1e3, t3 * 1e3, eL, eL3, eX, eX3, lds[0], lds[1], lds[2], infos[0], infos[1], infos[2], ok ? "OK" : "MISMATCH");
        fflush(stdout);
        (void)hipFree(P1); (void)hipFree(P2); (void)hipFree(P3); (void)hipFree(ldv); (void)hipFree(info); (void)hipFree(ticket);
    }
    printf(worst_fail ? "LEAF CHECK FAILED\n" : "LEAF CHECK PASSED\n");
    return worst_fail;
}
