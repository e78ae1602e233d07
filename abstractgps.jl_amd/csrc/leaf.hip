// leaf.hip — translation unit of panel64v2_kernel (leaf.hpp) and its launcher; compiled with -mllvm -amdgpu-mfma-vgpr-form.
#include "leaf.hpp"
#include "engine.hpp"

#include <algorithm>

namespace gpmi {

// one register-resident leaf on stream s: ncols = 64 (same contract as panel64_kernel: kernels.hpp) or 128 (kpre must be 0; mrows counts
// the rows below the 128×128 tile).  xr: rows of X per workgroup (64 / 128); 0 = 64 while that gives at most two (ncols = 64) / one
// (ncols = 128: 150 KB of LDS per workgroup) workgroups per CU, else 128.
int32_t launch_leaf_v2(hipStream_t s, double* Ajj, long lda, long mrows, int* info_dev, int col0, int n_valid, double* logdet_dev, int* ticket,
                       int kpre, int xr, int num_cus, int ncols) {
    const bool wide = ncols == 128;
    const bool xr64 = xr == 64 || (xr == 0 && mrows <= 64L * (wide ? 1 : 2) * num_cus);
    const unsigned nb = (unsigned)std::max(1L, (mrows + (xr64 ? 63 : 127)) / (xr64 ? 64 : 128));
    if (wide) {
        if (xr64)
            hipLaunchKernelGGL((panel64v2_kernel<64, 8>), dim3(nb), dim3(256), 0, s, Ajj, lda, (int)mrows, info_dev, col0, n_valid, logdet_dev, ticket, 0);
        else
            hipLaunchKernelGGL((panel64v2_kernel<128, 8>), dim3(nb), dim3(256), 0, s, Ajj, lda, (int)mrows, info_dev, col0, n_valid, logdet_dev, ticket, 0);
    } else if (xr64) {
        hipLaunchKernelGGL((panel64v2_kernel<64, 4>), dim3(nb), dim3(256), 0, s, Ajj, lda, (int)mrows, info_dev, col0, n_valid, logdet_dev, ticket, kpre);
    } else {
        hipLaunchKernelGGL((panel64v2_kernel<128, 4>), dim3(nb), dim3(256), 0, s, Ajj, lda, (int)mrows, info_dev, col0, n_valid, logdet_dev, ticket, kpre);
    }
    return (int32_t)hipGetLastError();  // hipError_t of the launch (0 = hipSuccess)
}

// C[m×N] −= P[m×K] · P[0:N, 0:K]ᵀ on stream s (leaf.hpp panel_updk_kernel); N multiple of 128, K multiple of 32, m >= N.
// rt = rows per workgroup / 16 (4, 2, 1; 0: the tallest tile that still gives one workgroup per CU)
int32_t launch_panel_updk(hipStream_t s, double* C, long ldc, const double* P, long ldp, long m, long N, long K, int rt, int num_cus) {
    const long nby = N / 128;
    if (rt != 1 && rt != 2 && rt != 4) {
        const long want = num_cus;
        rt = ((m + 63) / 64) * nby >= want ? 4 : (((m + 31) / 32) * nby >= want ? 2 : 1);
    }
    const dim3 grid((unsigned)((m + 16 * rt - 1) / (16 * rt)), (unsigned)nby);
    if (rt == 4) hipLaunchKernelGGL(panel_updk_kernel<4>, grid, dim3(256), 0, s, C, ldc, P, ldp, (int)m, (int)K);
    else if (rt == 2) hipLaunchKernelGGL(panel_updk_kernel<2>, grid, dim3(256), 0, s, C, ldc, P, ldp, (int)m, (int)K);
    else hipLaunchKernelGGL(panel_updk_kernel<1>, grid, dim3(256), 0, s, C, ldc, P, ldp, (int)m, (int)K);
    return (int32_t)hipGetLastError();
}

}  // namespace gpmi
