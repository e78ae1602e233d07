// Round 6: what ONE dependent kernel launch costs on this device — the floor inside every latency-bound launch of the factorisation (leaf 29 µs, in-panel update 12-57 µs,
// vector-solve launches 8-10 µs).  A chain of N launches of a kernel that does nothing (or touches one cache line per workgroup) on one stream, for the launch shapes of
// the leaf: 1 / 128 / 512 workgroups of 256 threads, with 0 or 152 KB of dynamic LDS (the 128-column leaf's footprint: one workgroup per CU).
//   hipcc --offload-arch=gfx950 -O3 tools/launch_floor.hip -o tools/bin/launch_floor && tools/bin/launch_floor
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void nop_kernel(double* p, int touch) {
    extern __shared__ double lds[];
    if (touch && threadIdx.x == 0) {
        lds[0] = p[blockIdx.x * 16];
        p[blockIdx.x * 16] = lds[0] + 1.0;
    }
}
int main() {
    double* p;
    hipMalloc(&p, 8 * 16 * 4096);
    hipMemset(p, 0, 8 * 16 * 4096);
    hipStream_t s;
    hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    hipFuncSetAttribute((const void*)nop_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 155648);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int N = 2000;
    for (int touch = 0; touch < 2; ++touch)
        for (size_t lds : {(size_t)0, (size_t)155648})
            for (int wgs : {1, 128, 512}) {
                float best = 1e30f;
                for (int rep = 0; rep < 4; ++rep) {
                    hipEventRecord(e0, s);
                    for (int i = 0; i < N; ++i) hipLaunchKernelGGL(nop_kernel, dim3(wgs), dim3(256), lds, s, p, touch);
                    hipEventRecord(e1, s);
                    hipEventSynchronize(e1);
                    float ms;
                    hipEventElapsedTime(&ms, e0, e1);
                    if (rep && ms < best) best = ms;
                }
                printf("{\"touch\": %d, \"lds_bytes\": %zu, \"workgroups\": %d, \"us_per_dependent_launch\": %.3f}\n", touch, lds, wgs, best * 1e3f / N);
            }
    return 0;
}
