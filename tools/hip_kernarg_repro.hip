// Stand-alone probe (no library code): do kernels always see the arguments they were launched with?
// T host threads; each repeatedly creates a FRESH stream (new hardware queue, new kernel-argument pool), launches L kernels with
// distinct scalar + pointer arguments back to back, drains, checks every kernel's output, destroys the stream.  A kernel that ran with
// another launch's (stale) arguments leaves a wrong tag or writes to the wrong slot.
//   hipcc --offload-arch=gfx950 -O2 tools/hip_kernarg_repro.hip -o tools/bin/hip_kernarg_repro -lpthread
//   GPU_MAX_HW_QUEUES=16 [HIP_FORCE_DEV_KERNARG=0] tools/bin/hip_kernarg_repro [threads=8] [rounds=300] [launches=64]
// Context: first fits of fresh multi-device contexts (24 streams of 8 rank threads on one device) went wrong in 20 % of
// the runs with kernel arguments in device memory (the default) and in 2.5 % with HIP_FORCE_DEV_KERNARG=0 (profiles/r3).
#include <hip/hip_runtime.h>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)
struct Args { long a, b, c, d; };  // a few more bytes of arguments, all derived from the tag
__global__ void tagk(unsigned* out, unsigned tag, Args x, int spin) {
    long t0 = clock64();
    while (clock64() - t0 < spin) {}
    if (threadIdx.x == 0 && blockIdx.x == 0) *out = (x.a == tag * 3L && x.b == tag + 7L && x.c == ~(long)tag && x.d == (long)tag * (long)tag) ? tag : 0xdeadu;
}
int main(int argc, char** argv) {
    const int T = argc > 1 ? atoi(argv[1]) : 8, R = argc > 2 ? atoi(argv[2]) : 300, L = argc > 3 ? atoi(argv[3]) : 64;
    std::atomic<long> bad{0}, launches{0};
    std::vector<std::thread> th;
    for (int t = 0; t < T; ++t)
        th.emplace_back([&, t]() {
            unsigned* out;
            CK(hipMalloc(&out, sizeof(unsigned) * L));
            std::vector<unsigned> h(L);
            int lo, hi;
            CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
            for (int r = 0; r < R; ++r) {
                hipStream_t s[3];  // three fresh streams of mixed priority per round, like one rank of the driver
                for (int i = 0; i < 3; ++i) CK(hipStreamCreateWithPriority(&s[i], hipStreamNonBlocking, i == 1 ? hi : lo));
                CK(hipMemsetAsync(out, 0, sizeof(unsigned) * L, s[0]));
                CK(hipStreamSynchronize(s[0]));
                for (int i = 0; i < L; ++i) {
                    const unsigned tag = (unsigned)(1 + ((t * 1000003 + r * 1009 + i * 17) & 0xfffff));
                    hipLaunchKernelGGL(tagk, dim3(4), dim3(64), 0, s[i % 3], out + i, tag, Args{tag * 3L, tag + 7L, ~(long)tag, (long)tag * tag}, 200 + 50 * (i % 5));
                }
                for (int i = 0; i < 3; ++i) CK(hipStreamSynchronize(s[i]));
                CK(hipMemcpy(h.data(), out, sizeof(unsigned) * L, hipMemcpyDeviceToHost));
                for (int i = 0; i < L; ++i) {
                    const unsigned tag = (unsigned)(1 + ((t * 1000003 + r * 1009 + i * 17) & 0xfffff));
                    if (h[i] != tag) {
                        if (bad++ < 10) printf("thread %d round %d launch %d: slot holds %#x, expected %#x\n", t, r, i, h[i], tag);
                    }
                }
                launches += L;
                for (int i = 0; i < 3; ++i) CK(hipStreamDestroy(s[i]));
            }
            CK(hipFree(out));
        });
    for (auto& x : th) x.join();
    printf("threads %d rounds %d launches/round %d HIP_FORCE_DEV_KERNARG=%s GPU_MAX_HW_QUEUES=%s: %ld of %ld launches ran with wrong arguments -> %s\n", T, R, L,
           getenv("HIP_FORCE_DEV_KERNARG") ? getenv("HIP_FORCE_DEV_KERNARG") : "(default)", getenv("GPU_MAX_HW_QUEUES") ? getenv("GPU_MAX_HW_QUEUES") : "(default)",
           bad.load(), launches.load(), bad.load() ? "STALE KERNEL ARGUMENTS" : "ok");
    return bad.load() ? 1 : 0;
}
