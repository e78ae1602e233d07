// Second stand-alone probe for stale kernel arguments (no library code) — closer to the launch mix of csrc/multi.hip than
// hip_kernarg_repro.hip: LARGE by-value arguments (the library passes 0.2–2 KB per launch: GridMap + pointers), several kernels per
// stream, events between the streams of a thread, memsets / 2-D copies in between, streams used for the FIRST time while the other
// threads are launching.  Every launch carries W words derived from its tag; the kernel checks all of them and reports the first
// word that is wrong and what it held.
//   hipcc --offload-arch=gfx950 -O2 tools/hip_kernarg_repro2.hip -o tools/bin/hip_kernarg_repro2 -lpthread
//   GPU_MAX_HW_QUEUES=16 tools/bin/hip_kernarg_repro2 [threads=8] [rounds=100] [launches=48] [words=256] [events=1] [memops=1] [lds=1]
#include <hip/hip_runtime.h>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)
constexpr int WMAX = 480;  // 1 920 bytes of arguments
struct Big { unsigned w[WMAX]; };
__host__ __device__ inline unsigned word_of(unsigned tag, int i) { return tag * 2654435761u + (unsigned)i * 40503u + 17u; }
// out[0] = tag when every word is right, else 0x80000000 | index of the first wrong word; out[1] = what that word held
__global__ void bigk(unsigned* out, unsigned tag, int words, Big x, int spin, int use_lds) {
    __shared__ unsigned sh[4096];
    __shared__ unsigned bad_i, bad_v;
    if (threadIdx.x == 0) { bad_i = 0xffffffffu; bad_v = 0; }
    if (use_lds) for (int i = threadIdx.x; i < 4096; i += blockDim.x) sh[i] = i;
    __syncthreads();
    long t0 = clock64();
    while (clock64() - t0 < spin) {}
    for (int i = threadIdx.x; i < words; i += blockDim.x)
        if (x.w[i] != word_of(tag, i)) {
            const unsigned old = atomicMin(&bad_i, (unsigned)i);
            if ((unsigned)i < old) bad_v = x.w[i];
        }
    __syncthreads();
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        out[0] = bad_i == 0xffffffffu ? tag + (use_lds ? sh[0] : 0) : (0x80000000u | bad_i);
        out[1] = bad_v;
    }
}
int main(int argc, char** argv) {
    const int T = argc > 1 ? atoi(argv[1]) : 8, R = argc > 2 ? atoi(argv[2]) : 100, L = argc > 3 ? atoi(argv[3]) : 48;
    const int W = argc > 4 ? std::min(atoi(argv[4]), WMAX) : 256, EV = argc > 5 ? atoi(argv[5]) : 1, MEM = argc > 6 ? atoi(argv[6]) : 1;
    const int LDS = argc > 7 ? atoi(argv[7]) : 1;
    std::atomic<long> bad{0}, launches{0};
    std::atomic<int> go{0};
    std::vector<std::thread> th;
    for (int t = 0; t < T; ++t)
        th.emplace_back([&, t]() {
            unsigned *out, *scratch;
            CK(hipMalloc(&out, sizeof(unsigned) * 2 * L));
            CK(hipMalloc(&scratch, 1 << 20));
            std::vector<unsigned> h(2 * L);
            int lo, hi;
            CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
            std::vector<hipEvent_t> ev(L);
            for (auto& e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            go++;
            while (go.load() < T) std::this_thread::yield();  // all threads start their first round together
            for (int r = 0; r < R; ++r) {
                hipStream_t s[3];  // fresh streams: new hardware queues / argument pools, first use while the others launch
                for (int i = 0; i < 3; ++i) CK(hipStreamCreateWithPriority(&s[i], hipStreamNonBlocking, i == 1 ? hi : lo));
                CK(hipMemsetAsync(out, 0, sizeof(unsigned) * 2 * L, s[0]));
                CK(hipEventRecord(ev[0], s[0]));
                CK(hipStreamWaitEvent(s[1], ev[0], 0));
                CK(hipStreamWaitEvent(s[2], ev[0], 0));
                for (int i = 0; i < L; ++i) {
                    const unsigned tag = (unsigned)(1 + ((t * 1000003 + r * 1009 + i * 17) & 0xfffff));
                    Big b;
                    for (int k = 0; k < W; ++k) b.w[k] = word_of(tag, k);
                    const int si = i % 3;
                    if (MEM && i % 5 == 2) CK(hipMemsetAsync(scratch, i, 1 << 16, s[si]));
                    if (MEM && i % 7 == 3) CK(hipMemcpy2DAsync(scratch + (1 << 17), 2048, scratch, 1024, 1024, 64, hipMemcpyDeviceToDevice, s[si]));
                    if (EV && i > 0 && i % 2 == 0) CK(hipStreamWaitEvent(s[si], ev[i - 1], 0));  // depend on the previous launch (other stream)
                    hipLaunchKernelGGL(bigk, dim3(1 + i % 4), dim3(256), 0, s[si], out + 2 * i, tag, W, b, 300 + 200 * (i % 5), LDS);
                    if (EV) CK(hipEventRecord(ev[i], s[si]));
                }
                for (int i = 0; i < 3; ++i) CK(hipStreamSynchronize(s[i]));
                CK(hipMemcpy(h.data(), out, sizeof(unsigned) * 2 * L, hipMemcpyDeviceToHost));
                for (int i = 0; i < L; ++i) {
                    const unsigned tag = (unsigned)(1 + ((t * 1000003 + r * 1009 + i * 17) & 0xfffff));
                    if (h[2 * i] != tag) {
                        if (bad++ < 12) {
                            if (h[2 * i] & 0x80000000u) {
                                const int wi = (int)(h[2 * i] & 0xffffff);
                                printf("thread %d round %d launch %d: argument word %d held %#x, expected %#x", t, r, i, wi, h[2 * i + 1], word_of(tag, wi));
                                for (int j = std::max(0, i - 6); j < i; ++j) {  // was it an earlier launch's word?
                                    const unsigned tj = (unsigned)(1 + ((t * 1000003 + r * 1009 + j * 17) & 0xfffff));
                                    if (word_of(tj, wi) == h[2 * i + 1]) printf("  (= launch %d's)", j);
                                }
                                printf("\n");
                            } else {
                                printf("thread %d round %d launch %d: result slot holds %#x, expected %#x\n", t, r, i, h[2 * i], tag);
                            }
                        }
                    }
                }
                launches += L;
                for (int i = 0; i < 3; ++i) CK(hipStreamDestroy(s[i]));
            }
            CK(hipFree(out));
            CK(hipFree(scratch));
        });
    for (auto& x : th) x.join();
    printf("threads %d rounds %d launches/round %d arg-bytes %d events %d memops %d lds %d HIP_FORCE_DEV_KERNARG=%s GPU_MAX_HW_QUEUES=%s: %ld of %ld launches wrong -> %s\n", T, R, L,
           4 * W + 24, EV, MEM, LDS, getenv("HIP_FORCE_DEV_KERNARG") ? getenv("HIP_FORCE_DEV_KERNARG") : "(default)",
           getenv("GPU_MAX_HW_QUEUES") ? getenv("GPU_MAX_HW_QUEUES") : "(default)", bad.load(), launches.load(), bad.load() ? "STALE KERNEL ARGUMENTS" : "ok");
    return bad.load() ? 1 : 0;
}
