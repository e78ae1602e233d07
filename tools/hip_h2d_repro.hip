// Third stand-alone probe (no library code): is a small asynchronous host-to-device copy from PAGEABLE memory always visible to the
// kernel queued behind it — on the same stream, and on another stream behind an event — while several threads use fresh queues?
// (What the multi-device driver does at the start of every fit: 16 KB of scaled inputs / noise / right-hand sides per rank uploaded
// from std::vector storage on the rank's main stream, the assembly kernel behind it.)  T threads; per round each creates three fresh
// streams of mixed priority, uploads B bytes of a tag-derived pattern per launch and has a kernel compare the device buffer with the
// pattern; a mismatch reports the first wrong word and what it held.
//   hipcc --offload-arch=gfx950 -O2 tools/hip_h2d_repro.hip -o tools/bin/hip_h2d_repro -lpthread
//   GPU_MAX_HW_QUEUES=16 tools/bin/hip_h2d_repro [threads=8] [rounds=100] [launches=24] [bytes=16384] [cross_stream=1]
#include <hip/hip_runtime.h>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)
__host__ __device__ inline unsigned word_of(unsigned tag, int i) { return tag * 2654435761u + (unsigned)i * 40503u + 17u; }
// res[0] = tag if buf[0..n) == pattern(tag), else 0x80000000 | first wrong index; res[1] = the word found there
__global__ void checkk(const unsigned* buf, int n, unsigned tag, unsigned* res) {
    __shared__ unsigned bad_i, bad_v;
    if (threadIdx.x == 0) { bad_i = 0xffffffffu; bad_v = 0; }
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += blockDim.x)
        if (buf[i] != word_of(tag, i)) {
            const unsigned old = atomicMin(&bad_i, (unsigned)i);
            if ((unsigned)i < old) bad_v = buf[i];
        }
    __syncthreads();
    if (threadIdx.x == 0) { res[0] = bad_i == 0xffffffffu ? tag : (0x80000000u | bad_i); res[1] = bad_v; }
}
int main(int argc, char** argv) {
    const int T = argc > 1 ? atoi(argv[1]) : 8, R = argc > 2 ? atoi(argv[2]) : 100, L = argc > 3 ? atoi(argv[3]) : 24;
    const int B = argc > 4 ? atoi(argv[4]) : 16384, X = argc > 5 ? atoi(argv[5]) : 1, W = B / 4;
    std::atomic<long> bad{0}, launches{0};
    std::atomic<int> go{0};
    std::vector<std::thread> th;
    for (int t = 0; t < T; ++t)
        th.emplace_back([&, t]() {
            unsigned *dev, *res;
            CK(hipMalloc(&dev, (size_t)B * L));  // one device buffer per launch of a round: no reuse inside a round
            CK(hipMalloc(&res, sizeof(unsigned) * 2 * L));
            std::vector<std::vector<unsigned>> host((size_t)L, std::vector<unsigned>((size_t)W));  // pageable
            std::vector<unsigned> h(2 * L);
            std::vector<hipEvent_t> ev(L);
            for (auto& e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            int lo, hi;
            CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
            go++;
            while (go.load() < T) std::this_thread::yield();
            for (int r = 0; r < R; ++r) {
                hipStream_t s[3];
                for (int i = 0; i < 3; ++i) CK(hipStreamCreateWithPriority(&s[i], hipStreamNonBlocking, i == 1 ? hi : lo));
                CK(hipMemsetAsync(res, 0, sizeof(unsigned) * 2 * L, s[0]));
                CK(hipMemsetAsync(dev, 0xee, (size_t)B * L, s[0]));  // stale contents are recognisable
                CK(hipStreamSynchronize(s[0]));
                for (int i = 0; i < L; ++i) {
                    const unsigned tag = (unsigned)(1 + ((t * 1000003 + r * 1009 + i * 17) & 0xfffff));
                    for (int k = 0; k < W; ++k) host[i][k] = word_of(tag, k);
                    const int si = i % 3, sk = X ? (i + 1) % 3 : si;
                    unsigned* d = dev + (size_t)i * W;
                    CK(hipMemcpyAsync(d, host[i].data(), (size_t)B, hipMemcpyHostToDevice, s[si]));
                    if (sk != si) {
                        CK(hipEventRecord(ev[i], s[si]));
                        CK(hipStreamWaitEvent(s[sk], ev[i], 0));
                    }
                    hipLaunchKernelGGL(checkk, dim3(1), dim3(256), 0, s[sk], d, W, tag, res + 2 * i);
                }
                for (int i = 0; i < 3; ++i) CK(hipStreamSynchronize(s[i]));
                CK(hipMemcpy(h.data(), res, sizeof(unsigned) * 2 * L, hipMemcpyDeviceToHost));
                for (int i = 0; i < L; ++i) {
                    const unsigned tag = (unsigned)(1 + ((t * 1000003 + r * 1009 + i * 17) & 0xfffff));
                    if (h[2 * i] != tag && bad++ < 12) {
                        if (h[2 * i] & 0x80000000u)
                            printf("thread %d round %d upload %d: word %u of the device buffer held %#x, expected %#x%s\n", t, r, i, h[2 * i] & 0xffffffu, h[2 * i + 1],
                                   word_of(tag, (int)(h[2 * i] & 0xffffffu)), h[2 * i + 1] == 0xeeeeeeeeu ? "  (= the stale fill)" : "");
                        else
                            printf("thread %d round %d upload %d: result slot holds %#x, expected %#x\n", t, r, i, h[2 * i], tag);
                    }
                }
                launches += L;
                for (int i = 0; i < 3; ++i) CK(hipStreamDestroy(s[i]));
            }
            CK(hipFree(dev));
            CK(hipFree(res));
        });
    for (auto& x : th) x.join();
    printf("threads %d rounds %d uploads/round %d bytes %d consumer %s GPU_MAX_HW_QUEUES=%s: %ld of %ld uploads seen wrong by the kernel behind them -> %s\n", T, R, L, B,
           X ? "on another stream behind an event" : "on the same stream", getenv("GPU_MAX_HW_QUEUES") ? getenv("GPU_MAX_HW_QUEUES") : "(default)", bad.load(), launches.load(),
           bad.load() ? "STALE UPLOAD" : "ok");
    return bad.load() ? 1 : 0;
}
