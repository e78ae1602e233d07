// Which physical CUs does a stream created with hipExtStreamCreateWithCUMask run on?  Launches a grid of spinning workgroups
// on streams with different masks and prints, per mask, the set of (XCC, SE, CU) ids the workgroups reported.
// Build: hipcc --offload-arch=gfx950 -O2 tools/cumask_probe.hip -o tools/bin/cumask_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>
#include <set>
#include <map>
#include <string>
__global__ void probe(uint32_t* out, long spin) {
    if (threadIdx.x == 0) {
        uint32_t hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);    // HW_REG_HW_ID
        uint32_t xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);  // HW_REG_XCC_ID
        out[2 * blockIdx.x] = hw;
        out[2 * blockIdx.x + 1] = xcc;
    }
    long t0 = clock64();
    while (clock64() - t0 < spin) {}
}
static void run(const char* tag, hipStream_t s, uint32_t* dev, int nblk) {
    hipMemsetAsync(dev, 0xff, sizeof(uint32_t) * 2 * nblk, s);
    hipLaunchKernelGGL(probe, dim3(nblk), dim3(256), 0, s, dev, 200000L);
    hipError_t e = hipStreamSynchronize(s);
    std::vector<uint32_t> h(2 * nblk);
    hipMemcpy(h.data(), dev, sizeof(uint32_t) * 2 * nblk, hipMemcpyDeviceToHost);
    std::map<int, std::set<int>> per;
    for (int i = 0; i < nblk; ++i) {
        uint32_t hw = h[2 * i], xcc = h[2 * i + 1] & 0xf;
        int cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 0x7;
        per[(int)xcc].insert(se * 100 + sh * 16 + cu);
    }
    int tot = 0;
    printf("%s: err=%d\n", tag, (int)e);
    for (auto& kv : per) {
        printf("  xcc %d: %zu CUs:", kv.first, kv.second.size());
        for (int v : kv.second) printf(" %d.%d", v / 100, v % 100);
        printf("\n");
        tot += (int)kv.second.size();
    }
    printf("  total distinct CUs %d\n", tot);
}
int main() {
    uint32_t* dev;
    const int nblk = 4096;
    hipMalloc(&dev, sizeof(uint32_t) * 2 * nblk);
    hipStream_t s0;
    hipStreamCreate(&s0);
    run("unmasked", s0, dev, nblk);
    struct M { const char* tag; uint32_t w[8]; } masks[] = {
        {"bits 0..31", {0xffffffffu, 0, 0, 0, 0, 0, 0, 0}},
        {"bits 0..7", {0xffu, 0, 0, 0, 0, 0, 0, 0}},
        {"bits 224..255", {0, 0, 0, 0, 0, 0, 0, 0xffffffffu}},
        {"all but bits 224..255", {~0u, ~0u, ~0u, ~0u, ~0u, ~0u, ~0u, 0}},
        {"every 8th bit", {0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u}},
    };
    for (auto& m : masks) {
        hipStream_t s;
        hipError_t e = hipExtStreamCreateWithCUMask(&s, 8, m.w);
        if (e != hipSuccess) { printf("%s: create failed %d\n", m.tag, (int)e); continue; }
        run(m.tag, s, dev, nblk);
        hipStreamDestroy(s);
    }
    return 0;
}
