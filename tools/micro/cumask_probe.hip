// Which CUs does a stream created with hipExtStreamCreateWithCUMask get?  Every workgroup of a 2048-workgroup launch records its XCC id and its
// HW_ID (SE / SH / CU fields); the host prints, per mask, the number of distinct (xcc, se, cu) triples per XCC.  hipcc --offload-arch=gfx950 -O2.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <set>
#include <map>
#include <vector>
__global__ void probe(uint32_t* out, int spin) {
    if (threadIdx.x == 0) {
        uint32_t xcc, hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        out[2 * blockIdx.x] = xcc;
        out[2 * blockIdx.x + 1] = hw;
    }
    // keep the workgroup resident for a while so that the launch spreads over every CU it may use
    long long t0 = clock64();
    while (clock64() - t0 < spin) {}
}
static void run(const char* name, hipStream_t s) {
    const int n = 2048;
    uint32_t* d;
    hipMalloc(&d, 2 * n * sizeof(uint32_t));
    hipLaunchKernelGGL(probe, dim3(n), dim3(256), 0, s, d, 200000);
    hipStreamSynchronize(s);
    std::vector<uint32_t> h(2 * n);
    hipMemcpy(h.data(), d, 2 * n * sizeof(uint32_t), hipMemcpyDeviceToHost);
    std::map<uint32_t, std::set<uint32_t>> per;
    for (int i = 0; i < n; ++i) {
        const uint32_t xcc = h[2 * i] & 0xf, hw = h[2 * i + 1];
        // HW_ID (gfx9): wave_id [3:0], simd_id [5:4], pipe_id [7:6], cu_id [11:8], sh_id [12], se_id [15:13]
        per[xcc].insert((hw >> 8) & 0xff);
    }
    int total = 0;
    printf("%-28s", name);
    for (auto& kv : per) { printf(" xcc%u:%zu", kv.first, kv.second.size()); total += (int)kv.second.size(); }
    if (per.size()) { printf(" | xcc0 ids:"); for (uint32_t id : per.begin()->second) printf(" %02x", id); }
    printf("  | total %d CUs\n", total);
    hipFree(d);
}
int main() {
    hipStream_t s0;
    hipStreamCreate(&s0);
    run("plain stream", s0);
    const int widths[] = {32, 64, 128, 192, 224, 256};
    for (int wdt : widths) {
        uint32_t mask[8] = {0};
        for (int i = 0; i < wdt; ++i) mask[i >> 5] |= 1u << (i & 31);
        hipStream_t s;
        if (hipExtStreamCreateWithCUMask(&s, 8, mask) != hipSuccess) { printf("mask %d: create failed\n", wdt); continue; }
        char nm[64];
        snprintf(nm, sizeof nm, "first %d bits", wdt);
        run(nm, s);
        hipStreamDestroy(s);
    }
    // the LAST 64 bits: which CUs, and are they disjoint from the first 192 bits' CUs?
    {
        uint32_t ml[8] = {0, 0, 0, 0, 0, 0, 0xffffffffu, 0xffffffffu};
        hipStream_t sl;
        if (hipExtStreamCreateWithCUMask(&sl, 8, ml) == hipSuccess) { run("bits 192..255", sl); hipStreamDestroy(sl); }
        uint32_t mm[8] = {0, 0, 0, 0, 0xffffffffu, 0xffffffffu, 0, 0};
        if (hipExtStreamCreateWithCUMask(&sl, 8, mm) == hipSuccess) { run("bits 128..191", sl); hipStreamDestroy(sl); }
    }
    // every fourth bit: 64 CUs spread over the mask
    uint32_t m4[8];
    for (int i = 0; i < 8; ++i) m4[i] = 0x11111111u;
    hipStream_t s4;
    if (hipExtStreamCreateWithCUMask(&s4, 8, m4) == hipSuccess) { run("every 4th bit (64)", s4); hipStreamDestroy(s4); }
    return 0;
}
