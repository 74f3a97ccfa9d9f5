// Do kernels on a CU-masked stream run beside kernels of other streams?  Two spin kernels of ~1 ms (128 workgroups each: they fit side by side
// on half a chip each) on (a) two plain streams, (b) one stream masked to 128 CUs + one plain, (c) two streams masked to disjoint halves.
// Wall time ~1 ms = concurrent, ~2 ms = serialised.  hipcc --offload-arch=gfx950 -O2.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdint>
__global__ void spin(long long cycles) {
    long long t0 = wall_clock64();
    while (wall_clock64() - t0 < cycles) {}
}
static double pair(hipStream_t a, hipStream_t b) {
    hipDeviceSynchronize();
    const long long cyc = 100000;          // 100 MHz constant clock: 1 ms
    auto t0 = std::chrono::steady_clock::now();
    hipLaunchKernelGGL(spin, dim3(128), dim3(256), 0, a, cyc);
    hipLaunchKernelGGL(spin, dim3(128), dim3(256), 0, b, cyc);
    hipDeviceSynchronize();
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}
static hipStream_t masked(int lo, int hi) {
    uint32_t mask[8] = {0};
    for (int i = lo; i < hi; ++i) mask[i >> 5] |= 1u << (i & 31);
    hipStream_t s = nullptr;
    if (hipExtStreamCreateWithCUMask(&s, 8, mask) != hipSuccess) printf("mask create failed\n");
    return s;
}
int main() {
    hipStream_t p0, p1;
    hipStreamCreateWithFlags(&p0, hipStreamNonBlocking);
    hipStreamCreateWithFlags(&p1, hipStreamNonBlocking);
    hipStream_t m_lo = masked(0, 128), m_hi = masked(128, 256);
    for (int rep = 0; rep < 3; ++rep) {
        printf("plain + plain: %.2f ms | masked[0,128) + plain: %.2f ms | masked[0,128) + masked[128,256): %.2f ms | same plain stream twice: %.2f ms\n",
               pair(p0, p1), pair(m_lo, p1), pair(m_lo, m_hi), pair(p0, p0));
    }
    return 0;
}
