// Cross-stream edge latency on this runtime: kernel A (stream 1) -> [edge] -> kernel B (stream 2); B's first wall_clock64 minus A's last, in us
// (100 MHz constant clock), for (a) hipEventRecord + hipStreamWaitEvent with a default-fence event, (b) the same with hipEventDisableSystemFence,
// (c) hipStreamWriteValue32 + hipStreamWaitValue32 on signal memory, (d) same stream back to back (no edge).  hipcc --offload-arch=gfx950 -O2.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>
__global__ void stampA(unsigned long long* out, int spin) {
    long long t0 = wall_clock64();
    while (wall_clock64() - t0 < spin) {}
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = wall_clock64();
}
__global__ void stampB(unsigned long long* out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) out[1] = wall_clock64();
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main() {
    hipStream_t s1, s2;
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    unsigned long long* d;
    CK(hipMalloc(&d, 16));
    hipEvent_t e_def, e_nof;
    CK(hipEventCreateWithFlags(&e_def, hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&e_nof, hipEventDisableTiming | hipEventDisableSystemFence));
    uint32_t* sig = nullptr;
    bool have_sig = hipExtMallocWithFlags((void**)&sig, 64, hipMallocSignalMemory) == hipSuccess;
    if (!have_sig) { (void)hipGetLastError(); printf("signal memory unavailable\n"); }
    else CK(hipMemset(sig, 0, 64));
    auto measure = [&](int mode, const char* name) -> int {
        std::vector<double> v;
        for (int it = 0; it < 40; ++it) {
            CK(hipDeviceSynchronize());
            hipLaunchKernelGGL(stampA, dim3(256), dim3(256), 0, s1, d, 2000);          // 20 us of spinning on the whole chip
            if (mode == 0) { CK(hipEventRecord(e_def, s1)); CK(hipStreamWaitEvent(s2, e_def, 0)); }
            else if (mode == 1) { CK(hipEventRecord(e_nof, s1)); CK(hipStreamWaitEvent(s2, e_nof, 0)); }
            else if (mode == 2) { CK(hipStreamWriteValue32(s1, sig, (uint32_t)(it + 1), 0)); CK(hipStreamWaitValue32(s2, sig, (uint32_t)(it + 1), hipStreamWaitValueGte, 0xffffffffu)); }
            hipLaunchKernelGGL(stampB, dim3(256), dim3(256), 0, mode == 3 ? s1 : s2, d);
            CK(hipDeviceSynchronize());
            unsigned long long h[2];
            CK(hipMemcpy(h, d, 16, hipMemcpyDeviceToHost));
            if (it >= 5) v.push_back((double)((long long)h[1] - (long long)h[0]) / 100.0);
        }
        std::sort(v.begin(), v.end());
        printf("%-44s min %.2f us  median %.2f us  max %.2f us\n", name, v.front(), v[v.size() / 2], v.back());
        return 0;
    };
    if (measure(3, "same stream, back to back")) return 1;
    if (measure(0, "event (default) record + wait")) return 1;
    if (measure(1, "event (no system fence) record + wait")) return 1;
    if (have_sig && measure(2, "stream write value + wait value")) return 1;
    return 0;
}
