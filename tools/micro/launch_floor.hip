// Measures the per-launch floor of dependent kernels on one stream: eager vs hipGraph, trivial vs small streaming kernels.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
__global__ void trivial(float* p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] += 1.f; }
__global__ void stream8mb(const float4* __restrict__ in, float4* __restrict__ out, int n4) {
    int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n4) { float4 v = in[i]; v.x += 1.f; out[i] = v; } }
int main() {
    float *a, *b; CK(hipMalloc(&a, 64 << 20)); CK(hipMalloc(&b, 64 << 20)); CK(hipMemset(a, 0, 64 << 20));
    hipStream_t s; CK(hipStreamCreate(&s));
    const int R = 2000;
    auto run = [&](const char* name, int wgs, bool big) -> double {
        for (int w = 0; w < 2; ++w) {
            CK(hipStreamSynchronize(s));
            auto t0 = std::chrono::high_resolution_clock::now();
            for (int i = 0; i < R; ++i) {
                if (big) hipLaunchKernelGGL(stream8mb, dim3(wgs), dim3(256), 0, s, (const float4*)a, (float4*)b, wgs * 256);
                else hipLaunchKernelGGL(trivial, dim3(wgs), dim3(256), 0, s, a, wgs * 256);
            }
            CK(hipStreamSynchronize(s));
            double us = std::chrono::duration<double, std::micro>(std::chrono::high_resolution_clock::now() - t0).count() / R;
            if (w == 1) { printf("%-40s eager  %.2f us/launch\n", name, us); return us; }
        }
        return 0; };
    run("trivial 1 WG", 1, false); run("trivial 256 WG", 256, false); run("trivial 1024 WG", 1024, false);
    run("stream 4+4 MB (1024 WG)", 1024, true); run("stream 16+16 MB (4096 WG)", 4096, true);
    // graph of 200 dependent trivial launches
    for (int wgs : {256, 1024}) {
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
        for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(trivial, dim3(wgs), dim3(256), 0, s, a, wgs * 256);
        CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int w = 0; w < 2; ++w) {
            CK(hipStreamSynchronize(s));
            auto t0 = std::chrono::high_resolution_clock::now();
            for (int i = 0; i < 10; ++i) CK(hipGraphLaunch(ge, s));
            CK(hipStreamSynchronize(s));
            double us = std::chrono::duration<double, std::micro>(std::chrono::high_resolution_clock::now() - t0).count() / 2000;
            if (w == 1) printf("trivial %4d WG                          graph  %.2f us/launch\n", wgs, us);
        }
    }
    return 0;
}
