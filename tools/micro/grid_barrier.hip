// Cost of a device-wide barrier inside one persistent launch (256 or 512 co-resident workgroups), to compare with the ~2.2-2.9 us
// gap between dependent launches (tools/micro/launch_floor.hip).  Barrier = one agent-scope atomic add per workgroup on a monotonic
// counter + a polled load; every spin is BOUNDED (a stuck barrier sets an error flag and falls through) so the grid always drains.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__device__ __forceinline__ void grid_barrier(unsigned* counter, unsigned target, int* err) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        atomicAdd(counter, 1u);
        long spins = 0;
        while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
            if (++spins > 20000000L) { *err = 1; break; }
            __builtin_amdgcn_s_sleep(1);
        }
    }
    __syncthreads();
}

// each phase touches `bytes_per_wg` of memory (0 = pure barrier cost) and then synchronises the grid
__global__ __launch_bounds__(256) void persistent(unsigned* counter, int* err, float* buf, int phases, int floats_per_wg) {
    const unsigned nwg = gridDim.x;
    float acc = 0.f;
    for (int p = 0; p < phases; ++p) {
        float* mine = buf + ((long)((blockIdx.x + p) % nwg)) * floats_per_wg;
        for (int i = threadIdx.x; i < floats_per_wg; i += 256) acc += mine[i];
        if (floats_per_wg && threadIdx.x == 0) mine[0] = acc * 1e-30f + 1.f;
        grid_barrier(counter, (unsigned)(p + 1) * nwg, err);
    }
    if (acc == 12345.678f) buf[0] = acc;
}

int main() {
    unsigned* counter; int* err; float* buf;
    CK(hipMalloc(&counter, 4)); CK(hipMalloc(&err, 4)); CK(hipMalloc(&buf, 64 << 20)); CK(hipMemset(buf, 0, 64 << 20));
    hipStream_t s; CK(hipStreamCreate(&s));
    for (int nwg : {256, 512}) {
        for (int fl : {0, 4096, 16384}) {          // 0, 16 KB, 64 KB per workgroup and phase
            const int phases = 2000;
            double best = 1e30;
            for (int rep = 0; rep < 3; ++rep) {
                CK(hipMemsetAsync(counter, 0, 4, s)); CK(hipMemsetAsync(err, 0, 4, s));
                CK(hipStreamSynchronize(s));
                auto t0 = std::chrono::high_resolution_clock::now();
                hipLaunchKernelGGL(persistent, dim3(nwg), dim3(256), 0, s, counter, err, buf, phases, fl);
                CK(hipStreamSynchronize(s));
                double us = std::chrono::duration<double, std::micro>(std::chrono::high_resolution_clock::now() - t0).count() / phases;
                if (us < best) best = us;
            }
            int e = 0; CK(hipMemcpy(&e, err, 4, hipMemcpyDeviceToHost));
            printf("%3d workgroups, %5.0f KB/phase total: %.2f us per phase (barrier incl.)%s\n", nwg, nwg * fl * 4.0 / 1024, best, e ? "  [BARRIER TIMED OUT]" : "");
        }
    }
    return 0;
}
