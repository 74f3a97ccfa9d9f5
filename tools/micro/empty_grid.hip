// What a launch shaped like the h2 product costs before it does any arithmetic: G workgroups x 512 threads x 66 KB of dynamic LDS,
// (a) empty, (b) 48 lockstep barriers, (c) + a 64 KB store tail per workgroup in 64-byte row pieces / in 512-byte rows.  hipcc --offload-arch=gfx950 -O2.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
extern __shared__ unsigned char sm[];
__global__ __launch_bounds__(512, 2) void k_empty(float* out, int iters, int mode, int ldc) {
    float acc[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) acc[i] = (float)(threadIdx.x + i);
    for (int it = 0; it < iters; ++it) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int i = 0; i < 32; ++i) acc[i] += 1.f;
    }
    if (mode == 0) { if (acc[3] == -1.f) out[0] = acc[5]; return; }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m0 = (blockIdx.x % 6) * 128, n0 = (blockIdx.x / 6) * 128;
    if (mode == 1) {          // the 16x16 accumulator layout: a store instruction writes 4 rows x 16 columns (64-byte pieces)
        const int wm = (wave >> 2) * 64, wn = (wave & 3) * 32, l15 = lane & 15, rq = lane >> 4;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) out[(long)(m0 + wm + i * 16 + 4 * rq + r) * ldc + n0 + wn + j * 16 + l15] = acc[(j * 4 + i) * 4 + r];
    } else {                  // whole 512-byte rows: a wave writes 2 rows of 128 columns per instruction (16 B per lane)
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int row = wave * 16 + q * 2 + (lane >> 5), c4 = (lane & 31) * 4;
            float* d = out + (long)(m0 + row) * ldc + n0 + c4;
            d[0] = acc[4 * q]; d[1] = acc[4 * q + 1]; d[2] = acc[4 * q + 2]; d[3] = acc[4 * q + 3];
        }
    }
}
int main() {
    float* out; CK(hipMalloc(&out, 64 << 20));
    hipStream_t s; CK(hipStreamCreate(&s));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_empty), hipFuncAttributeMaxDynamicSharedMemorySize, 99 * 1024));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int R = 200;
    for (int lds : {0, 66 * 1024, 99 * 1024})
        for (int grid : {240, 480, 1024})
            for (int cfg = 0; cfg < 4; ++cfg) {
                const int iters = cfg == 0 ? 0 : 48, mode = cfg <= 1 ? 0 : cfg - 1;
                for (int w = 0; w < 2; ++w) {
                    CK(hipEventRecord(e0, s));
                    for (int i = 0; i < R; ++i) hipLaunchKernelGGL(k_empty, dim3(grid), dim3(512), lds, s, out, iters, mode, 5001);
                    CK(hipEventRecord(e1, s));
                    CK(hipStreamSynchronize(s));
                }
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                printf("lds %6d  grid %5d  %-34s %7.2f us\n", lds, grid, cfg == 0 ? "empty" : cfg == 1 ? "48 barriers" : cfg == 2 ? "48 barriers + 64-byte-piece tail" : "48 barriers + 512-byte-row tail", ms * 1e3 / R);
            }
    return 0;
}
