// Semantics check of v_permlane16_swap / v_permlane32_swap on gfx950 with both operands the same register:
// expected r32[0] + r32[1] = x[l] + x[l ^ 32], r16[0] + r16[1] = x[l] + x[l ^ 16]  (cross-row sums without LDS).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(float* out) {
    const int l = threadIdx.x;
    const unsigned v = __float_as_uint((float)(1 << (l >> 4)) * 1000.f + (float)(l & 15));      // row r: 2^r * 1000 + lane-in-row
    auto a = __builtin_amdgcn_permlane32_swap(v, v, false, false);
    auto b = __builtin_amdgcn_permlane16_swap(v, v, false, false);
    out[l] = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    out[64 + l] = __uint_as_float(b[0]) + __uint_as_float(b[1]);
}
int main() {
    float* d; float h[128];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) {
        auto x = [](int i) { return (float)(1 << (i >> 4)) * 1000.f + (float)(i & 15); };
        if (h[l] != x(l) + x(l ^ 32)) ++bad;
        if (h[64 + l] != x(l) + x(l ^ 16)) ++bad;
    }
    printf("permlane swap sums: %s (lane 0: %.0f %.0f, lane 17: %.0f %.0f, lane 40: %.0f %.0f)\n", bad ? "MISMATCH" : "ok", h[0], h[64], h[17], h[81], h[40], h[104]);
    return bad != 0;
}
