// Shared device helpers for the ECHR gfx950 kernels (wave64, CDNA4).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define ECHR_WAVE 64

namespace echr {

// ---- error plumbing (host) ---------------------------------------------------------------------
void set_error(const char* fmt, ...);
int check_launch(const char* what);

#define ECHR_REQUIRE(cond, ...)                 \
    do {                                        \
        if (!(cond)) {                          \
            echr::set_error(__VA_ARGS__);       \
            return -22; /* -EINVAL */           \
        }                                       \
    } while (0)

// ---- wave64 reductions via cross-lane shuffles ---------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}

// ---- transcendental helpers ---------------------------------------------------------------------
// tanh(x) = 1 - 2 / (exp(2x) + 1): v_exp_f32 + v_rcp_f32 (the 1-ulp hardware reciprocal -- __frcp_rn expands to the full IEEE division
// sequence, ten instructions per element in kernels that evaluate 84 M of them), |abs err| ~1e-7, saturates cleanly.
__device__ __forceinline__ float fast_tanh(float x) {
    float e = __expf(2.0f * x);
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(e + 1.0f);
}
__device__ __forceinline__ float fast_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }

// ---- Philox-4x32-10 dropout (bit-identical to echr_amd/philox.py) ---------------------------------
struct DropCfg {
    unsigned k0, k1;     // seed lo / hi
    unsigned offset;     // per-forward call counter (counter word 3)
    unsigned thresh;     // keep iff word >= thresh
    float scale;         // 1 / (1 - p)
    int active;          // 0 = eval mode (multiplier 1)
};

__device__ __forceinline__ unsigned philox_word(unsigned elem, unsigned step, unsigned site, unsigned offset,
                                                unsigned k0, unsigned k1) {
    unsigned c0 = elem >> 2, c1 = step, c2 = site, c3 = offset;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        unsigned hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        unsigned hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        unsigned n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    unsigned sel = elem & 3u;
    return sel == 0 ? c0 : (sel == 1 ? c1 : (sel == 2 ? c2 : c3));
}

// multiplicative dropout factor (0 or scale; 1 when inactive) for flat element `elem` of a site/step
__device__ __forceinline__ float drop_mult(const DropCfg& d, unsigned elem, unsigned step, unsigned site) {
    if (!d.active) return 1.0f;
    unsigned w = philox_word(elem, step, site, d.offset, d.k0, d.k1);
    return w >= d.thresh ? d.scale : 0.0f;
}

}  // namespace echr
