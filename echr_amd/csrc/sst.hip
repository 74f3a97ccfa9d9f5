// SST proposal encoder on gfx950 (reference: models/sst_model.py:5-40 -- nn.LSTM(video_dim -> hidden, 2 layers, batch_first,
// inter-layer dropout) over ONE video [1,T,D], Linear(hidden -> K) + sigmoid) and its weighted-BCE criterion
// (misc/utils.py:78-99).  SURVEY section 8-f row 1: the producer of `tap_feats` for the caption path.
//
// Batch size is 1, so the recurrence is a chain of GEMVs: there is no MFMA shape in it.  Layout of the work:
//   * layer 0's input-side products are batched over the T rows on the fp32 MFMA GEMM (gemm.hip);
//   * the two layers run as a WAVEFRONT: launch k = layer 0 at step k || layer 1 at step k-1 (T+1 dependent launches instead of
//     2T).  A workgroup owns 2 hidden units; wave g (of 4) computes gate g's rows of W_hh . h(t-1) (layer 1: also W_ih1 . h0(t))
//     with float4 lanes + cross-lane reduction, then 2 threads finish the cell (the 4 gates of a unit meet in LDS).  The weight
//     rows of a workgroup always land on the same XCD -> they stay L2-resident across the launches;
//   * backward mirrors it with the transposed matrices: launch k = layer 1 at step T-1-k || layer 0 at step T-k, where layer 0's
//     upstream gradient W_ih1^T . dG1(t) (through the inter-layer dropout mask) is formed inside the step;
//   * weight gradients are batched TN GEMMs over the T rows afterwards.
#include "echr_common.h"
#include "echr_internal.h"

namespace echr {

DropCfg make_drop(const echr_dropout* d, float p);
enum { SITE_SST = 5 };
constexpr int UPW = 2;          // hidden units per workgroup (256 workgroups at H = 512: one per CU)

// dot of one weight row with a vector held in LDS; lanes stride the k axis in float4.  Eight row loads are issued before the first
// FMA (a 2048-long row is exactly one batch per lane): the step kernels are latency-bound, not bandwidth-bound.
__device__ __forceinline__ float row_dot(const float* __restrict__ wrow, const float* __restrict__ v, int K, int lane) {
    float acc = 0.f;
    int k = lane * 4;
    for (; k + 7 * 256 < K; k += 8 * 256) {
        float4 w[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) w[j] = *reinterpret_cast<const float4*>(wrow + k + j * 256);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float4 x4 = *reinterpret_cast<const float4*>(v + k + j * 256);
            acc += w[j].x * x4.x + w[j].y * x4.y + w[j].z * x4.z + w[j].w * x4.w;
        }
    }
    for (; k < K; k += 256) {
        const float4 w4 = *reinterpret_cast<const float4*>(wrow + k);
        const float4 x4 = *reinterpret_cast<const float4*>(v + k);
        acc += w4.x * x4.x + w4.y * x4.y + w4.z * x4.z + w4.w * x4.w;
    }
    return wave_sum(acc);
}

// ---- wavefront over the two layers ----------------------------------------------------------------------------------
// Layer 1 at step t needs only layer 0's output of step t, so launch k runs layer 0 at step k and layer 1 at step k-1 side by
// side (blockIdx.y = role): T+1 dependent launches per direction instead of 2T.  Layer 1 then cannot take its input-side
// pre-activations from a batched GEMM over all T rows; its step multiplies [W_ih1 | W_hh1] with [h0d(t) ; h1(t-1)] instead.
struct SstFwdRole {
    const float* W[2];        // up to two [4H, H] matrices ...
    const float* v[2];        // ... each times one [H] vector (null: that product is skipped, e.g. h(-1) = 0)
    const float* base;        // [4H] pre-activation base: a GIN row (layer 0) or b_ih (layer 1)
    const float* base2;       // [4H] second bias (layer 1: b_hh) or null
    const float* cprev;
    float *act, *hout, *cout, *hdrop;
    int t, active;
};
__global__ __launch_bounds__(256) void sst_wave_fwd_kernel(SstFwdRole r0, SstFwdRole r1, int H, DropCfg dc) {
    const SstFwdRole r = blockIdx.y ? r1 : r0;
    if (!r.active) return;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* sv = sm;                  // [2][H]
    float* pre = sm + 2 * H;         // [4][UPW]
    const int u0 = blockIdx.x * UPW;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // the finishing threads fetch their cell state now: the load overlaps the GEMV instead of trailing it
    float cprev_v = 0.f;
    if (threadIdx.x < UPW && u0 + (int)threadIdx.x < H && r.cprev) cprev_v = r.cprev[u0 + threadIdx.x];
#pragma unroll
    for (int m = 0; m < 2; ++m)
        if (r.v[m]) for (int k = threadIdx.x; k < H; k += 256) sv[m * H + k] = r.v[m][k];
    __syncthreads();
    float acc[UPW];
#pragma unroll
    for (int i = 0; i < UPW; ++i) acc[i] = 0.f;
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        if (!r.v[m]) continue;
        const float* W = r.W[m];
        for (int k = lane * 4; k < H; k += 256) {
            const float4 x4 = *reinterpret_cast<const float4*>(sv + m * H + k);
#pragma unroll
            for (int i = 0; i < UPW; ++i) {
                const int u = min(u0 + i, H - 1);
                const float4 w4 = *reinterpret_cast<const float4*>(W + (long)(wave * H + u) * H + k);
                acc[i] += w4.x * x4.x + w4.y * x4.y + w4.z * x4.z + w4.w * x4.w;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < UPW; ++i) {
        const float d = wave_sum(acc[i]);
        if (lane == 0 && u0 + i < H) {
            const int row = wave * H + u0 + i;
            pre[wave * UPW + i] = d + r.base[row] + (r.base2 ? r.base2[row] : 0.f);
        }
    }
    __syncthreads();
    if (threadIdx.x < UPW && u0 + (int)threadIdx.x < H) {
        const int i = threadIdx.x, u = u0 + i;
        const float gi = fast_sigmoid(pre[i]), gf = fast_sigmoid(pre[UPW + i]), gg = tanhf(pre[2 * UPW + i]), go = fast_sigmoid(pre[3 * UPW + i]);
        const float c = gf * cprev_v + gi * gg;
        const float h = go * tanhf(c);
        r.act[u] = gi; r.act[H + u] = gf; r.act[2 * H + u] = gg; r.act[3 * H + u] = go;
        r.cout[u] = c; r.hout[u] = h;
        if (r.hdrop) r.hdrop[u] = h * drop_mult(dc, (unsigned)(r.t * H + u), 0u, SITE_SST);
    }
}

// backward wavefront: launch k runs layer 1 at step t = T-1-k and layer 0 at step t+1.
//   d h(t) = dh_base[u] + drop?(WT[0][u,:] . vec[0]) + WT[1][u,:] . vec[1]
//   layer 1: WT[0] = W_hh1^T, vec[0] = dG1(t+1); dh_base = upstream gradient row.
//   layer 0: WT[0] = W_ih1^T (through the inter-layer dropout mask of (t,u)), vec[0] = dG1(t); WT[1] = W_hh0^T, vec[1] = dG0(t+1).
struct SstBwdRole {
    const float* WT[2];       // [H, 4H] transposed matrices
    const float* vec[2];      // [4H] vectors or null
    const float* dh_base;     // [H] or null
    const float *act, *c, *cprev;
    float *dc, *dg;
    int drop_first, t, active;
};
__global__ __launch_bounds__(256) void sst_wave_bwd_kernel(SstBwdRole r0, SstBwdRole r1, int H, DropCfg dcfg) {
    const SstBwdRole r = blockIdx.y ? r1 : r0;
    if (!r.active) return;
    static_assert(UPW == 2, "wave -> (unit, matrix) map below assumes two units per workgroup");
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* sg = sm;                  // [2][4H]
    float* part = sm + 8 * H;        // [2][UPW]
    const int u0 = blockIdx.x * UPW;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // the finishing threads fetch their cell operands now: the loads overlap the two dot products instead of trailing them
    float p_base = 0.f, p_gi = 0.f, p_gf = 0.f, p_gg = 0.f, p_go = 0.f, p_c = 0.f, p_cprev = 0.f, p_dc = 0.f;
    if (threadIdx.x < UPW && u0 + (int)threadIdx.x < H) {
        const int u = u0 + threadIdx.x;
        p_base = r.dh_base ? r.dh_base[u] : 0.f;
        p_gi = r.act[u]; p_gf = r.act[H + u]; p_gg = r.act[2 * H + u]; p_go = r.act[3 * H + u];
        p_c = r.c[u]; p_cprev = r.cprev ? r.cprev[u] : 0.f; p_dc = r.dc[u];
    }
    // both vectors staged with float4 loads issued back to back (H % 4 == 0 and 256-byte aligned rows: checked on the host)
#pragma unroll
    for (int m = 0; m < 2; ++m)
        if (r.vec[m])
            for (int k = threadIdx.x; k < H; k += 256) reinterpret_cast<float4*>(sg + m * 4 * H)[k] = reinterpret_cast<const float4*>(r.vec[m])[k];
    __syncthreads();
    {
        const int i = wave % UPW, m = wave / UPW;       // 4 waves = 2 units x 2 matrices
        const int u = min(u0 + i, H - 1);
        float d = 0.f;
        if (r.vec[m]) d = row_dot(r.WT[m] + (long)u * 4 * H, sg + m * 4 * H, 4 * H, lane);
        if (lane == 0) part[m * UPW + i] = d;
    }
    __syncthreads();
    if (threadIdx.x < UPW && u0 + (int)threadIdx.x < H) {
        const int i = threadIdx.x, u = u0 + i;
        float first = part[i];
        if (r.drop_first) first *= drop_mult(dcfg, (unsigned)(r.t * H + u), 0u, SITE_SST);
        const float dh = p_base + first + part[UPW + i];
        const float gi = p_gi, gf = p_gf, gg = p_gg, go = p_go;
        const float tc = tanhf(p_c);
        const float dcv = dh * go * (1.f - tc * tc) + p_dc;
        r.dg[u] = dcv * gg * gi * (1.f - gi);
        r.dg[H + u] = dcv * p_cprev * gf * (1.f - gf);
        r.dg[2 * H + u] = dcv * gi * (1.f - gg * gg);
        r.dg[3 * H + u] = dh * tc * go * (1.f - go);
        r.dc[u] = dcv * gf;
    }
}

// proposal head epilogue: scores = sigmoid(z) in place
__global__ void sigmoid_kernel(float* __restrict__ z, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) z[i] = 1.f / (1.f + expf(-z[i]));
}
// dz = g_scores * s * (1 - s)
__global__ void sigmoid_bwd_kernel(const float* __restrict__ s, const float* __restrict__ g, float* __restrict__ dz, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dz[i] = g[i] * s[i] * (1.f - s[i]);
}
// weighted BCE of misc/utils.py:78-99:  labels *= masks; w = labels*w0 + (1-labels)*w1 (w0 = 1-w1, per anchor k);
// loss = K * mean_{t,k} w * -(y log p + (1-y) log(1-p)),  p = scores*masks, logs clamped at -100 like torch's BCELoss.
__global__ __launch_bounds__(1024) void tap_bce_fwd_kernel(const float* __restrict__ scores, const float* __restrict__ masks,
                                                           const float* __restrict__ labels, const float* __restrict__ w1,
                                                           float* __restrict__ loss, int T, int K) {
    // one workgroup of 16 waves (the sum stays in one fixed order: a scalar output that the tests compare bit for bit run to run);
    // four independent elements per thread and pass keep the logf latency covered
    __shared__ float red[16];
    const long n = (long)T * K;
    float s4[4] = {0.f, 0.f, 0.f, 0.f};
    for (long i0 = threadIdx.x; i0 < n; i0 += 4096) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long i = i0 + 1024 * u;
            if (i < n) {
                const int k = (int)(i % K);
                const float y = labels[i] * masks[i], p = scores[i] * masks[i];
                const float w = y * (1.f - w1[k]) + (1.f - y) * w1[k];
                s4[u] -= w * (y * fmaxf(logf(p), -100.f) + (1.f - y) * fmaxf(logf(1.f - p), -100.f));
            }
        }
    }
    float s = wave_sum((s4[0] + s4[1]) + (s4[2] + s4[3]));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < 16; ++w) t += red[w];
        loss[0] = t / (float)n * (float)K;
    }
}
// the same sum in two launches (64 workgroups of partial sums in one fixed order each, then one wave adds the 64 partials in index order):
// bit-reproducible like the single-workgroup form, 6x faster at T x K = 256 x 256
constexpr int BCE_PARTS = 64;
__global__ __launch_bounds__(256) void tap_bce_part_kernel(const float* __restrict__ scores, const float* __restrict__ masks,
                                                           const float* __restrict__ labels, const float* __restrict__ w1,
                                                           float* __restrict__ part, int T, int K) {
    __shared__ float red[4];
    const long n = (long)T * K;
    const long per = (n + BCE_PARTS - 1) / BCE_PARTS, i0 = per * blockIdx.x, i1 = min(n, i0 + per);
    float s4[4] = {0.f, 0.f, 0.f, 0.f};
    for (long b = i0 + threadIdx.x; b < i1; b += 1024) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long i = b + 256 * u;
            if (i < i1) {
                const int k = (int)(i % K);
                const float y = labels[i] * masks[i], p = scores[i] * masks[i];
                const float w = y * (1.f - w1[k]) + (1.f - y) * w1[k];
                s4[u] -= w * (y * fmaxf(logf(p), -100.f) + (1.f - y) * fmaxf(logf(1.f - p), -100.f));
            }
        }
    }
    const float s = wave_sum((s4[0] + s4[1]) + (s4[2] + s4[3]));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ __launch_bounds__(64) void tap_bce_final_kernel(const float* __restrict__ part, float* __restrict__ loss, long n, int K) {
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int i = 0; i < BCE_PARTS; ++i) t += part[i];
        loss[0] = t / (float)n * (float)K;
    }
}
__global__ void tap_bce_bwd_kernel(const float* __restrict__ scores, const float* __restrict__ masks, const float* __restrict__ labels,
                                   const float* __restrict__ w1, const float* __restrict__ g_loss, float* __restrict__ g_scores, int T, int K) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)T * K) return;
    const int k = (int)(i % K);
    const float m = masks[i], y = labels[i] * m, p = scores[i] * m;
    const float w = y * (1.f - w1[k]) + (1.f - y) * w1[k];
    // d/dp of -(y log p + (1-y) log(1-p)) with torch's clamp: the clamped branch has zero slope
    float d = 0.f;
    if (logf(p) > -100.f) d -= y / p;
    if (logf(1.f - p) > -100.f) d += (1.f - y) / (1.f - p);
    g_scores[i] = g_loss[0] * w * d * m / (float)T;       // K / (T*K)
}

// =====================================================================================================================
// PERSISTENT form of the two wavefronts (H = 512): ONE launch per direction instead of T + 1 dependent launches.
//   * 64 workgroups x 256 threads, a workgroup owns 8 hidden units of BOTH layers; its slices of W_hh0, W_ih1 and W_hh1
//     (3 x 32 rows x 512: 192 KB) live in REGISTERS for all T steps (192 VGPRs per thread, one wave per SIMD);
//   * the exchanged vectors ARE the saved activations: h0 / dropped h0 / h1 rows (forward) and the gate-gradient rows dG0 / dG1
//     (reverse) are pre-filled with a sentinel bit pattern (0xFFFFFFFF, a NaN no fp32 operation produces), producers overwrite their
//     elements with write-through stores, consumers POLL the rows they need until no sentinel word is left -- one memory round trip
//     per step (no counters, no second "payload" fetch), each word validated on its own, so a torn read can only look "not ready";
//   * every spin is bounded like the decoder's persistent kernels (abort word + host flag, csrc/persist.hip).
// A step is: poll (h0(k-1), h0d(k-1), h1(k-2): 6 KB / dG1, dG0: 16 KB) -> LDS -> register GEMV slices + DPP reductions -> cell math -> publish.
// =====================================================================================================================
namespace {

typedef unsigned u32s;
typedef u32s u32x4s __attribute__((ext_vector_type(4)));
constexpr int SP_WG = 64, SP_U = 8, SP_H = 512;
constexpr u32s SP_SENT = 0xFFFFFFFFu;
constexpr u32s SP_SPIN_DEFAULT = 4000000u;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t sp_rsrc(const void* p, u32s bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
__device__ __forceinline__ u32x4s sp_ld16(__amdgpu_buffer_rsrc_t r, u32s off) { return __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 16); }   // sc1
__device__ __forceinline__ void sp_store(float* p, float v) {
    if (__float_as_uint(v) == SP_SENT) v = __uint_as_float(0x7FC00000u);          // never publish the sentinel pattern itself
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <int CTRL> __device__ __forceinline__ float sp_dpp(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float sp_sum16(float v) {       // 16-lane DPP row: quad xor 1, quad xor 2, row_half_mirror, row_mirror
    v += sp_dpp<0xB1>(v); v += sp_dpp<0x4E>(v); v += sp_dpp<0x141>(v); v += sp_dpp<0x140>(v);
    return v;
}
__device__ __forceinline__ float sp_sum32(float v) {       // + the neighbouring row (lane ^ 16): v_permlane16_swap of a register with itself
    v = sp_sum16(v);
    const u32s u = __float_as_uint(v);
    const auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    return __uint_as_float(a[0]) + __uint_as_float(a[1]);
}
__device__ __forceinline__ float sp_sum64(float v) {
    v = sp_sum32(v);
    const u32s u = __float_as_uint(v);
    const auto a = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return __uint_as_float(a[0]) + __uint_as_float(a[1]);
}

struct SpSync { u32s* abort_word; u32s* host_flag; u32s spin_limit; };

// every thread: wait until each of its (needed) 16-byte pieces holds no sentinel word.  The loads of one sweep are issued together, so a
// sweep costs ONE memory round trip however many pieces the thread owns (two sweeps in flight were measured: slower -- the straggling loads
// of the second sweep delay the consumer).  false = give up (abort raised here or elsewhere)
// `between` runs once, after the first sweep's loads have been issued and before they are waited for (index-only work and prefetches
// of the step overlap the round trip).
template <int NP, typename F>
__device__ __forceinline__ bool sp_poll(const __amdgpu_buffer_rsrc_t (&r)[NP], const u32s (&off)[NP], const bool (&need)[NP], float4 (&out)[NP],
                                        const SpSync& y, u32s code, F&& between) {
    bool done[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) done[i] = !need[i];
    u32s spins = 0;
    bool first = true;
    for (;;) {
        u32x4s v[NP];
#pragma unroll
        for (int i = 0; i < NP; ++i) if (!done[i]) v[i] = sp_ld16(r[i], off[i]);
        if (first) { between(); first = false; }
        bool all = true;
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            if (done[i]) continue;
            if (v[i].x != SP_SENT && v[i].y != SP_SENT && v[i].z != SP_SENT && v[i].w != SP_SENT) {
                out[i] = make_float4(__uint_as_float(v[i].x), __uint_as_float(v[i].y), __uint_as_float(v[i].z), __uint_as_float(v[i].w));
                done[i] = true;
            } else all = false;
        }
        if (all) return true;
        if ((++spins & 31) == 0) {
            if (__hip_atomic_load(y.abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return false;
            if (spins > y.spin_limit) {
                __hip_atomic_store(y.abort_word, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(y.host_flag, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                return false;
            }
        }
        __builtin_amdgcn_s_sleep(1);
    }
}

#define SP_STAMP(i) do { if (P.stamps && blockIdx.x == 0 && tid == 0 && k < 256) P.stamps[k * 16 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)

struct SstPF {
    int T;
    const float *w_hh0, *w_ih1, *w_hh1, *b_ih1, *b_hh1, *GIN0;
    float *HS0, *H0D, *TAP;                  // [T,H] each: exchange rows = saved activations (sentinel-filled before the launch)
    float *ACT0, *ACT1, *CS0, *CS1;
    SpSync y;
    DropCfg dc;
    unsigned long long* stamps;      // diagnostic (null = off): [step][16] s_memrealtime stamps of workgroup 0
};

typedef float f2s __attribute__((ext_vector_type(2)));
// acc += w . x as two packed FMAs (v_pk_fma_f32: both halves of a float4 pair per instruction)
__device__ __forceinline__ void sp_dot4(f2s& acc, const float4& w, const float4& x) {
    acc = __builtin_elementwise_fma(f2s{w.x, w.y}, f2s{x.x, x.y}, acc);
    acc = __builtin_elementwise_fma(f2s{w.z, w.w}, f2s{x.z, x.w}, acc);
}

__global__ __launch_bounds__(256, 1) void sst_persist_fwd_kernel(SstPF P) {
    __shared__ __attribute__((aligned(16))) float sv[3 * SP_H];      // h0(k-1) | h0d(k-1) | h1(k-2)
    __shared__ int fail;
    constexpr int H = SP_H;
    const int tid = threadIdx.x, G = tid >> 5, kp = tid & 31;
    const int u = blockIdx.x * SP_U + G;                              // this 32-lane group's hidden unit
    const int T = P.T;
    if (tid == 0) fail = 0;
    // register-resident weight slices: gate g, k = 4 kp + 128 j (+0..3)
    float4 W0[4][4], W1[4][8];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const long row = (long)(g * H + u) * H;
#pragma unroll
        for (int j = 0; j < 4; ++j) W0[g][j] = *reinterpret_cast<const float4*>(P.w_hh0 + row + 4 * kp + 128 * j);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = 4 * kp + 128 * j;
            W1[g][j] = k < H ? *reinterpret_cast<const float4*>(P.w_ih1 + row + k) : *reinterpret_cast<const float4*>(P.w_hh1 + row + k - H);
        }
    }
    // lane 0 of the group finishes layer 0's cell of the step, lane 1 layer 1's (the same instructions on different data)
    const bool fin1 = kp == 1;
    float bias1[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) bias1[g] = P.b_ih1[g * H + u] + P.b_hh1[g * H + u];
    float cst = 0.f;                                                  // lane 0: c0, lane 1: c1
    if (tid < 128) reinterpret_cast<float4*>(sv)[256 + tid] = make_float4(0.f, 0.f, 0.f, 0.f);       // h1(-1) = 0 (no row is polled for it)
    __syncthreads();
    const u32s ROWB = H * 4;
    for (int k = 0; k <= T; ++k) {
        const bool l0 = k < T, l1 = k >= 1;
        float gin[4] = {0.f, 0.f, 0.f, 0.f};
        float mk = 1.f;
        auto prefetch = [&]() {
            if (l0 && kp == 0) {
#pragma unroll
                for (int g = 0; g < 4; ++g) gin[g] = P.GIN0[(long)k * 4 * H + g * H + u];
            }
            if (l0) mk = drop_mult(P.dc, (unsigned)(k * H + u), 0u, SITE_SST);
        };
        // index-only work and prefetches FIRST: a row becomes visible ~0.3 us after its producers' stores were issued, so a sweep issued right
        // after this workgroup's own publish would miss and cost a second round trip (measured: 0.28 + 0.56 us this way, 1.7 us the other)
        prefetch();
        SP_STAMP(0);
        // ---- poll the rows published by the previous step: 384 float4 over 256 threads ----
        if (k >= 1) {
            // threads 0..127: h0(k-1) and, from step 2 on, h1(k-2); threads 128..255: h0d(k-1)
            const __amdgpu_buffer_rsrc_t r[2] = {sp_rsrc((tid < 128 ? P.HS0 : P.H0D) + (long)(k - 1) * H, ROWB), sp_rsrc(P.TAP + (long)max(k - 2, 0) * H, ROWB)};
            const u32s off[2] = {(u32s)((tid & 127) * 16), (u32s)((tid & 127) * 16)};
            const bool need[2] = {true, k >= 2 && tid < 128};
            float4 v[2];
            if (sp_poll<2>(r, off, need, v, P.y, 8000u + (u32s)(k % 1000), [] {})) {
                reinterpret_cast<float4*>(sv)[tid] = v[0];
                if (need[1]) reinterpret_cast<float4*>(sv)[256 + tid] = v[1];
            } else fail = 1;
        }
        SP_STAMP(1);
        __syncthreads();
        if (fail) return;
        SP_STAMP(2);
        float s0[4] = {0.f, 0.f, 0.f, 0.f}, s1[4] = {0.f, 0.f, 0.f, 0.f};
        if (k >= 1) {
            const float4* v4 = reinterpret_cast<const float4*>(sv);
            f2s a0[4], a1[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) { a0[g] = f2s{0.f, 0.f}; a1[g] = f2s{0.f, 0.f}; }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float4 x = v4[kp + 32 * j];
#pragma unroll
                for (int g = 0; g < 4; ++g) sp_dot4(a0[g], W0[g][j], x);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float4 x = v4[128 + kp + 32 * j];
#pragma unroll
                for (int g = 0; g < 4; ++g) sp_dot4(a1[g], W1[g][j], x);
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) { s0[g] = sp_sum32(a0[g].x + a0[g].y); s1[g] = sp_sum32(a1[g].x + a1[g].y); }
        }
        SP_STAMP(3);
        // (no barrier here: the next step's rows are complete only after every wave of THIS workgroup has published, i.e. after its reads of sv)
        // ---- cell math: lane 0 layer 0 at step k, lane 1 layer 1 at step k - 1 ----
        {
            const float pi = fin1 ? s1[0] + bias1[0] : s0[0] + gin[0], pf = fin1 ? s1[1] + bias1[1] : s0[1] + gin[1];
            const float pg = fin1 ? s1[2] + bias1[2] : s0[2] + gin[2], po = fin1 ? s1[3] + bias1[3] : s0[3] + gin[3];
            const float gi = fast_sigmoid(pi), gf = fast_sigmoid(pf), gg = fast_tanh(pg), go = fast_sigmoid(po);
            const float c = gf * cst + gi * gg;
            const float h = go * fast_tanh(c);
            if (kp == 0 && l0) {
                cst = c;
                sp_store(P.HS0 + (long)k * H + u, h);
                sp_store(P.H0D + (long)k * H + u, h * mk);
                float* act = P.ACT0 + (long)k * 4 * H + u;
                act[0] = gi; act[H] = gf; act[2 * H] = gg; act[3 * H] = go;
                P.CS0[(long)k * H + u] = c;
            } else if (fin1 && l1) {
                const int t = k - 1;
                cst = c;
                sp_store(P.TAP + (long)t * H + u, h);
                float* act = P.ACT1 + (long)t * 4 * H + u;
                act[0] = gi; act[H] = gf; act[2 * H] = gg; act[3 * H] = go;
                P.CS1[(long)t * H + u] = c;
            }
        }
        SP_STAMP(4);
    }
}

struct SstPB {
    int T;
    const float *w_hh0, *w_ih1, *w_hh1;
    const float *ACT0, *ACT1, *CS0, *CS1, *DHO;
    float *DG0, *DG1;                        // [T,4H] each: exchange rows = the gate gradients the weight-gradient GEMMs read afterwards
    SpSync y;
    DropCfg dc;
    unsigned long long* stamps;
};

// reverse wavefront: step k = layer 1 at t = T-1-k  ||  layer 0 at t0 = T-k.  Wave w owns units 2w, 2w+1 of the workgroup's eight: the
// columns u of W_hh1, W_ih1 and W_hh0 (2048 long each) are split over the 64 lanes, r = 4 lane + 256 j (+0..3).
__global__ __launch_bounds__(256, 1) void sst_persist_bwd_kernel(SstPB P) {
    extern __shared__ __attribute__((aligned(16))) float sdyn[];
    float* sg = sdyn;                                                 // [2][4H]: dG1(T-k) | dG0(T-k+1)
    float* sstage = sdyn + 2 * 4 * SP_H;                              // [8][4H]: set-up staging of one matrix' columns
    __shared__ int fail;
    constexpr int H = SP_H, H4 = 4 * SP_H;
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    const int T = P.T;
    if (tid == 0) fail = 0;
    // register-resident columns, staged through LDS one matrix at a time: the workgroup's eight columns are 32 contiguous bytes of every
    // row, so rows are read as float4 pairs (coalesced over rows would be 2 KB apart) and transposed into [unit][r] in LDS
    float4 A1[2][8], A2[2][8], A3[2][8];
    {
        float* stage = sstage;                                        // [8 units][2048]
        const float* Ws[3] = {P.w_hh1, P.w_ih1, P.w_hh0};
#pragma unroll
        for (int mtx = 0; mtx < 3; ++mtx) {
            for (int idx = tid; idx < H4 * 2; idx += 256) {
                const int row = idx >> 1, half = idx & 1;
                const float4 v = *reinterpret_cast<const float4*>(Ws[mtx] + (long)row * H + blockIdx.x * SP_U + 4 * half);
                stage[(4 * half + 0) * H4 + row] = v.x; stage[(4 * half + 1) * H4 + row] = v.y;
                stage[(4 * half + 2) * H4 + row] = v.z; stage[(4 * half + 3) * H4 + row] = v.w;
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float4 v = *reinterpret_cast<const float4*>(stage + (2 * w + i) * H4 + 4 * lane + 256 * j);
                    if (mtx == 0) A1[i][j] = v; else if (mtx == 1) A2[i][j] = v; else A3[i][j] = v;
                }
            __syncthreads();
        }
    }
    // lanes 0..3 of the wave finish one cell each: unit 2w + (lane & 1), lanes 0 / 1 layer 1 at step t, lanes 2 / 3 layer 0 at step t0
    const int ui = lane & 1;
    const bool fl0 = (lane & 2) != 0;
    const int u = blockIdx.x * SP_U + 2 * w + ui;
    float dcst = 0.f;                                                 // d c carried to the previous timestep (this lane's cell)
    __syncthreads();
    const u32s ROWB = H4 * 4;
    for (int k = 0; k <= T; ++k) {
        const bool l1 = k < T, l0 = k >= 1;
        const int t = T - 1 - k, t0 = T - k;
        const bool act_on = lane < 4 && (fl0 ? l0 : l1);
        const int tt = fl0 ? t0 : t;
        // saved activations of the cell this lane finishes: issued behind the first sweep of the wait, consumed after it
        float ac[4] = {0.f, 0.f, 0.f, 0.f}, cc = 0.f, cp = 0.f, dho = 0.f, mk = 1.f;
        auto prefetch = [&]() {
            if (act_on) {
                const float* A = fl0 ? P.ACT0 : P.ACT1;
                const float* Cs = fl0 ? P.CS0 : P.CS1;
#pragma unroll
                for (int g = 0; g < 4; ++g) ac[g] = A[(long)tt * H4 + g * H + u];
                cc = Cs[(long)tt * H + u]; cp = tt ? Cs[(long)(tt - 1) * H + u] : 0.f;
                if (!fl0) dho = P.DHO[(long)tt * H + u];
            }
            if (l0) mk = drop_mult(P.dc, (unsigned)(t0 * H + u), 0u, SITE_SST);
        };
        prefetch();          // ahead of the wait, like the forward kernel
        SP_STAMP(0);
        if (k >= 1) {
            const __amdgpu_buffer_rsrc_t r1 = sp_rsrc(P.DG1 + (long)t0 * H4, ROWB), r0 = sp_rsrc(P.DG0 + (long)min(t0 + 1, T - 1) * H4, ROWB);
            const __amdgpu_buffer_rsrc_t r[4] = {r1, r1, r0, r0};
            const u32s off[4] = {(u32s)(tid * 16), (u32s)((tid + 256) * 16), (u32s)(tid * 16), (u32s)((tid + 256) * 16)};
            const bool need[4] = {true, true, k >= 2, k >= 2};
            float4 v[4];
            if (sp_poll<4>(r, off, need, v, P.y, 8500u + (u32s)(k % 500), [] {})) {
                reinterpret_cast<float4*>(sg)[tid] = v[0];
                reinterpret_cast<float4*>(sg)[tid + 256] = v[1];
                if (k >= 2) { reinterpret_cast<float4*>(sg)[512 + tid] = v[2]; reinterpret_cast<float4*>(sg)[768 + tid] = v[3]; }
            } else fail = 1;
        }
        SP_STAMP(1);
        __syncthreads();
        if (fail) return;
        SP_STAMP(2);
        float d1[2] = {0.f, 0.f}, d2[2] = {0.f, 0.f}, d3[2] = {0.f, 0.f};
        if (k >= 1) {
            const float4* g4 = reinterpret_cast<const float4*>(sg);
            f2s b1[2], b2[2], b3[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) b1[i] = b2[i] = b3[i] = f2s{0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float4 x = g4[lane + 64 * j];
#pragma unroll
                for (int i = 0; i < 2; ++i) { sp_dot4(b1[i], A1[i][j], x); sp_dot4(b2[i], A2[i][j], x); }
            }
            if (k >= 2) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float4 x = g4[512 + lane + 64 * j];
#pragma unroll
                    for (int i = 0; i < 2; ++i) sp_dot4(b3[i], A3[i][j], x);
                }
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) { d1[i] = sp_sum64(b1[i].x + b1[i].y); d2[i] = sp_sum64(b2[i].x + b2[i].y); d3[i] = sp_sum64(b3[i].x + b3[i].y); }
        }
        SP_STAMP(3);
        // (no barrier: a row of the next step completes only after every wave of this workgroup has published, i.e. after its reads of sg)
        {
            const float e1 = ui ? d1[1] : d1[0], e2 = ui ? d2[1] : d2[0], e3 = ui ? d3[1] : d3[0];
            const float dh = fl0 ? e2 * mk + e3 : dho + e1;
            const float gi = ac[0], gf = ac[1], gg = ac[2], go = ac[3];
            const float tc = fast_tanh(cc);
            const float dcv = dh * go * (1.f - tc * tc) + dcst;
            if (act_on) {
                float* dg = (fl0 ? P.DG0 : P.DG1) + (long)tt * H4 + u;
                sp_store(dg, dcv * gg * gi * (1.f - gi));
                sp_store(dg + H, dcv * cp * gf * (1.f - gf));
                sp_store(dg + 2 * H, dcv * gi * (1.f - gg * gg));
                sp_store(dg + 3 * H, dh * tc * go * (1.f - go));
                dcst = dcv * gf;
            }
        }
        SP_STAMP(4);
    }
}

}  // namespace

static inline long rup(long x, long a) { return (x + a - 1) / a * a; }

struct SstWs { float *GIN0, *ACT[2], *HS[2], *CS[2], *H0D; long total; };
static SstWs carve(int T, int D, int H, int K, float* base) {
    SstWs w;
    long off = 0;
    auto take = [&](long n) { float* p = base ? base + off : nullptr; off += rup(n, 64); return p; };
    w.GIN0 = take((long)T * 4 * H);        // layer 0's input-side pre-activations (layer 1 forms its own inside the wavefront step)
    w.HS[0] = take((long)T * H); w.H0D = take((long)T * H);          // adjacent: the persistent kernel's exchange rows, one sentinel fill
    w.HS[1] = take((long)T * H);
    for (int l = 0; l < 2; ++l) { w.ACT[l] = take((long)T * 4 * H); w.CS[l] = take((long)T * H); }
    w.total = off;
    return w;
}
struct SstWsB { float *DG[2], *DHO, *DC[2], *DZ, *WT[2], *WT_IH1; long total; };
static SstWsB carve_b(int T, int D, int H, int K, float* base) {
    SstWsB w;
    long off = 0;
    auto take = [&](long n) { float* p = base ? base + off : nullptr; off += rup(n, 64); return p; };
    for (int l = 0; l < 2; ++l) w.DG[l] = take((long)T * 4 * H);      // adjacent: the persistent kernel's exchange rows, one sentinel fill
    for (int l = 0; l < 2; ++l) w.WT[l] = take((long)H * 4 * H);
    w.DHO = take((long)T * H); w.DC[0] = take(H); w.DC[1] = take(H); w.DZ = take((long)T * K);
    w.WT_IH1 = take((long)H * 4 * H);
    w.total = off;
    return w;
}

#define RC(x) do { int _rc = (x); if (_rc) return _rc; } while (0)

// the persistent form needs H = 512 (64 workgroups x 8 units) and the abort plumbing of csrc/persist.hip
static bool sst_persist_ok(int H) {
    static int cus = -1;
    if (cus < 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 0;
    }
    return config().sst_persist && H == SP_H && cus >= SP_WG && persist_abort_word() && persist_host_flag();
}
static SpSync sst_sync() {
    SpSync y;
    y.abort_word = const_cast<u32s*>(persist_abort_word());
    y.host_flag = persist_host_flag();
    y.spin_limit = config().persist_spin_limit > 0 ? (u32s)config().persist_spin_limit : SP_SPIN_DEFAULT;
    return y;
}
template <typename K, typename P>
static int sst_persist_launch(K kernel, P& args, const char* what, hipStream_t st, size_t lds = 0) {
    if (lds > 48 * 1024) {
        static bool attr_set = false;
        if (!attr_set) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
                set_error("%s: cannot reserve %zu bytes of LDS", what, lds);
                return -5;
            }
            attr_set = true;
        }
    }
    if (config().persist_coop) {          // shared device: start only when all 64 workgroups can be resident (see persist.hip)
        void* kargs[1] = {&args};
        if (hipLaunchCooperativeKernel(reinterpret_cast<const void*>(kernel), dim3(SP_WG), dim3(256), kargs, lds, st) == hipSuccess) return 0;
        coop_refused(what, hipGetErrorString(hipGetLastError()));
    }
    hipLaunchKernelGGL(kernel, dim3(SP_WG), dim3(256), lds, st, args);
    return check_launch(what);
}

}  // namespace echr

using namespace echr;

extern "C" int64_t echr_sst_ws_floats(int32_t T, int32_t D, int32_t H, int32_t K) { return carve(T, D, H, K, nullptr).total; }
extern "C" int64_t echr_sst_ws_bwd_floats(int32_t T, int32_t D, int32_t H, int32_t K) { return carve_b(T, D, H, K, nullptr).total; }

static int sst_check(const echr_sst_args* a, const char* who) {
    ECHR_REQUIRE(a, "%s: null args", who);
    ECHR_REQUIRE(a->T > 0 && a->D > 0 && a->H > 0 && a->K > 0 && a->H % 4 == 0 && a->H <= 4096, "%s: bad dims (H must be a multiple of 4)", who);
    ECHR_REQUIRE(a->x && a->ws && a->tap_feats && a->scores, "%s: missing buffers", who);
    for (int l = 0; l < 2; ++l) ECHR_REQUIRE(a->w_ih[l] && a->w_hh[l] && a->b_ih[l] && a->b_hh[l], "%s: missing LSTM parameters", who);
    ECHR_REQUIRE(a->w_sc && a->b_sc, "%s: missing head parameters", who);
    return 0;
}

static int sst_head_impl(const echr_sst_args* a, hipStream_t st);
static int sst_fwd_impl(const echr_sst_args* a, const echr_dropout* drop, void* stream, bool with_head);
extern "C" int echr_sst_fwd(const echr_sst_args* a, const echr_dropout* drop, void* stream) { return sst_fwd_impl(a, drop, stream, true); }
// the two halves of echr_sst_fwd on their own (joint iteration: the caption side waits for tap_feats alone, so its caller queues the proposal
// head -- scores = sigmoid(tap_feats . W_sc^T + b_sc), models/sst_model.py:38-39 -- BEHIND the caption call instead of in front of it)
extern "C" int echr_sst_fwd_states(const echr_sst_args* a, const echr_dropout* drop, void* stream) { return sst_fwd_impl(a, drop, stream, false); }
extern "C" int echr_sst_head_fwd(const echr_sst_args* a, void* stream) {
    RC(sst_check(a, "sst_head_fwd"));
    return sst_head_impl(a, (hipStream_t)stream);
}
static int sst_fwd_impl(const echr_sst_args* a, const echr_dropout* drop, void* stream, bool with_head) {
    RC(sst_check(a, "sst_fwd"));
    hipStream_t st = (hipStream_t)stream;
    const int T = a->T, D = a->D, H = a->H, K = a->K;
    SstWs w = carve(T, D, H, K, a->ws);
    const DropCfg dc = make_drop(drop, a->p_drop);
    const int nwg = (H + UPW - 1) / UPW;
    // layer 0's input-side pre-activations for all T rows: X . W_ih0^T + b_ih0 + b_hh0 (batched MFMA GEMM)
    echr_gemm_desc d0 = desc_nt(a->x, D, a->w_ih[0], D, w.GIN0, 4 * H, T, 4 * H, D);
    d0.bias = a->b_ih[0]; d0.bias2 = a->b_hh[0]; d0.split_k = T >= 128 ? 1 : -1;      // T x 4H output tiles fill the chip from T = 128 on: no split, no zero fill
    RC(gemm(d0, st));
    if (sst_persist_ok(H)) {
        RC(persist_check_async());
        // exchange rows = saved activations: sentinel-fill h0 | h0d (adjacent) and the output rows h1 = tap_feats, then ONE launch
        if (hipMemsetAsync(w.HS[0], 0xFF, sizeof(float) * (size_t)((w.H0D - w.HS[0]) + (long)T * H), st) != hipSuccess ||
            hipMemsetAsync(a->tap_feats, 0xFF, sizeof(float) * (size_t)T * H, st) != hipSuccess) { set_error("sst_fwd: memset failed"); return -5; }
        SstPF P;
        P.T = T; P.w_hh0 = a->w_hh[0]; P.w_ih1 = a->w_ih[1]; P.w_hh1 = a->w_hh[1]; P.b_ih1 = a->b_ih[1]; P.b_hh1 = a->b_hh[1]; P.GIN0 = w.GIN0;
        P.HS0 = w.HS[0]; P.H0D = w.H0D; P.TAP = a->tap_feats; P.ACT0 = w.ACT[0]; P.ACT1 = w.ACT[1]; P.CS0 = w.CS[0]; P.CS1 = w.CS[1];
        P.y = sst_sync(); P.dc = dc;
        P.stamps = config().persist_stamps == 3 ? persist_stamp_buffer(T + 1, st) : nullptr;
        // algorithmic bytes: the three recurrent matrices once + per step the pre-activation row in and gates / cells / outputs of both layers out
        ProfScope prof(PROF_SST, 2.0 * T * 3.0 * 4 * H * H, 4.0 * (3.0 * 4 * H * H + (double)T * (4.0 * H + 2 * (4.0 * H + 2.0 * H) + H)), st);
        RC(sst_persist_launch(sst_persist_fwd_kernel, P, "sst_persist_fwd", st));
    } else {
    // wavefront: launch k = layer 0 at step k  ||  layer 1 at step k-1 (reads the dropped layer-0 output of step k-1)
        for (int k = 0; k <= T; ++k) {
            SstFwdRole r0{}, r1{};
            if (k < T) {
                r0.active = 1; r0.t = k;
                r0.W[0] = a->w_hh[0]; r0.v[0] = k ? w.HS[0] + (long)(k - 1) * H : nullptr;
                r0.base = w.GIN0 + (long)k * 4 * H;
                r0.cprev = k ? w.CS[0] + (long)(k - 1) * H : nullptr;
                r0.act = w.ACT[0] + (long)k * 4 * H; r0.hout = w.HS[0] + (long)k * H; r0.cout = w.CS[0] + (long)k * H;
                r0.hdrop = w.H0D + (long)k * H;
            }
            if (k >= 1) {
                const int t = k - 1;
                r1.active = 1; r1.t = t;
                r1.W[0] = a->w_ih[1]; r1.v[0] = w.H0D + (long)t * H;
                r1.W[1] = a->w_hh[1]; r1.v[1] = t ? a->tap_feats + (long)(t - 1) * H : nullptr;
                r1.base = a->b_ih[1]; r1.base2 = a->b_hh[1];
                r1.cprev = t ? w.CS[1] + (long)(t - 1) * H : nullptr;
                r1.act = w.ACT[1] + (long)t * 4 * H; r1.hout = a->tap_feats + (long)t * H; r1.cout = w.CS[1] + (long)t * H;
            }
            hipLaunchKernelGGL(sst_wave_fwd_kernel, dim3(nwg, 2), dim3(256), (2 * H + 4 * UPW) * sizeof(float), st, r0, r1, H, dc);
        }
        RC(check_launch("sst_wave_fwd"));
    }
    return with_head ? sst_head_impl(a, st) : 0;
}
static int sst_head_impl(const echr_sst_args* a, hipStream_t st) {
    const int T = a->T, H = a->H, K = a->K;
    // proposal head
    echr_gemm_desc d = desc_nt(a->tap_feats, H, a->w_sc, H, a->scores, K, T, K, H);
    d.bias = a->b_sc; d.split_k = -1;
    RC(gemm(d, st));
    const long n = (long)T * K;
    hipLaunchKernelGGL(sigmoid_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, a->scores, n);
    return check_launch("sst_head");
}

extern "C" int echr_sst_bwd(const echr_sst_args* a, const echr_sst_grads* g, const echr_dropout* drop, void* stream) {
    RC(sst_check(a, "sst_bwd"));
    ECHR_REQUIRE(g && g->ws_bwd && (g->g_tap || g->g_scores), "sst_bwd: missing buffers");
    hipStream_t st = (hipStream_t)stream;
    const int T = a->T, D = a->D, H = a->H, K = a->K;
    SstWs w = carve(T, D, H, K, a->ws);
    SstWsB b = carve_b(T, D, H, K, g->ws_bwd);
    const DropCfg dc = make_drop(drop, a->p_drop);
    const int nwg = (H + UPW - 1) / UPW;
    const bool z = g->zeroed != 0;             // gradient buffers arrive zero-filled (flat arena): accumulate, no per-product fills
    const float zb = z ? 1.f : 0.f;
    echr_gemm_desc d;
    // d tap_feats = g_tap + (g_scores * s(1-s)) . W_sc ; head parameter gradients
    if (g->g_tap) RC(hipMemcpyAsync(b.DHO, g->g_tap, sizeof(float) * T * H, hipMemcpyDeviceToDevice, st) == hipSuccess ? 0 : -5);
    else RC(fill_zero(b.DHO, (long)T * H, st));
    if (g->g_scores) {
        const long n = (long)T * K;
        hipLaunchKernelGGL(sigmoid_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, a->scores, g->g_scores, b.DZ, n);
        RC(check_launch("sigmoid_bwd"));
        d = desc_nn(b.DZ, K, a->w_sc, H, b.DHO, H, T, H, K); d.beta = 1.f; d.split_k = -1;
        RC(gemm(d, st));
        d = desc_tn(b.DZ, K, a->tap_feats, H, g->g_w_sc, H, K, H, T); d.split_k = -1; d.beta = zb;
        RC(gemm(d, st));
        RC(colsum(b.DZ, K, T, K, g->g_b_sc, z, st));
    } else if (!z) {
        RC(fill_zero(g->g_w_sc, (long)K * H, st));
        RC(fill_zero(g->g_b_sc, K, st));
    }
    if (sst_persist_ok(H)) {
        RC(persist_check_async());
        if (hipMemsetAsync(b.DG[0], 0xFF, sizeof(float) * (size_t)((b.DG[1] - b.DG[0]) + (long)T * 4 * H), st) != hipSuccess) { set_error("sst_bwd: memset failed"); return -5; }
        SstPB P;
        P.T = T; P.w_hh0 = a->w_hh[0]; P.w_ih1 = a->w_ih[1]; P.w_hh1 = a->w_hh[1];
        P.ACT0 = w.ACT[0]; P.ACT1 = w.ACT[1]; P.CS0 = w.CS[0]; P.CS1 = w.CS[1]; P.DHO = b.DHO; P.DG0 = b.DG[0]; P.DG1 = b.DG[1];
        P.y = sst_sync(); P.dc = dc;
        P.stamps = config().persist_stamps == 4 ? persist_stamp_buffer(T + 1, st) : nullptr;
        ProfScope prof(PROF_SST, 2.0 * T * 3.0 * 4 * H * H, 4.0 * (3.0 * 4 * H * H + (double)T * (2 * (4.0 * H + 2.0 * H) + H + 2 * 4.0 * H)), st);
        RC(sst_persist_launch(sst_persist_bwd_kernel, P, "sst_persist_bwd", st, sizeof(float) * (2 + 8) * 4 * SP_H));
    } else {
    {
            const TransposeJob tj[3] = {{a->w_hh[0], H, b.WT[0], 4 * H, 4 * H, H}, {a->w_hh[1], H, b.WT[1], 4 * H, 4 * H, H},
                                        {a->w_ih[1], H, b.WT_IH1, 4 * H, 4 * H, H}};
            RC(transpose_multi(tj, 3, st));
            float* zp[2] = {b.DC[0], b.DC[1]};
            const long zn[2] = {H, H};
            RC(fill_zero_multi(zp, zn, 2, st));
        }
        // wavefront: launch k = layer 1 at step T-1-k  ||  layer 0 at step T-k (its upstream gradient is W_ih1^T . dG1 of the same step,
        // through the inter-layer dropout mask)
        for (int k = 0; k <= T; ++k) {
            SstBwdRole r1{}, r0{};
            if (k < T) {
                const int t = T - 1 - k;
                r1.active = 1; r1.t = t;
                r1.WT[0] = b.WT[1]; r1.vec[0] = t + 1 < T ? b.DG[1] + (long)(t + 1) * 4 * H : nullptr;
                r1.dh_base = b.DHO + (long)t * H;
                r1.act = w.ACT[1] + (long)t * 4 * H; r1.c = w.CS[1] + (long)t * H; r1.cprev = t ? w.CS[1] + (long)(t - 1) * H : nullptr;
                r1.dc = b.DC[1]; r1.dg = b.DG[1] + (long)t * 4 * H;
            }
            if (k >= 1) {
                const int t = T - k;
                r0.active = 1; r0.t = t; r0.drop_first = 1;
                r0.WT[0] = b.WT_IH1; r0.vec[0] = b.DG[1] + (long)t * 4 * H;
                r0.WT[1] = b.WT[0]; r0.vec[1] = t + 1 < T ? b.DG[0] + (long)(t + 1) * 4 * H : nullptr;
                r0.act = w.ACT[0] + (long)t * 4 * H; r0.c = w.CS[0] + (long)t * H; r0.cprev = t ? w.CS[0] + (long)(t - 1) * H : nullptr;
                r0.dc = b.DC[0]; r0.dg = b.DG[0] + (long)t * 4 * H;
            }
            hipLaunchKernelGGL(sst_wave_bwd_kernel, dim3(nwg, 2), dim3(256), (8 * H + 2 * UPW) * sizeof(float), st, r1, r0, H, dc);
        }
        RC(check_launch("sst_wave_bwd"));
    }
    // parameter gradients (sums over the T rows); h(t-1) pairs with dG(t): rows 1..T-1.  The two layers' products are independent small
    // GEMMs (20-25 us each, half a chip): layer 0's run on the library's helper stream beside layer 1's
    hipStream_t s0 = config().tsrm_fork ? aux_fork(st) : nullptr;
    const bool fork = s0 != nullptr;
    if (!fork) s0 = st;
    if (z && T > 1 && fork) {
        // zeroed gradient buffers (the arena path): the two recurrent products share their shape (4H x H over T - 1 rows) -> one grouped launch
        // on this stream beside the two input products on the helper stream, and ONE column-sum launch for both layers' biases
        // (four GEMM + two column-sum launches on two streams before)
        echr_gemm_desc gh[2];
        for (int l = 0; l < 2; ++l) {
            gh[l] = desc_tn(b.DG[l] + 4 * H, 4 * H, l == 0 ? w.HS[0] : a->tap_feats, H, g->g_w_hh[l], H, 4 * H, H, T - 1);
            gh[l].split_k = -1; gh[l].beta = 1.f;
        }
        RC(gemm_grouped(gh, 2, st));
        d = desc_tn(b.DG[0], 4 * H, a->x, D, g->g_w_ih[0], D, 4 * H, D, T); d.split_k = -1; d.beta = 1.f;
        RC(gemm(d, s0));
        d = desc_tn(b.DG[1], 4 * H, w.H0D, H, g->g_w_ih[1], H, 4 * H, H, T); d.split_k = -1; d.beta = 1.f;
        RC(gemm(d, s0));
        const ColsumJob cj[2] = {{b.DG[0], 4 * H, T, 4 * H, g->g_b_ih[0], g->g_b_hh[0], nullptr}, {b.DG[1], 4 * H, T, 4 * H, g->g_b_ih[1], g->g_b_hh[1], nullptr}};
        RC(colsum_multi(cj, 2, st));
        return aux_join(st);
    }
    for (int l = 1; l >= 0; --l) {
        hipStream_t sl = l == 0 ? s0 : st;
        const float* hs = l == 0 ? w.HS[0] : a->tap_feats;
        const float* xin = l == 0 ? a->x : w.H0D;
        const int din = l == 0 ? D : H;
        d = desc_tn(b.DG[l], 4 * H, xin, din, g->g_w_ih[l], din, 4 * H, din, T); d.split_k = -1; d.beta = zb;
        RC(gemm(d, sl));
        if (T > 1) {
            d = desc_tn(b.DG[l] + 4 * H, 4 * H, hs, H, g->g_w_hh[l], H, 4 * H, H, T - 1); d.split_k = -1; d.beta = zb;
            RC(gemm(d, sl));
        } else if (!z) {
            RC(fill_zero(g->g_w_hh[l], (long)4 * H * H, sl));
        }
        RC(colsum2(b.DG[l], 4 * H, T, 4 * H, g->g_b_ih[l], g->g_b_hh[l], z, sl));
    }
    if (fork) RC(aux_join(st));
    return 0;
}

extern "C" int echr_tap_bce_fwd(const float* scores, const float* masks, const float* labels, const float* w1, float* loss, int32_t T,
                                int32_t K, void* stream) {
    ECHR_REQUIRE(scores && masks && labels && w1 && loss && T > 0 && K > 0, "tap_bce_fwd: bad arguments");
    hipLaunchKernelGGL(tap_bce_fwd_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, scores, masks, labels, w1, loss, T, K);
    return check_launch("tap_bce_fwd");
}
extern "C" int echr_tap_bce_fwd_ws(const float* scores, const float* masks, const float* labels, const float* w1, float* loss, float* partials,
                                   int32_t T, int32_t K, void* stream) {
    ECHR_REQUIRE(scores && masks && labels && w1 && loss && partials && T > 0 && K > 0, "tap_bce_fwd_ws: bad arguments");
    hipLaunchKernelGGL(tap_bce_part_kernel, dim3(BCE_PARTS), dim3(256), 0, (hipStream_t)stream, scores, masks, labels, w1, partials, T, K);
    hipLaunchKernelGGL(tap_bce_final_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, partials, loss, (long)T * K, K);
    return check_launch("tap_bce_fwd_ws");
}
extern "C" int echr_tap_bce_bwd(const float* scores, const float* masks, const float* labels, const float* w1, const float* g_loss,
                                float* g_scores, int32_t T, int32_t K, void* stream) {
    ECHR_REQUIRE(scores && masks && labels && w1 && g_loss && g_scores && T > 0 && K > 0, "tap_bce_bwd: bad arguments");
    const long n = (long)T * K;
    hipLaunchKernelGGL(tap_bce_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, scores, masks, labels, w1,
                       g_loss, g_scores, T, K);
    return check_launch("tap_bce_bwd");
}
