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

static inline long rup(long x, long a) { return (x + a - 1) / a * a; }

struct SstWs { float *GIN0, *ACT[2], *HS[2], *CS[2], *H0D; long total; };
static SstWs carve(int T, int D, int H, int K, float* base) {
    SstWs w;
    long off = 0;
    auto take = [&](long n) { float* p = base ? base + off : nullptr; off += rup(n, 64); return p; };
    w.GIN0 = take((long)T * 4 * H);        // layer 0's input-side pre-activations (layer 1 forms its own inside the wavefront step)
    for (int l = 0; l < 2; ++l) { w.ACT[l] = take((long)T * 4 * H); w.HS[l] = take((long)T * H); w.CS[l] = take((long)T * H); }
    w.H0D = take((long)T * H);
    w.total = off;
    return w;
}
struct SstWsB { float *DG[2], *DHO, *DC[2], *DZ, *WT[2], *WT_IH1; long total; };
static SstWsB carve_b(int T, int D, int H, int K, float* base) {
    SstWsB w;
    long off = 0;
    auto take = [&](long n) { float* p = base ? base + off : nullptr; off += rup(n, 64); return p; };
    for (int l = 0; l < 2; ++l) { w.DG[l] = take((long)T * 4 * H); w.WT[l] = take((long)H * 4 * H); }
    w.DHO = take((long)T * H); w.DC[0] = take(H); w.DC[1] = take(H); w.DZ = take((long)T * K);
    w.WT_IH1 = take((long)H * 4 * H);
    w.total = off;
    return w;
}

#define RC(x) do { int _rc = (x); if (_rc) return _rc; } while (0)

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

extern "C" int echr_sst_fwd(const echr_sst_args* a, const echr_dropout* drop, void* stream) {
    RC(sst_check(a, "sst_fwd"));
    hipStream_t st = (hipStream_t)stream;
    const int T = a->T, D = a->D, H = a->H, K = a->K;
    SstWs w = carve(T, D, H, K, a->ws);
    const DropCfg dc = make_drop(drop, a->p_drop);
    const int nwg = (H + UPW - 1) / UPW;
    // layer 0's input-side pre-activations for all T rows: X . W_ih0^T + b_ih0 + b_hh0 (batched MFMA GEMM)
    echr_gemm_desc d0 = desc_nt(a->x, D, a->w_ih[0], D, w.GIN0, 4 * H, T, 4 * H, D);
    d0.bias = a->b_ih[0]; d0.bias2 = a->b_hh[0]; d0.split_k = -1;
    RC(gemm(d0, st));
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
        d = desc_tn(b.DZ, K, a->tap_feats, H, g->g_w_sc, H, K, H, T); d.split_k = -1;
        RC(gemm(d, st));
        RC(colsum(b.DZ, K, T, K, g->g_b_sc, false, st));
    } else {
        RC(fill_zero(g->g_w_sc, (long)K * H, st));
        RC(fill_zero(g->g_b_sc, K, st));
    }
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
    // parameter gradients (sums over the T rows); h(t-1) pairs with dG(t): rows 1..T-1
    for (int l = 1; l >= 0; --l) {
        const float* hs = l == 0 ? w.HS[0] : a->tap_feats;
        const float* xin = l == 0 ? a->x : w.H0D;
        const int din = l == 0 ? D : H;
        d = desc_tn(b.DG[l], 4 * H, xin, din, g->g_w_ih[l], din, 4 * H, din, T); d.split_k = -1;
        RC(gemm(d, st));
        if (T > 1) {
            d = desc_tn(b.DG[l] + 4 * H, 4 * H, hs, H, g->g_w_hh[l], H, 4 * H, H, T - 1); d.split_k = -1;
            RC(gemm(d, st));
        } else {
            RC(fill_zero(g->g_w_hh[l], (long)4 * H * H, st));
        }
        RC(colsum2(b.DG[l], 4 * H, T, 4 * H, g->g_b_ih[l], g->g_b_hh[l], false, st));
    }
    return 0;
}

extern "C" int echr_tap_bce_fwd(const float* scores, const float* masks, const float* labels, const float* w1, float* loss, int32_t T,
                                int32_t K, void* stream) {
    ECHR_REQUIRE(scores && masks && labels && w1 && loss && T > 0 && K > 0, "tap_bce_fwd: bad arguments");
    hipLaunchKernelGGL(tap_bce_fwd_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, scores, masks, labels, w1, loss, T, K);
    return check_launch("tap_bce_fwd");
}
extern "C" int echr_tap_bce_bwd(const float* scores, const float* masks, const float* labels, const float* w1, const float* g_loss,
                                float* g_scores, int32_t T, int32_t K, void* stream) {
    ECHR_REQUIRE(scores && masks && labels && w1 && g_loss && g_scores && T > 0 && K > 0, "tap_bce_bwd: bad arguments");
    const long n = (long)T * K;
    hipLaunchKernelGGL(tap_bce_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, scores, masks, labels, w1,
                       g_loss, g_scores, T, K);
    return check_launch("tap_bce_bwd");
}
