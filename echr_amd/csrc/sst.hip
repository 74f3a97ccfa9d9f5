// SST proposal encoder on gfx950 (reference: models/sst_model.py:5-40 -- nn.LSTM(video_dim -> hidden, 2 layers, batch_first,
// inter-layer dropout) over ONE video [1,T,D], Linear(hidden -> K) + sigmoid) and its weighted-BCE criterion
// (misc/utils.py:78-99).  SURVEY section 8-f row 1: the producer of `tap_feats` for the caption path.
//
// Batch size is 1, so the recurrence is a chain of GEMVs: there is no MFMA shape in it.  Layout of the work:
//   * input-side products of a whole layer are batched over the T rows on the fp32 MFMA GEMM (gemm.hip);
//   * one launch per (layer, timestep): workgroup u owns 8 hidden units, wave g (of 4) computes gate g's 8 rows of
//     W_hh . h(t-1) with float4 lanes + cross-lane reduction, then 8 threads finish the cell (the 4 gates of a unit meet in
//     LDS).  The 4 MB of W_hh are split over 64 workgroups that always land on the same XCDs -> they stay L2-resident
//     across the T launches;
//   * backward mirrors it with W_hh^T: workgroup u first forms d h(t)[its 8 units] = W_hh^T[u-rows] . dG(t+1) (all 4H of
//     dG(t+1) are final from the previous launch) and then runs the cell backward for step t -- again one launch per step;
//   * weight gradients are batched TN GEMMs over the T rows afterwards.
#include "echr_common.h"
#include "echr_internal.h"

namespace echr {

DropCfg make_drop(const echr_dropout* d, float p);
enum { SITE_SST = 5 };
constexpr int UPW = 2;          // hidden units per workgroup (256 workgroups at H = 512: one per CU)

// dot of one weight row with a vector held in LDS; lanes stride the k axis in float4
__device__ __forceinline__ float row_dot(const float* __restrict__ wrow, const float* __restrict__ v, int K, int lane) {
    float acc = 0.f;
    for (int k = lane * 4; k < K; k += 256) {
        const float4 w4 = *reinterpret_cast<const float4*>(wrow + k);
        const float4 x4 = *reinterpret_cast<const float4*>(v + k);
        acc += w4.x * x4.x + w4.y * x4.y + w4.z * x4.z + w4.w * x4.w;
    }
    return wave_sum(acc);
}

// forward cell of one timestep.  gin: [4H] input-side pre-activations (biases included); hprev/cprev: [H] (dropout is
// applied to the layer OUTPUT hdrop only, the recurrence uses the raw h as nn.LSTM does); act out: [4H] activations.
__global__ __launch_bounds__(256) void sst_step_fwd_kernel(const float* __restrict__ Whh, const float* __restrict__ gin,
                                                           const float* __restrict__ hprev, const float* __restrict__ cprev,
                                                           float* __restrict__ act, float* __restrict__ hout, float* __restrict__ cout,
                                                           float* __restrict__ hdrop, int H, int t, DropCfg dc) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* sh = sm;                 // [H]
    float* pre = sm + H;            // [4][UPW]
    const int u0 = blockIdx.x * UPW;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int k = threadIdx.x; k < H; k += 256) sh[k] = hprev ? hprev[k] : 0.f;
    __syncthreads();
    {
        // wave g owns gate g: its UPW rows are multiplied with independent accumulators (loads of all rows in flight together)
        float acc[UPW];
#pragma unroll
        for (int i = 0; i < UPW; ++i) acc[i] = 0.f;
        if (hprev) {
            for (int k = lane * 4; k < H; k += 256) {
                const float4 x4 = *reinterpret_cast<const float4*>(sh + k);
#pragma unroll
                for (int i = 0; i < UPW; ++i) {
                    const int u = min(u0 + i, H - 1);
                    const float4 w4 = *reinterpret_cast<const float4*>(Whh + (long)(wave * H + u) * H + k);
                    acc[i] += w4.x * x4.x + w4.y * x4.y + w4.z * x4.z + w4.w * x4.w;
                }
            }
        }
#pragma unroll
        for (int i = 0; i < UPW; ++i) {
            const float d = wave_sum(acc[i]);
            if (lane == 0 && u0 + i < H) pre[wave * UPW + i] = d + gin[wave * H + u0 + i];
        }
    }
    __syncthreads();
    if (threadIdx.x < UPW && u0 + threadIdx.x < H) {
        const int i = threadIdx.x, u = u0 + i;
        const float gi = fast_sigmoid(pre[i]), gf = fast_sigmoid(pre[UPW + i]), gg = tanhf(pre[2 * UPW + i]), go = fast_sigmoid(pre[3 * UPW + i]);
        const float c = gf * (cprev ? cprev[u] : 0.f) + gi * gg;
        const float h = go * tanhf(c);
        act[u] = gi; act[H + u] = gf; act[2 * H + u] = gg; act[3 * H + u] = go;
        cout[u] = c; hout[u] = h;
        if (hdrop) hdrop[u] = h * drop_mult(dc, (unsigned)(t * H + u), 0u, SITE_SST);
    }
}

// backward cell of one timestep: d h(t) = dh_out[t] (from above) + W_hh^T . dG(t+1); then the cell backward -> dG(t), dc.
// WhhT: [H, 4H] (row u = column u of W_hh).  dgnext: [4H] of step t+1 or null at the last step.
__global__ __launch_bounds__(256) void sst_step_bwd_kernel(const float* __restrict__ WhhT, const float* __restrict__ dgnext,
                                                           const float* __restrict__ dh_out, const float* __restrict__ act,
                                                           const float* __restrict__ c, const float* __restrict__ cprev,
                                                           float* __restrict__ dc, float* __restrict__ dg, int H) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* sg = sm;                 // [4H]
    float* part = sm + 4 * H;       // [4][UPW]
    const int u0 = blockIdx.x * UPW;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (dgnext) {
        for (int k = threadIdx.x; k < 4 * H; k += 256) sg[k] = dgnext[k];
        __syncthreads();
        // UPW rows of W_hh^T, each 4H long, split over the 4 waves: wave w takes unit (w % UPW), slice (w / UPW) of the row
        constexpr int SL = 4 / UPW;                      // k-slices per row
        const int i = wave % UPW, sl = wave / UPW;
        const int u = min(u0 + i, H - 1);
        const int klen = 4 * H / SL;
        const float d = row_dot(WhhT + (long)u * 4 * H + sl * klen, sg + sl * klen, klen, lane);
        if (lane == 0) part[sl * UPW + i] = d;
    }
    __syncthreads();
    if (threadIdx.x < UPW && u0 + threadIdx.x < H) {
        const int i = threadIdx.x, u = u0 + i;
        float dh = dh_out[u];
        if (dgnext) for (int sl = 0; sl < 4 / UPW; ++sl) dh += part[sl * UPW + i];
        const float gi = act[u], gf = act[H + u], gg = act[2 * H + u], go = act[3 * H + u];
        const float tc = tanhf(c[u]);
        const float dcv = dh * go * (1.f - tc * tc) + dc[u];
        dg[u] = dcv * gg * gi * (1.f - gi);
        dg[H + u] = dcv * (cprev ? cprev[u] : 0.f) * gf * (1.f - gf);
        dg[2 * H + u] = dcv * gi * (1.f - gg * gg);
        dg[3 * H + u] = dh * tc * go * (1.f - go);
        dc[u] = dcv * gf;
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
// x[t, j] *= dropout(t, j)  (layer-0 gradient passes through the inter-layer dropout)
__global__ void sst_drop_mul_kernel(float* __restrict__ x, int T, int H, DropCfg dc) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < (long)T * H) x[i] *= drop_mult(dc, (unsigned)i, 0u, SITE_SST);
}

// weighted BCE of misc/utils.py:78-99:  labels *= masks; w = labels*w0 + (1-labels)*w1 (w0 = 1-w1, per anchor k);
// loss = K * mean_{t,k} w * -(y log p + (1-y) log(1-p)),  p = scores*masks, logs clamped at -100 like torch's BCELoss.
__global__ __launch_bounds__(256) void tap_bce_fwd_kernel(const float* __restrict__ scores, const float* __restrict__ masks,
                                                          const float* __restrict__ labels, const float* __restrict__ w1,
                                                          float* __restrict__ loss, int T, int K) {
    __shared__ float red[4];
    float s = 0.f;
    for (long i = threadIdx.x; i < (long)T * K; i += 256) {
        const int k = (int)(i % K);
        const float y = labels[i] * masks[i], p = scores[i] * masks[i];
        const float w = y * (1.f - w1[k]) + (1.f - y) * w1[k];
        s -= w * (y * fmaxf(logf(p), -100.f) + (1.f - y) * fmaxf(logf(1.f - p), -100.f));
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) loss[0] = (red[0] + red[1] + red[2] + red[3]) / (float)((long)T * K) * (float)K;
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

struct SstWs { float *GIN[2], *ACT[2], *HS[2], *CS[2], *H0D; long total; };
static SstWs carve(int T, int D, int H, int K, float* base) {
    SstWs w;
    long off = 0;
    auto take = [&](long n) { float* p = base ? base + off : nullptr; off += rup(n, 64); return p; };
    for (int l = 0; l < 2; ++l) { w.GIN[l] = take((long)T * 4 * H); w.ACT[l] = take((long)T * 4 * H); w.HS[l] = take((long)T * H); w.CS[l] = take((long)T * H); }
    w.H0D = take((long)T * H);
    w.total = off;
    return w;
}
struct SstWsB { float *DG[2], *DHO, *DC, *DZ, *WT[2]; long total; };
static SstWsB carve_b(int T, int D, int H, int K, float* base) {
    SstWsB w;
    long off = 0;
    auto take = [&](long n) { float* p = base ? base + off : nullptr; off += rup(n, 64); return p; };
    for (int l = 0; l < 2; ++l) { w.DG[l] = take((long)T * 4 * H); w.WT[l] = take((long)H * 4 * H); }
    w.DHO = take((long)T * H); w.DC = take(H); w.DZ = take((long)T * K);
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
    for (int l = 0; l < 2; ++l) {
        // input-side pre-activations of the whole layer: X_l . W_ih^T + b_ih + b_hh   (layer 1 reads the dropped layer-0 output)
        const float* xin = l == 0 ? a->x : w.H0D;
        const int din = l == 0 ? D : H;
        echr_gemm_desc d = desc_nt(xin, din, a->w_ih[l], din, w.GIN[l], 4 * H, T, 4 * H, din);
        d.bias = a->b_ih[l]; d.bias2 = a->b_hh[l]; d.split_k = -1;
        RC(gemm(d, st));
        float* hs = l == 0 ? w.HS[0] : a->tap_feats;
        for (int t = 0; t < T; ++t) {
            const float* hp = t ? hs + (long)(t - 1) * H : nullptr;
            const float* cp = t ? w.CS[l] + (long)(t - 1) * H : nullptr;
            hipLaunchKernelGGL(sst_step_fwd_kernel, dim3(nwg), dim3(256), (H + 4 * UPW) * sizeof(float), st, a->w_hh[l],
                               w.GIN[l] + (long)t * 4 * H, hp, cp, w.ACT[l] + (long)t * 4 * H, hs + (long)t * H, w.CS[l] + (long)t * H,
                               l == 0 ? w.H0D + (long)t * H : nullptr, H, t, dc);
        }
        RC(check_launch("sst_step_fwd"));
    }
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
    for (int l = 1; l >= 0; --l) {
        RC(transpose(a->w_hh[l], H, b.WT[l], 4 * H, 4 * H, H, 4 * H, st));
        RC(fill_zero(b.DC, H, st));
        const float* hs = l == 0 ? w.HS[0] : a->tap_feats;
        for (int t = T - 1; t >= 0; --t) {
            hipLaunchKernelGGL(sst_step_bwd_kernel, dim3(nwg), dim3(256), (4 * H + 4 * UPW) * sizeof(float), st, b.WT[l],
                               t + 1 < T ? b.DG[l] + (long)(t + 1) * 4 * H : nullptr, b.DHO + (long)t * H, w.ACT[l] + (long)t * 4 * H,
                               w.CS[l] + (long)t * H, t ? w.CS[l] + (long)(t - 1) * H : nullptr, b.DC, b.DG[l] + (long)t * 4 * H, H);
        }
        RC(check_launch("sst_step_bwd"));
        // parameter gradients of the layer (sums over the T rows); h(t-1) pairs with dG(t): rows 1..T-1
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
        if (l == 1) {   // gradient into layer 0's (dropped) output: dG1 . W_ih1, through the dropout mask
            d = desc_nn(b.DG[1], 4 * H, a->w_ih[1], H, b.DHO, H, T, H, 4 * H); d.split_k = -1;
            RC(gemm(d, st));
            const long n = (long)T * H;
            hipLaunchKernelGGL(sst_drop_mul_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, b.DHO, T, H, dc);
            RC(check_launch("sst_drop_mul"));
        }
    }
    return 0;
}

extern "C" int echr_tap_bce_fwd(const float* scores, const float* masks, const float* labels, const float* w1, float* loss, int32_t T,
                                int32_t K, void* stream) {
    ECHR_REQUIRE(scores && masks && labels && w1 && loss && T > 0 && K > 0, "tap_bce_fwd: bad arguments");
    hipLaunchKernelGGL(tap_bce_fwd_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, scores, masks, labels, w1, loss, T, K);
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
