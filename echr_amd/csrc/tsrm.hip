// TSRM event-relation encoder on gfx950 (reference: models/MA_attention_8_NEW.py:35-49, :51-79, :101-177;
// fST0, use_posit=1).  The dense parts (event_emb, pair_pos_fc1/fc2, query/key, per-head products) run on
// the fp32 MFMA GEMM of gemm.hip as plain or head-batched problems; this file adds the on-device
// position-embedding generator (float64 math like the reference's numpy), the gated softmax over
// events with dropout, and its backward.
//
// Re-association used (documented in DESIGN.md): the reference multiplies softmax weights with the
// un-projected 512-d event features per head and then applies a grouped 1x1 conv (:169-173); here the conv
// weight is applied to the features first (XW = X . Wout^T, one [N,Df]x[Df,Do] GEMM) and each head then
// mixes its own dgo-column slice of XW -- same math, 1/G of the flops, fp32 rounding differs at 1e-7.
#include <cmath>
#include <cstdlib>
#include "echr_common.h"
#include <atomic>
#include "echr_internal.h"

namespace echr {

DropCfg make_drop(const echr_dropout* d, float p);
enum { SITE_TSRM = 0 };

// pos[i,j,:] for the event pair (i,j): [sin(dc*f_k), cos(dc*f_k), sin(dl*f_k), cos(dl*f_k)], k < Df/4, where
// dc = max(|c_i - c_j| / l_i, 1e-3) (float64), dl = log(l_j / l_i) evaluated in float32 as the reference does
// (lengths are cast to float32, MA_attention_8_NEW.py:70), arguments scaled by 100 / 10000^(4k/Df).
// sin and cos of a float64 argument to float32 accuracy: quadrant reduction in float64 (two-term pi/2; the arguments reach 10^4 rad, so
// it must not happen in float32), then the classic degree-9 / degree-10 polynomials on |r| <= pi/4 in float32 (truncation < 3e-9)
__device__ __forceinline__ void sincos_f64arg(double x, float& s, float& c) {
    const double q = rint(x * 0.6366197723675814);                        // 2 / pi
    const float r = (float)fma(-q, 6.123233995736766e-17, fma(-q, 1.5707963267948966, x));
    const float r2 = r * r;
    const float sp = r + r * r2 * (-1.6666667e-1f + r2 * (8.3333333e-3f + r2 * (-1.9841270e-4f + r2 * 2.7557319e-6f)));
    const float cp = 1.f + r2 * (-0.5f + r2 * (4.1666667e-2f + r2 * (-1.3888889e-3f + r2 * (2.4801587e-5f + r2 * -2.7557319e-7f))));
    const int n = (int)(long long)q & 3;
    const float ss = (n & 1) ? cp : sp, cs = (n & 1) ? sp : cp;
    s = (n & 2) ? -ss : ss;
    c = ((n + 1) & 2) ? -cs : cs;
}

// One thread forms 16 consecutive frequencies of one pair: the frequency scale 100 / 10000^(4k/Df) by one pow and a float64 recurrence, the
// arguments in float64, sin / cos by sincos_f64arg -- the values the reference's float64 numpy sin / cos round to, to ~1e-7.
__global__ __launch_bounds__(256) void posemb_kernel(const int* __restrict__ ev_start, const int* __restrict__ ev_len, float* __restrict__ pos,
                                                     int N, int Df) {
    const int F4 = Df / 4, KC = (F4 + 15) / 16;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)N * N * KC) return;
    const int kc = (int)(idx % KC);
    const long ij = idx / KC;
    const int j = (int)(ij % N), i = (int)(ij / N);
    const double ci = 0.5 * ((double)ev_start[i] + (double)(ev_start[i] + ev_len[i]));
    const double cj = 0.5 * ((double)ev_start[j] + (double)(ev_start[j] + ev_len[j]));
    const float li = (float)ev_len[i], lj = (float)ev_len[j];
    double dc = fabs((ci - cj) / (double)li);
    dc = dc > 1e-3 ? dc : 1e-3;
    const float ratio = __fdiv_rn(lj, li);
    const double dl = (double)(float)log((double)ratio);
    const double step = pow(10000.0, -4.0 / (double)Df);                 // 1 / (ratio of consecutive frequency divisors)
    double inv_dim = 100.0 * pow(10000.0, -(4.0 / (double)Df) * (double)(16 * kc));
    float* o = pos + ij * Df;
    for (int kk = 0; kk < 16; ++kk) {
        const int k = 16 * kc + kk;
        if (k >= F4) break;
        float sc, cc, sl, cl;
        sincos_f64arg(dc * inv_dim, sc, cc);
        sincos_f64arg(dl * inv_dim, sl, cl);
        o[k] = sc;
        o[F4 + k] = cc;
        o[2 * F4 + k] = sl;
        o[3 * F4 + k] = cl;
        inv_dim *= step;
    }
}

// How the position gate and the scaled affinity combine ahead of the softmax (MA_attention_8_NEW.py:148-157): fST0 gate * aff (the recipe),
// fST1 gate + aff, fST2 log(clamp(gate, 1e-6)) + aff, fST3 the gate alone; 4 = use_posit = 0: the affinity alone (no position branch at all).
__device__ __forceinline__ float fst_combine(int mode, float gate, float aff) {
    switch (mode) {
        case 0: return gate * aff;
        case 1: return gate + aff;
        case 2: return logf(fmaxf(gate, 1e-6f)) + aff;
        case 3: return gate;
        default: return aff;
    }
}
// gradients of the combination w.r.t. (gate, aff) times ds
__device__ __forceinline__ void fst_grad(int mode, float gate, float aff, float ds, float& dgate, float& daff) {
    switch (mode) {
        case 0: dgate = ds * aff; daff = ds * gate; break;
        case 1: dgate = ds; daff = ds; break;
        case 2: dgate = gate >= 1e-6f ? ds / gate : 0.f; daff = ds; break;          // (torch.clamp(min): the gradient passes where x >= min)
        case 3: dgate = ds; daff = 0.f; break;
        default: dgate = 0.f; daff = ds; break;
    }
}

// one wave per (event n, head g): w[m] = softmax_m(gate[n,m,g] * aff[g,n,m]); wd = w * dropout
__global__ __launch_bounds__(64) void tsrm_softmax_fwd_kernel(const float* __restrict__ GATE, const float* __restrict__ AFF,
                                                              float* __restrict__ WSM, float* __restrict__ WD, int N, int G, DropCfg dc, int mode) {
    const int n = blockIdx.x, g = blockIdx.y, lane = threadIdx.x;
    const float* aff = AFF + ((long)g * N + n) * N;
    auto gt = [&](int j) { return mode == 4 ? 0.f : GATE[((long)n * N + j) * G + g]; };
    float m = -INFINITY;
    for (int j = lane; j < N; j += 64) m = fmaxf(m, fst_combine(mode, gt(j), aff[j]));
    m = wave_max(m);
    float s = 0.f;
    for (int j = lane; j < N; j += 64) s += expf(fst_combine(mode, gt(j), aff[j]) - m);
    s = wave_sum(s);
    const float inv = 1.f / s;
    for (int j = lane; j < N; j += 64) {
        const float w = expf(fst_combine(mode, gt(j), aff[j]) - m) * inv;
        const long o = ((long)g * N + n) * N + j;
        WSM[o] = w;
        WD[o] = w * drop_mult(dc, (unsigned)(((long)n * G + g) * N + j), 0u, SITE_TSRM);
    }
}

// The same for many events (evaluation batches): one workgroup per event n stages its gate rows [N, G] -- contiguous in memory, read
// coalesced -- in LDS (row pitch G + 1: conflict-free column reads), then each wave takes heads g = wave, wave + 4, ... and streams
// the affinity row of (g, n), contiguous over m.  The form above reads the gates 64 bytes apart, three times (0.29 ms at N = 1000).
__global__ __launch_bounds__(256) void tsrm_softmax_rows_kernel(const float* __restrict__ GATE, const float* __restrict__ AFF,
                                                               float* __restrict__ WSM, float* __restrict__ WD, int N, int G, DropCfg dc, int mode) {
    extern __shared__ float sg[];                      // [N][G + 1]
    const int n = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6, P = G + 1;
    const float* gp = GATE + (long)n * N * G;
    for (int i = tid; i < N * G; i += 256) sg[(i / G) * P + (i % G)] = mode == 4 ? 0.f : gp[i];
    __syncthreads();
    for (int g = w; g < G; g += 4) {
        const float* aff = AFF + ((long)g * N + n) * N;
        float m = -INFINITY;
        for (int j = lane; j < N; j += 64) m = fmaxf(m, fst_combine(mode, sg[j * P + g], aff[j]));
        m = wave_max(m);
        float s = 0.f;
        for (int j = lane; j < N; j += 64) s += expf(fst_combine(mode, sg[j * P + g], aff[j]) - m);
        s = wave_sum(s);
        const float inv = 1.f / s;
        for (int j = lane; j < N; j += 64) {
            const float wv = expf(fst_combine(mode, sg[j * P + g], aff[j]) - m) * inv;
            const long o = ((long)g * N + n) * N + j;
            WSM[o] = wv;
            WD[o] = wv * drop_mult(dc, (unsigned)(((long)n * G + g) * N + j), 0u, SITE_TSRM);
        }
    }
}

// ds = w * (dw - sum_m w dw), dw = dWD * dropout;  dGATE[n,m,g] = ds * aff, dAFF[g,n,m] = ds * gate
__global__ __launch_bounds__(64) void tsrm_softmax_bwd_kernel(const float* __restrict__ GATE, const float* __restrict__ AFF,
                                                              const float* __restrict__ WSM, const float* __restrict__ DWD,
                                                              float* __restrict__ DGATE, float* __restrict__ DAFF, int N, int G, DropCfg dc, int mode) {
    const int n = blockIdx.x, g = blockIdx.y, lane = threadIdx.x;
    const long base = ((long)g * N + n) * N;
    float s = 0.f;
    for (int j = lane; j < N; j += 64)
        s += WSM[base + j] * DWD[base + j] * drop_mult(dc, (unsigned)(((long)n * G + g) * N + j), 0u, SITE_TSRM);
    s = wave_sum(s);
    for (int j = lane; j < N; j += 64) {
        const float dw = DWD[base + j] * drop_mult(dc, (unsigned)(((long)n * G + g) * N + j), 0u, SITE_TSRM);
        const float ds = WSM[base + j] * (dw - s);
        const long go = ((long)n * N + j) * G + g;
        float dgt, daf;
        fst_grad(mode, mode == 4 ? 0.f : GATE[go], AFF[base + j], ds, dgt, daf);
        DGATE[go] = dgt;
        DAFF[base + j] = daf;
    }
}

// ------------------------------------------------------------------------------------------------------
// Few events (training: N <= 64 per video): the per-head attention of MA_attention_8_NEW.py:138-160 in ONE launch forward and TWO backward,
// one wave per (event row, head).  At N = 64, dg = 32 a head is 64 x 64 x 32 -- 0.26 MFLOP per product: the batched GEMM launches these
// replace (forward: Q K^T, softmax, WD . XW; backward: d WD, d XW, softmax', d Q, d K) were three / five dependent launches of ~1 us of
// arithmetic each on the iteration's critical chains, each one a 16-workgroup grid that exposes every memory latency.  Here lane = the other
// event j: its K_j / XW_j row sits in registers (requested up front, with everything else the wave will read: ONE latency round), the row's
// own operand is wave-uniform (scalar loads), the softmax runs across the lanes, and the products that contract over j go through a 64-float
// LDS row (lane = (feature, half of j)).  N * G = 1024 independent waves spread over the chip.
// ------------------------------------------------------------------------------------------------------
constexpr int HEAD_MAXN = 64, HEAD_DG = 32;
__device__ __forceinline__ void head_row32(const float* __restrict__ p, float (&r)[HEAD_DG], bool on) {
#pragma unroll
    for (int v = 0; v < HEAD_DG / 4; ++v) {
        const float4 x = on ? *reinterpret_cast<const float4*>(p + 4 * v) : make_float4(0.f, 0.f, 0.f, 0.f);
        r[4 * v] = x.x; r[4 * v + 1] = x.y; r[4 * v + 2] = x.z; r[4 * v + 3] = x.w;
    }
}
// column c = lane & 31 of rows half*32 .. half*32+31 (half = lane >> 5) of a [N, ld] matrix: 128-byte coalesced per half and row
__device__ __forceinline__ void head_col32(const float* __restrict__ p, long ld, int N, int lane, float (&r)[HEAD_DG]) {
    const int c = lane & 31, h = lane >> 5;
#pragma unroll
    for (int u = 0; u < HEAD_DG; ++u) {
        const int j = h * 32 + u;
        r[u] = j < N ? p[(long)j * ld + c] : 0.f;
    }
}
// sum_j row[j] * col[j][c] with row in LDS and the lane's 32 column entries in registers; complete in every lane
__device__ __forceinline__ float head_contract(const float* srow, const float (&col)[HEAD_DG], int lane) {
    const float4* s4 = reinterpret_cast<const float4*>(srow + (lane >> 5) * 32);
    float o = 0.f;
#pragma unroll
    for (int v = 0; v < HEAD_DG / 4; ++v) {
        const float4 x = s4[v];
        o = fmaf(x.x, col[4 * v], o); o = fmaf(x.y, col[4 * v + 1], o); o = fmaf(x.z, col[4 * v + 2], o); o = fmaf(x.w, col[4 * v + 3], o);
    }
    return o + __shfl_xor(o, 32);
}
__global__ __launch_bounds__(64) void tsrm_rowhead_fwd_kernel(const float* __restrict__ Q, const float* __restrict__ K, const float* __restrict__ XW,
                                                              const float* __restrict__ GATE, const float* __restrict__ b_out,
                                                              float* __restrict__ AFF, float* __restrict__ WSM, float* __restrict__ WD,
                                                              float* __restrict__ OUT, int N, int Df, int Do, int G, float scale, DropCfg dc, int mode) {
    __shared__ __attribute__((aligned(16))) float sw[HEAD_MAXN];
    constexpr int DG = HEAD_DG;
    const int r = blockIdx.x, g = blockIdx.y, j = threadIdx.x;
    const bool on = j < N;
    const float gate = (on && mode != 4) ? GATE[((long)r * N + j) * G + g] : 0.f;
    float kj[DG], xw[DG];
    head_row32(K + (long)j * Df + g * DG, kj, on);
    head_col32(XW + g * DG, Do, N, j, xw);
    const float* __restrict__ qr = Q + (long)r * Df + g * DG;          // wave-uniform
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < DG; ++k) acc = fmaf(qr[k], kj[k], acc);
    const float aff = scale * acc;
    // gated softmax over j (the arithmetic of tsrm_softmax_fwd_kernel)
    const float v = on ? fst_combine(mode, gate, aff) : -INFINITY;
    const float m = wave_max(v);
    const float e = on ? expf(v - m) : 0.f;
    const float inv = 1.f / wave_sum(e);
    const float w = e * inv;
    const float wd = on ? w * drop_mult(dc, (unsigned)(((long)r * G + g) * N + j), 0u, SITE_TSRM) : 0.f;
    if (on) {
        const long o = ((long)g * N + r) * N + j;
        AFF[o] = aff;
        WSM[o] = w;
        WD[o] = wd;
    }
    sw[j] = wd;
    __syncthreads();
    // OUT[r, g*DG + c] = sum_j WD[r][j] XW[j][c] + b_out
    const float o = head_contract(sw, xw, j);
    if (j < DG) OUT[(long)r * Do + g * DG + j] = o + (b_out ? b_out[g * DG + j] : 0.f);
}
// backward, row side: d OUT_r -> d WD[r][:] -> softmax' -> d GATE[r, :, g], d AFF[g][r][:], d Q[r]
__global__ __launch_bounds__(64) void tsrm_rowhead_bwd_kernel(const float* __restrict__ DOUT, const float* __restrict__ K, const float* __restrict__ XW,
                                                              const float* __restrict__ GATE, const float* __restrict__ AFF, const float* __restrict__ WSM,
                                                              float* __restrict__ DGATE, float* __restrict__ DAFF, float* __restrict__ DQ,
                                                              int N, int Df, int Do, int G, float scale, DropCfg dc, int mode) {
    __shared__ __attribute__((aligned(16))) float sw[HEAD_MAXN];
    constexpr int DG = HEAD_DG;
    const int r = blockIdx.x, g = blockIdx.y, j = threadIdx.x;
    const bool on = j < N;
    const long base = ((long)g * N + r) * N;
    const long go = ((long)r * N + j) * G + g;
    const float gate = (on && mode != 4) ? GATE[go] : 0.f;
    const float wsm = on ? WSM[base + j] : 0.f;
    const float aff = on ? AFF[base + j] : 0.f;
    float xj[DG], kc[DG];
    head_row32(XW + (long)j * Do + g * DG, xj, on);
    head_col32(K + g * DG, Df, N, j, kc);
    const float* __restrict__ dr = DOUT + (long)r * Do + g * DG;       // wave-uniform
    float dwd = 0.f;
#pragma unroll
    for (int c = 0; c < DG; ++c) dwd = fmaf(dr[c], xj[c], dwd);
    // the arithmetic of tsrm_softmax_bwd_kernel
    const float dw = on ? dwd * drop_mult(dc, (unsigned)(((long)r * G + g) * N + j), 0u, SITE_TSRM) : 0.f;
    const float s = wave_sum(wsm * dw);
    const float ds = wsm * (dw - s);
    float dgt = 0.f, daff = 0.f;
    if (on) fst_grad(mode, gate, aff, ds, dgt, daff);
    if (on) {
        DGATE[go] = dgt;
        DAFF[base + j] = daff;
    }
    sw[j] = daff;
    __syncthreads();
    const float o = head_contract(sw, kc, j);
    if (j < DG) DQ[(long)r * Df + g * DG + j] = scale * o;
}
// backward, column side: d XW[j] = sum_i WD[i][j] d OUT[i],  d K[j] = scale * sum_i d AFF[i][j] Q[i]   (one wave per (j, head); lane = i for
// the two column gathers, lane = (feature, half of i) for the contraction)
__global__ __launch_bounds__(64) void tsrm_colhead_bwd_kernel(const float* __restrict__ DOUT, const float* __restrict__ Q, const float* __restrict__ WD,
                                                              const float* __restrict__ DAFF, float* __restrict__ DXW, float* __restrict__ DK,
                                                              int N, int Df, int Do, int G, float scale) {
    __shared__ __attribute__((aligned(16))) float sw[2][HEAD_MAXN];
    constexpr int DG = HEAD_DG;
    const int jc = blockIdx.x, g = blockIdx.y, i = threadIdx.x;
    const long o = ((long)g * N + i) * N + jc;
    const float wd = i < N ? WD[o] : 0.f;
    const float da = i < N ? DAFF[o] : 0.f;
    float dc[DG], qc[DG];
    head_col32(DOUT + g * DG, Do, N, i, dc);
    head_col32(Q + g * DG, Df, N, i, qc);
    sw[0][i] = wd;
    sw[1][i] = da;
    __syncthreads();
    const float dx = head_contract(sw[0], dc, i);
    const float dk = head_contract(sw[1], qc, i);
    if (i < DG) {
        DXW[(long)jc * Do + g * DG + i] = dx;
        DK[(long)jc * Df + g * DG + i] = scale * dk;
    }
}
// the fused per-head kernels serve N <= 64 events with heads of 32 features: the reference's 512 / 16 (MA_attention_8_NEW.py:14-22)
// (ECHR_TSRM_HEADS=0: the batched-GEMM form, for A/B runs)
static bool head_fused_ok(int N, int Df, int Do, int G) {
    static const bool off = [] { const char* e = getenv("ECHR_TSRM_HEADS"); return e && e[0] == '0'; }();
    return !off && N <= HEAD_MAXN && Df == G * HEAD_DG && Do == G * HEAD_DG;
}

static inline long rup(long x, long a) { return (x + a - 1) / a * a; }

struct TsrmWs { float *X, *POS, *P1, *GATE, *Q, *K, *XW, *AFF, *WSM, *WD, *PK_POS, *PK_WFC1; long total, zero_floats; };
static TsrmWs carve(int N, int Din, int Df, int Do, int G, float* base) {
    TsrmWs w;
    long off = 0;
    auto take = [&](long n) { float* p = base ? base + off : nullptr; off += rup(n, 64); return p; };
    const long NN = (long)N * N;
    // X | GATE | Q | K | XW are split-K (atomically accumulated) GEMM outputs: contiguous, zeroed by one fill
    w.X = take((long)N * Df); w.GATE = take(NN * G); w.Q = take((long)N * Df); w.K = take((long)N * Df); w.XW = take((long)N * Do);
    w.zero_floats = off;
    w.POS = take(NN * Df); w.P1 = take(NN * Df);
    w.AFF = take(NN * G); w.WSM = take(NN * G); w.WD = take(NN * G);
    w.PK_POS = take(h2_floats((int)NN, Df)); w.PK_WFC1 = take(h2_floats(Df, Df));      // h2-packed operands of the fc1 product
    w.total = off;
    return w;
}
struct TsrmWsB { float *DWD, *DGATE, *DAFF, *DQ, *DK, *DXW, *DX, *DP1, *PK_DP1T, *PK_POST, *PMP; long total; };
constexpr int PM_ROWS = 32, PM_LD = 16 * 512 + 512 + 64;          // pair_mlp_bwd_kernel: pair rows per block; floats per block of its partial sums (d W_fc2 | d b_fc1 | d b_fc2)
static TsrmWsB carve_b(int N, int Din, int Df, int Do, int G, float* base) {
    TsrmWsB w;
    long off = 0;
    auto take = [&](long n) { float* p = base ? base + off : nullptr; off += rup(n, 64); return p; };
    const long NN = (long)N * N;
    w.DWD = take(NN * G); w.DGATE = take(NN * G); w.DAFF = take(NN * G);
    w.DQ = take((long)N * Df); w.DK = take((long)N * Df); w.DXW = take((long)N * Do); w.DX = take((long)N * Df);
    w.DP1 = take(NN * Df);
    w.PK_DP1T = take(h2_floats(Df, (int)NN)); w.PK_POST = take(h2_floats(Df, (int)NN));  // transposed packs for the fc1 weight gradient
    w.PMP = take(((NN + PM_ROWS - 1) / PM_ROWS) * (long)PM_LD);                          // per-block partial sums of pair_mlp_bwd_kernel
    w.total = off;
    return w;
}

// ------------------------------------------------------------------------------------------------------
// Position-MLP backward in ONE pass over the fc1 activations of the N^2 pairs (MA_attention_8_NEW.py:108-116 backward):
//     d P1   = (d GATE . W_fc2) * (1 - P1^2)      [NN, 512]   (gradient at the input of tanh: feeds d W_fc1)
//     d W_fc2 += d GATE^T . P1                    [16, 512]
//     d b_fc1 += column sums of d P1,  d b_fc2 += column sums of d GATE
// A thread owns two adjacent fc1 columns: its 16 x 2 slice of W_fc2 and its 16 x 2 accumulators of d W_fc2 live in registers; a block walks
// PM_ROWS pair rows (d GATE rows through LDS, broadcast reads).  Replaces two fp32 GEMM launches whose shapes (M = 16, resp. K = 16) left them
// latency-bound (68 + 17 us beside the backward tail's other kernels) and two of the six column-sum jobs.
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 4) void pair_mlp_bwd_kernel(const float* __restrict__ DGATE, const float* __restrict__ P1, const float* __restrict__ W2,
                                                           float* __restrict__ DP1, float* __restrict__ PART, int NN) {
    constexpr int G = 16, Df = 512;
    __shared__ float dg[PM_ROWS][G];
    const int tid = threadIdx.x, c0 = 2 * tid;
    const long r0 = (long)blockIdx.x * PM_ROWS;
    const int nr = (int)min((long)PM_ROWS, (long)NN - r0);
    for (int i = tid; i < PM_ROWS * G; i += 256) dg[i / G][i % G] = (i / G) < nr ? DGATE[(r0 + i / G) * G + (i % G)] : 0.f;
    float2 w2[G], gw[G];
#pragma unroll
    for (int g = 0; g < G; ++g) { w2[g] = *reinterpret_cast<const float2*>(W2 + (long)g * Df + c0); gw[g] = make_float2(0.f, 0.f); }
    float2 pv[PM_ROWS / 4];
    float2 bsum = make_float2(0.f, 0.f);
    __syncthreads();
#pragma unroll 1          // (fully unrolled the compiler hoisted every load of the block and spilled 850 registers: 270 us instead of 10)
    for (int rb = 0; rb < PM_ROWS; rb += PM_ROWS / 4) {
#pragma unroll
        for (int i = 0; i < PM_ROWS / 4; ++i) pv[i] = (rb + i) < nr ? *reinterpret_cast<const float2*>(P1 + (r0 + rb + i) * Df + c0) : make_float2(0.f, 0.f);
#pragma unroll 1
        for (int i = 0; i < PM_ROWS / 4; ++i) {
            const int r = rb + i;
            float2 sacc = make_float2(0.f, 0.f);
#pragma unroll
            for (int g4 = 0; g4 < G; g4 += 4) {
                const float4 d4 = *reinterpret_cast<const float4*>(&dg[r][g4]);
                const float dv[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    sacc.x = fmaf(dv[j], w2[g4 + j].x, sacc.x); sacc.y = fmaf(dv[j], w2[g4 + j].y, sacc.y);
                    gw[g4 + j].x = fmaf(dv[j], pv[i].x, gw[g4 + j].x); gw[g4 + j].y = fmaf(dv[j], pv[i].y, gw[g4 + j].y);
                }
            }
            const float2 dp = make_float2(sacc.x * (1.f - pv[i].x * pv[i].x), sacc.y * (1.f - pv[i].y * pv[i].y));
            if (r < nr) *reinterpret_cast<float2*>(DP1 + (r0 + r) * Df + c0) = dp;
            bsum.x += dp.x; bsum.y += dp.y;
        }
    }
    // per-block partial sums, plain coalesced stores: [16 x 512] d W_fc2 | [512] d b_fc1 | [16] d b_fc2; a multi-problem column-sum launch folds
    // the blocks (128 blocks adding atomically into the same 8.7 K addresses ran at the contended-atomic rate: 100 us)
    float* part = PART + (long)blockIdx.x * PM_LD;
#pragma unroll
    for (int g = 0; g < G; ++g) *reinterpret_cast<float2*>(part + g * Df + c0) = gw[g];
    *reinterpret_cast<float2*>(part + G * Df + c0) = bsum;
    if (tid < G) {
        float s = 0.f;
        for (int r = 0; r < nr; ++r) s += dg[r][tid];
        part[G * Df + Df + tid] = s;
    }
}

// ------------------------------------------------------------------------------------------------------
// Inference over many pairs: fc1 is linear in the embedding, and the embedding of pair (i, j) is [phi(dc_ij) | phi(dl_ij)] with
// dc_ij = max(|c_i - c_j| / l_i, 1e-3) and dl_ij = log(l_j / l_i) -- functions of the INTEGER triples (|2 c_i - 2 c_j|, l_i) and (l_i, l_j).
// Segment indices are small integers (proposal lengths <= a few hundred, videos a few hundred segments), so 10^6 pairs share a few 10^4
// distinct arguments: F_c[key] = W1[:, :Df/2] . phi(dc) + b1 and F_l[key] = W1[:, Df/2:] . phi(dl) are tabulated once (two small GEMMs over
// the distinct keys), and the per-pair work is tanh(F_c[kc] + F_l[kl]) . W2^T -- two 2-KB row gathers instead of 512 sin / cos and a
// 512 x 512 product per pair.  Same arithmetic as the dense path up to the order of one addition.
// ------------------------------------------------------------------------------------------------------
// phi over the keys of one table: mode 0: key = dcn * Lm1 + l_i (dcn = |2 c_i - 2 c_j|), mode 1: key = l_i * Lm1 + l_j.  out [R, Df / 2] = [sin | cos].
__global__ __launch_bounds__(256) void pair_phi_kernel(int mode, int Lm1, long R, int Df, float* __restrict__ out) {
    const int F4 = Df / 4;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= R * F4) return;
    const int k = (int)(idx % F4);
    const long key = idx / F4;
    const int hi = (int)(key / Lm1), lo = (int)(key % Lm1);
    double x;
    bool ok;
    if (mode == 0) {
        ok = lo >= 1;                                         // l_i
        const double dc = fabs((0.5 * (double)hi) / (double)(float)(ok ? lo : 1));
        x = dc > 1e-3 ? dc : 1e-3;
    } else {
        ok = hi >= 1 && lo >= 1;                              // l_i, l_j
        x = (double)(float)log((double)__fdiv_rn((float)(ok ? lo : 1), (float)(ok ? hi : 1)));
    }
    float sv = 0.f, cv = 0.f;
    if (ok) sincos_f64arg(x * (100.0 * pow(10000.0, -(4.0 / (double)Df) * (double)k)), sv, cv);
    out[key * (Df / 2) + k] = sv;
    out[key * (Df / 2) + F4 + k] = cv;
}

// gate[r, :] = W2 . tanh(F_c[kc(r)] + F_l[kl(r)]) + b2 for every pair r = (i, j): skinny_nt_kernel's structure (one wave per 16 pairs,
// v_mfma_f32_16x16x4_f32, W2 in registers), its A fragments formed on the fly from the two gathered table rows.
template <int KS>          // Df / 16
__global__ __launch_bounds__(256) void pair_gate_kernel(const float* __restrict__ FC, const float* __restrict__ FL, const int* __restrict__ ev_start,
                                                        const int* __restrict__ ev_len, int N, int Lm1, int span, const float* __restrict__ W2, long ldw,
                                                        const float* __restrict__ bias, float* __restrict__ C, long ldc, int Nc, int tiles) {
    typedef float f32x4s __attribute__((ext_vector_type(4)));
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, r = lane & 15, kk = lane >> 4;
    const long M = (long)N * N;
    float4 bw[KS];
#pragma unroll
    for (int s_ = 0; s_ < KS; ++s_)
        bw[s_] = r < Nc ? *reinterpret_cast<const float4*>(W2 + (long)r * ldw + 16 * s_ + 4 * kk) : make_float4(0.f, 0.f, 0.f, 0.f);
    const float bv = (bias && r < Nc) ? bias[r] : 0.f;
    for (int tile = blockIdx.x * 4 + w; tile < tiles; tile += gridDim.x * 4) {
        const long r0 = (long)tile * 16;
        const long row = r0 + r < M ? r0 + r : M - 1;
        const int i = (int)(row / N), j = (int)(row % N);
        // (clamped to the table bounds: bounds that do not cover the events give wrong gates, never a wild read)
        const int li = min(max(ev_len[i], 0), Lm1 - 1), lj = min(max(ev_len[j], 0), Lm1 - 1);
        const int c2i = 2 * ev_start[i] + ev_len[i], c2j = 2 * ev_start[j] + ev_len[j];
        const float* fc = FC + ((long)min(abs(c2i - c2j), span) * Lm1 + li) * (16 * KS) + 4 * kk;
        const float* fl = FL + ((long)li * Lm1 + lj) * (16 * KS) + 4 * kk;
        f32x4s acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int h = 0; h < 4; ++h) {          // four quarters of the row: 2 x KS / 4 gathered float4 in flight
            float4 x[KS / 4], y[KS / 4];
#pragma unroll
            for (int s_ = 0; s_ < KS / 4; ++s_) {
                x[s_] = *reinterpret_cast<const float4*>(fc + 16 * (h * (KS / 4) + s_));
                y[s_] = *reinterpret_cast<const float4*>(fl + 16 * (h * (KS / 4) + s_));
            }
#pragma unroll
            for (int s_ = 0; s_ < KS / 4; ++s_) {
                const float4 b4 = bw[h * (KS / 4) + s_];
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(tanhf(x[s_].x + y[s_].x), b4.x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(tanhf(x[s_].y + y[s_].y), b4.y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(tanhf(x[s_].z + y[s_].z), b4.z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(tanhf(x[s_].w + y[s_].w), b4.w, acc, 0, 0, 0);
            }
        }
        if (r < Nc) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const long orow = r0 + 4 * kk + q;
                if (orow < M) C[orow * ldc + r] = acc[q] + bv;
            }
        }
    }
}

// rows of the two tables for these bounds, or 0 when the table path does not apply (unknown bounds, too few pairs, tables not much
// smaller than the pair list, shapes the gate kernel has no instantiation for)
static long pair_table_rows(const echr_tsrm_args* a, long* rc, long* rl) {
    const long NN = (long)a->N * a->N;
    if (!a->inference || !config().pair_tables || a->max_len <= 0 || a->max_span < 0 || NN < 16384 || a->G > 16 || (a->Df != 512 && a->Df != 256) ||
        !config().gemm_h2)
        return 0;
    const long Lm1 = (long)a->max_len + 1;
    *rc = ((long)a->max_span + 1) * Lm1;
    *rl = Lm1 * Lm1;
    return (*rc + *rl) * 2 <= NN ? *rc + *rl : 0;
}

static int check(const echr_tsrm_args* a, const char* who) {
    ECHR_REQUIRE(a, "%s: null args", who);
    ECHR_REQUIRE(a->N > 0 && a->Din > 0 && a->Df > 0 && a->Do > 0 && a->G > 0, "%s: bad dims", who);
    ECHR_REQUIRE(a->Df % a->G == 0 && a->Do % a->G == 0 && a->Df % 4 == 0, "%s: need Df%%G==0, Do%%G==0, Df%%4==0", who);
    ECHR_REQUIRE(a->ws && a->out && a->ech && a->ev_start && a->ev_len, "%s: missing buffers", who);
    return 0;
}

#define RC(x) do { int _rc = (x); if (_rc) return _rc; } while (0)

// The same embedding with one thread per frequency: the 4 x Df/4 outputs of a pair leave as four contiguous runs (the form above writes
// 4-byte pieces 64 bytes apart: 3.9 ms for the 2 GB of N = 1000).  A block handles PPB pairs; their two coordinates are formed once, in
// shared memory, by the block's first PPB threads; every thread keeps its frequency's scale 100 / 10000^(4k/Df).
constexpr int POSEMB_PPB = 32;
__global__ __launch_bounds__(256) void posemb_rows_kernel(const int* __restrict__ ev_start, const int* __restrict__ ev_len, float* __restrict__ pos,
                                                          int N, int Df) {
    __shared__ double sdc[POSEMB_PPB], sdl[POSEMB_PPB];
    const int F4 = Df / 4, slots = 256 / F4;          // F4 divides 256 (checked by the caller)
    const int k = threadIdx.x % F4, sub = threadIdx.x / F4;
    const long p0 = (long)blockIdx.x * POSEMB_PPB, NN = (long)N * N;
    if (threadIdx.x < POSEMB_PPB && p0 + threadIdx.x < NN) {
        const long ij = p0 + threadIdx.x;
        const int j = (int)(ij % N), i = (int)(ij / N);
        const double ci = 0.5 * ((double)ev_start[i] + (double)(ev_start[i] + ev_len[i]));
        const double cj = 0.5 * ((double)ev_start[j] + (double)(ev_start[j] + ev_len[j]));
        const float li = (float)ev_len[i], lj = (float)ev_len[j];
        double dc = fabs((ci - cj) / (double)li);
        sdc[threadIdx.x] = dc > 1e-3 ? dc : 1e-3;
        sdl[threadIdx.x] = (double)(float)log((double)__fdiv_rn(lj, li));
    }
    const double inv_dim = 100.0 * pow(10000.0, -(4.0 / (double)Df) * (double)k);
    __syncthreads();
    for (int q = sub; q < POSEMB_PPB; q += slots) {
        const long ij = p0 + q;
        if (ij >= NN) break;
        float sc, cc, sl, cl;
        sincos_f64arg(sdc[q] * inv_dim, sc, cc);
        sincos_f64arg(sdl[q] * inv_dim, sl, cl);
        float* o = pos + ij * Df + k;
        o[0] = sc; o[F4] = cc; o[2 * F4] = sl; o[3 * F4] = cl;
    }
}

// The embedding emitted directly as the h2-packed A operand of the fc1 product (csrc/gemm.hip: chunks [row block 128][k block 32] of two
// fp16 planes with the slot swizzle baked in, then the inverse scales), and -- when a backward pass may follow -- also as fp32.  |sin|,
// |cos| <= 1, so one fixed block scale 2^14 serves every row (the format allows any power of two per row and 256-k segment).  A thread owns
// (pair, 8 consecutive frequencies): four 16-byte slots per plane.  Saves the packing pass over the 2 GB (N = 1000) tensor and, in inference,
// the fp32 tensor itself.
constexpr int PK_ROUNDS = 4;          // a block handles PK_ROUNDS x (256 / (Df / 32)) pairs: the frequency table (one float64 pow each) is built once per block
__global__ __launch_bounds__(256) void posemb_packed_kernel(const int* __restrict__ ev_start, const int* __restrict__ ev_len, float* __restrict__ pos,
                                                            unsigned char* __restrict__ pk, int N, int Df) {
    __shared__ double sdc[16 * PK_ROUNDS], sdl[16 * PK_ROUNDS], sfr[256];
    extern __shared__ __attribute__((aligned(16))) unsigned char pk_lds[];
    uint4* stage = reinterpret_cast<uint4*>(pk_lds);          // [2 planes][KT][ppr rows][4 slots] x 16 B: one round's packed rows
    const int F4 = Df / 4, FG = F4 / 8;               // frequency groups per pair (16 at Df = 512)
    const int ppr = 256 / FG, ppb = ppr * PK_ROUNDS;  // pairs per round / per block
    const int fg = threadIdx.x % FG, pr = threadIdx.x / FG;
    const long NN = (long)N * N, p0 = (long)blockIdx.x * ppb;
    if (threadIdx.x < ppb && p0 + threadIdx.x < NN) {
        const long ij = p0 + threadIdx.x;
        const int j = (int)(ij % N), i = (int)(ij / N);
        const double ci = 0.5 * ((double)ev_start[i] + (double)(ev_start[i] + ev_len[i]));
        const double cj = 0.5 * ((double)ev_start[j] + (double)(ev_start[j] + ev_len[j]));
        const float li = (float)ev_len[i], lj = (float)ev_len[j];
        double dc = fabs((ci - cj) / (double)li);
        sdc[threadIdx.x] = dc > 1e-3 ? dc : 1e-3;
        sdl[threadIdx.x] = (double)(float)log((double)__fdiv_rn(lj, li));
    }
    if (threadIdx.x < F4) sfr[threadIdx.x] = 100.0 * pow(10000.0, -(4.0 / (double)Df) * (double)threadIdx.x);
    __syncthreads();
    double fr[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) fr[j] = sfr[8 * fg + j];
    const int KT = Df / 32;
    const long RB = (NN + 127) / 128;
    float* inv = reinterpret_cast<float*>(pk + RB * KT * 16384);
    for (int rd = 0; rd < PK_ROUNDS; ++rd) {
        const long row0 = p0 + (long)rd * ppr, row = row0 + pr;          // the round's ppr pairs: consecutive rows of ONE 128-row block (ppr divides 128)
        if (row0 >= NN) return;                                           // uniform over the block
        const bool live = row < NN;
        float v[4][8];
        const double dc = sdc[rd * ppr + pr], dl = sdl[rd * ppr + pr];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            sincos_f64arg(dc * fr[j], v[0][j], v[1][j]);
            sincos_f64arg(dl * fr[j], v[2][j], v[3][j]);
        }
        if (pos && live) {
            float* o = pos + row * Df + 8 * fg;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                *reinterpret_cast<float4*>(o + q * F4) = make_float4(v[q][0], v[q][1], v[q][2], v[q][3]);
                *reinterpret_cast<float4*>(o + q * F4 + 4) = make_float4(v[q][4], v[q][5], v[q][6], v[q][7]);
            }
        }
        const long rb = row0 >> 7;
        const int r = (int)(row & 127);
        if (rd > 0) __syncthreads();                                      // the previous round's pieces have left the staging buffer
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int k0 = q * F4 + 8 * fg, kt = k0 >> 5, sl = ((k0 & 31) >> 3) ^ ((0x78 >> (2 * ((r >> 2) & 3))) & 3);
            unsigned hw[8], lw[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float xs = v[q][j] * 16384.f;
                const _Float16 h1 = (_Float16)xs;
                const _Float16 h2 = (_Float16)(xs - (float)h1);
                hw[j] = (unsigned)__builtin_bit_cast(unsigned short, h1);
                lw[j] = (unsigned)__builtin_bit_cast(unsigned short, h2);
            }
            uint4* dst = stage + (kt * ppr + pr) * 4 + sl;
            dst[0] = make_uint4(hw[0] | (hw[1] << 16), hw[2] | (hw[3] << 16), hw[4] | (hw[5] << 16), hw[6] | (hw[7] << 16));
            dst[KT * ppr * 4] = make_uint4(lw[0] | (lw[1] << 16), lw[2] | (lw[3] << 16), lw[4] | (lw[5] << 16), lw[6] | (lw[7] << 16));
            if (live && (k0 & 31) == 0) inv[(rb * KT + kt) * 128 + r] = 6.103515625e-05f;          // 2^-14
        }
        __syncthreads();
        // 2 planes x KT k blocks x (ppr x 64 B): piece = 16 B; a wave's 64 pieces are one contiguous ppr x 64-byte run (ppr = 16: 1 KB)
        const int pieces = 2 * KT * ppr * 4, ppc = ppr * 4;
        for (int pc = threadIdx.x; pc < pieces; pc += 256) {
            const int pl = pc / (KT * ppc), kt = (pc / ppc) % KT, within = pc % ppc;
            if (row0 + within / 4 < NN)
                *reinterpret_cast<uint4*>(pk + (rb * KT + kt) * 16384 + pl * 8192 + (int)(row0 & 127) * 64 + within * 16) = stage[pc];
        }
    }
}

bool posemb_packed_ok(int N, int Df) {
    const int F4 = Df / 4;
    return config().posemb_packed && Df % 32 == 0 && F4 % 8 == 0 && F4 <= 256 && F4 / 8 <= 256 && 256 % (F4 / 8) == 0 && 256 / (F4 / 8) <= 16 && 128 % (256 / (F4 / 8)) == 0 && (F4 % 32) == 0;
}
int posemb_packed(const int* ev_start, const int* ev_len, float* pos, float* pk, int N, int Df, hipStream_t st) {
    const long NN = (long)N * N;
    const int ppb = PK_ROUNDS * (256 / (Df / 32));
    hipLaunchKernelGGL(posemb_packed_kernel, dim3((unsigned)((NN + ppb - 1) / ppb)), dim3(256), 2 * (Df / 32) * (256 / (Df / 32)) * 64, st, ev_start, ev_len, pos,
                       reinterpret_cast<unsigned char*>(pk), N, Df);
    return check_launch("posemb_packed");
}

int posemb(const int* ev_start, const int* ev_len, float* pos, int N, int Df, hipStream_t st) {
    if (Df / 4 <= 256 && 256 % (Df / 4) == 0 && config().posemb_rows) {
        const long NN = (long)N * N;
        hipLaunchKernelGGL(posemb_rows_kernel, dim3((unsigned)((NN + POSEMB_PPB - 1) / POSEMB_PPB)), dim3(256), 0, st, ev_start, ev_len, pos, N, Df);
        return check_launch("posemb_rows");
    }
    const long tot = (long)N * N * ((Df / 4 + 15) / 16);
    hipLaunchKernelGGL(posemb_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, ev_start, ev_len, pos, N, Df);
    return check_launch("posemb");
}

}  // namespace echr

using namespace echr;

extern "C" int64_t echr_tsrm_ws_floats(int32_t N, int32_t Din, int32_t Df, int32_t Do, int32_t G) {
    return carve(N, Din, Df, Do, G, nullptr).total;
}
extern "C" int64_t echr_tsrm_ws_bwd_floats(int32_t N, int32_t Din, int32_t Df, int32_t Do, int32_t G) {
    return carve_b(N, Din, Df, Do, G, nullptr).total;
}

extern "C" int echr_tsrm_posemb(const int32_t* ev_start, const int32_t* ev_len, float* pos, int32_t N, int32_t Df, void* stream) {
    ECHR_REQUIRE(ev_start && ev_len && pos && N > 0 && Df > 0 && Df % 4 == 0, "tsrm_posemb: bad arguments");
    return posemb(ev_start, ev_len, pos, N, Df, (hipStream_t)stream);
}

static int tsrm_fwd_impl(const echr_tsrm_args* a, const echr_dropout* drop, void* stream, const float* x_given, const float* pos_given);

extern "C" int echr_tsrm_fwd(const echr_tsrm_args* a, const echr_dropout* drop, void* stream) {
    RC(check(a, "tsrm_fwd"));
    return tsrm_fwd_impl(a, drop, stream, nullptr, nullptr);
}

// attention_module_multi_head.forward on its own (MA_attention_8_NEW.py:101-177): the caller hands in the already embedded events
// roi_feat [N,Df] and the pairwise position embedding [N,N,Df]; inference-style entry (fST0, use_posit = 1)
extern "C" int echr_tsrm_attn_fwd(const echr_tsrm_args* a, const float* roi_feat, const float* pos_emb, const echr_dropout* drop, void* stream) {
    ECHR_REQUIRE(a && roi_feat && pos_emb, "tsrm_attn_fwd: null arguments");
    ECHR_REQUIRE(a->N > 0 && a->Df > 0 && a->Do > 0 && a->G > 0 && a->Df % a->G == 0 && a->Do % a->G == 0 && a->Df % 4 == 0, "tsrm_attn_fwd: bad dims");
    ECHR_REQUIRE(a->ws && a->out, "tsrm_attn_fwd: missing buffers");
    return tsrm_fwd_impl(a, drop, stream, roi_feat, pos_emb);
}

// the dense position branch (:39-41, :108-116): pair embedding -> fc1 (+ tanh) -> fc2 gates, on stream sp.  Independent of the event features.
// wfc1_ready: the packed image of W_fc1 is produced on another stream (tsrm_position_early: the caller's, which has slack there), this one waits
// for that event in front of the product instead of running the pack between the embedding and the product
static int position_branch(const echr_tsrm_args* a, const TsrmWs& w, bool do_posemb, bool packed_pos, hipStream_t sp, hipEvent_t wfc1_ready = nullptr) {
    const int N = a->N, Df = a->Df, G = a->G, NN = N * N;
    echr_gemm_desc d;
    if (do_posemb) {
        // many pairs: the embedding leaves its kernel already packed for the fc1 product (and as fp32 only if a backward pass may follow)
        if (packed_pos) RC(posemb_packed(a->ev_start, a->ev_len, a->inference ? nullptr : w.POS, w.PK_POS, N, Df, sp));
        else RC(posemb(a->ev_start, a->ev_len, w.POS, N, Df, sp));
    }
    // fc1 over the N*N event pairs: the one TSRM product big enough for the h2 path (tanh fused in the epilogue)
    if (config().gemm_h2 && NN >= 1024) {
        H2PackJob pj[2] = {pack_rows(a->w_fc1, Df, Df, Df, w.PK_WFC1), pack_rows(w.POS, Df, NN, Df, w.PK_POS)};
        if (!wfc1_ready) RC(h2_pack_multi(pj, packed_pos ? 1 : 2, sp));
        else {
            if (!packed_pos) RC(h2_pack_multi(pj + 1, 1, sp));
            if (hipStreamWaitEvent(sp, wfc1_ready, 0) != hipSuccess) { set_error("tsrm_fwd: stream wait failed"); return -5; }
        }
        d = desc_h2(w.PK_POS, w.PK_WFC1, w.P1, Df, NN, Df, Df);
        d.split_k = 1;
    } else {
        d = desc_nt(w.POS, Df, a->w_fc1, Df, w.P1, Df, NN, Df, Df);
    }
    d.bias = a->b_fc1; d.act = ECHR_ACT_TANH;
    RC(gemm(d, sp));
    if (gemm_skinny_ok(NN, G, Df, Df, Df, w.P1, a->w_fc2)) {
        RC(gemm_skinny_nt(w.P1, Df, a->w_fc2, Df, a->b_fc2, w.GATE, G, NN, G, Df, sp));          // many pairs: a stream over the fc1 activations
    } else {
        d = desc_nt(w.P1, Df, a->w_fc2, Df, w.GATE, G, NN, G, Df); d.bias = a->b_fc2; d.split_k = -1; d.beta = 1.f;
        RC(gemm(d, sp));
    }
    return 0;
}

// echr_train_step: the position branch depends on the index vectors and the parameters only, so it is started on the helper stream as soon
// as the indices are staged -- ahead of event pooling and of the decoder's prepare chain -- instead of inside echr_tsrm_fwd (it is the longer
// of the event encoder's two chains: ~85 us against ~55 us; started ~40 us earlier, the per-head attention no longer waits for it).  Only the
// form that STORES its gates (skinny fc2 product): echr_tsrm_fwd's zero fill then leaves GATE alone.  The following echr_tsrm_fwd on the same
// workspace picks the branch up (and joins it); any other call forgets it.
static const void* g_pos_early_ws = nullptr;
int echr::tsrm_position_early(const echr_tsrm_args* a, hipStream_t from) {
    g_pos_early_ws = nullptr;
    static const bool off = [] { const char* e = getenv("ECHR_TSRM_EARLY"); return e && e[0] == '0'; }();      // A/B switch
    if (off || !config().tsrm_fork || !a || a->inference || !a->ws || !a->ev_start || !a->ev_len || a->fst_mode == 4) return 0;
    const int N = a->N, Df = a->Df, G = a->G, NN = N * N;
    if (N <= 0 || Df <= 0 || G <= 0 || Df % 4 != 0) return 0;
    TsrmWs w = carve(N, a->Din, Df, a->Do, G, a->ws);
    if (!gemm_skinny_ok(NN, G, Df, Df, Df, w.P1, a->w_fc2)) return 0;
    hipStream_t sp = aux_fork(from);
    if (!sp) return 0;
    const bool packed_pos = config().gemm_h2 && NN >= 4096 && posemb_packed_ok(N, Df);
    // the W_fc1 pack (parameters only) leaves the branch's chain: it runs on the caller's stream, which idles ~40 us waiting for the gates anyway
    static const bool wfc1_here = [] { const char* e = getenv("ECHR_WFC1_ON_CALLER"); return !(e && e[0] == '0'); }();      // A/B switch
    static hipEvent_t ev_wfc1 = nullptr;
    hipEvent_t ready = nullptr;
    if (wfc1_here && config().gemm_h2 && NN >= 1024) {
        if (!ev_wfc1 && hipEventCreateWithFlags(&ev_wfc1, echr::sync_event_flags()) != hipSuccess) { (void)hipGetLastError(); ev_wfc1 = nullptr; }
        if (ev_wfc1) {
            H2PackJob pj = pack_rows(a->w_fc1, Df, Df, Df, w.PK_WFC1);
            RC(h2_pack_multi(&pj, 1, from));
            if (hipEventRecord(ev_wfc1, from) != hipSuccess) { set_error("tsrm_fwd: event record failed"); return -5; }
            ready = ev_wfc1;
        }
    }
    RC(position_branch(a, w, true, packed_pos, sp, ready));
    g_pos_early_ws = a->ws;
    return 0;
}

static int tsrm_fwd_impl(const echr_tsrm_args* a, const echr_dropout* drop, void* stream, const float* x_given, const float* pos_given) {
    hipStream_t st = (hipStream_t)stream;
    const int N = a->N, Din = a->Din, Df = a->Df, Do = a->Do, G = a->G;
    const int NN = N * N, dgq = Df / G, dgo = Do / G;
    TsrmWs w = carve(N, Din, Df, Do, G, a->ws);
    echr_gemm_desc d;
    const bool early = !x_given && !a->inference && g_pos_early_ws == a->ws;      // tsrm_position_early already runs the position branch
    g_pos_early_ws = nullptr;
    const int mode = a->fst_mode;
    ECHR_REQUIRE(mode >= 0 && mode <= 4, "tsrm_fwd: fst_mode must be 0..4");
    const bool posit = mode != 4;                 // use_posit = 0: no position branch, the gates are never read
    if (early) {          // X and Q | K | XW: the split-K products below accumulate into zeros; GATE is being STORED by the early branch
        float* zp[2] = {w.X, w.Q};
        const long zn[2] = {(long)(w.GATE - w.X), (long)((w.X + w.zero_floats) - w.Q)};
        RC(fill_zero_multi(zp, zn, 2, st));
    } else
    RC(fill_zero(w.X, w.zero_floats, st));           // X | GATE | Q | K | XW: the split-K products below accumulate into zeros
    if (x_given) {
        if (hipMemcpyAsync(w.X, x_given, (size_t)N * Df * sizeof(float), hipMemcpyDeviceToDevice, st) != hipSuccess ||
            hipMemcpyAsync(w.POS, pos_given, (size_t)NN * Df * sizeof(float), hipMemcpyDeviceToDevice, st) != hipSuccess) {
            set_error("tsrm_attn_fwd: input copy failed");
            return -5;
        }
    }
    // pairwise position features -> per-head gates (:39-41, :108-116): independent of the event features, so the branch runs on the
    // library's helper stream beside the embedding / query / key products (it needs the zero fill above: GATE accumulates)
    hipStream_t sp = (config().tsrm_fork && !early && posit) ? aux_fork(st) : nullptr;
    const bool fork = sp != nullptr || early;
    if (!sp) sp = st;
    long trc = 0, trl = 0;
    const long trows = (x_given || !posit) ? 0 : pair_table_rows(a, &trc, &trl);
    const bool packed_pos = !x_given && !trows && config().gemm_h2 && NN >= 4096 && posemb_packed_ok(N, Df);
    if (trows) {
        // inference over many pairs: tabulated fc1 (see pair_phi_kernel).  The tables live in the workspace regions the dense path would
        // fill (POS: F_c | F_l; P1: phi_c | phi_l and the packed operands) -- (rc + rl) <= N*N / 2 rows, so they fit
        d = desc_nt(a->ech, Din, a->w_emb, Din, w.X, Df, N, Df, Din); d.bias = a->b_emb; d.split_k = -1; d.beta = 1.f;
        RC(gemm(d, st));
        const int Lm1 = a->max_len + 1, Kh = Df / 2;
        float* FCt = w.POS;
        float* FLt = FCt + trc * Df;
        float* phc = w.P1;
        float* phl = phc + trc * Kh;
        float* pkc = phl + trl * Kh;
        float* pkl = pkc + h2_floats((int)trc, Kh);
        float* pkw0 = pkl + h2_floats((int)trl, Kh);
        float* pkw1 = pkw0 + h2_floats(Df, Kh);
        ECHR_REQUIRE(pkw1 + h2_floats(Df, Kh) <= w.P1 + NN * Df, "tsrm_fwd: pair tables do not fit the workspace");
        hipLaunchKernelGGL(pair_phi_kernel, dim3((unsigned)((trc * (Df / 4) + 255) / 256)), dim3(256), 0, sp, 0, Lm1, trc, Df, phc);
        hipLaunchKernelGGL(pair_phi_kernel, dim3((unsigned)((trl * (Df / 4) + 255) / 256)), dim3(256), 0, sp, 1, Lm1, trl, Df, phl);
        RC(check_launch("pair_phi"));
        H2PackJob pj[4] = {pack_rows(phc, Kh, (int)trc, Kh, pkc), pack_rows(phl, Kh, (int)trl, Kh, pkl),
                           pack_rows(a->w_fc1, Df, Df, Kh, pkw0), pack_rows(a->w_fc1 + Kh, Df, Df, Kh, pkw1)};
        RC(h2_pack_multi(pj, 4, sp));
        echr_gemm_desc t2[2] = {desc_h2(pkc, pkw0, FCt, Df, (int)trc, Df, Kh), desc_h2(pkl, pkw1, FLt, Df, (int)trl, Df, Kh)};
        t2[0].split_k = t2[1].split_k = 1;
        t2[0].bias = a->b_fc1;          // b1 is added once, in F_c
        RC(gemm(t2[0], sp));
        RC(gemm(t2[1], sp));
        const int tiles = (int)((NN + 15) / 16);
        const int grid = tiles / 4 < 2048 ? (tiles + 3) / 4 : 2048;
        if (Df == 512) hipLaunchKernelGGL(pair_gate_kernel<32>, dim3(grid), dim3(256), 0, sp, FCt, FLt, a->ev_start, a->ev_len, N, Lm1, a->max_span, a->w_fc2, (long)Df, a->b_fc2, w.GATE, (long)G, G, tiles);
        else hipLaunchKernelGGL(pair_gate_kernel<16>, dim3(grid), dim3(256), 0, sp, FCt, FLt, a->ev_start, a->ev_len, N, Lm1, a->max_span, a->w_fc2, (long)Df, a->b_fc2, w.GATE, (long)G, G, tiles);
        RC(check_launch("pair_gate"));
    }
    if (!x_given && !trows) {
        // event embedding (:44)
        d = desc_nt(a->ech, Din, a->w_emb, Din, w.X, Df, N, Df, Din); d.bias = a->b_emb; d.split_k = -1; d.beta = 1.f;
        RC(gemm(d, st));
    }
    if (!trows && !early && posit) RC(position_branch(a, w, !x_given, packed_pos, sp));
    // query / key / (pre-applied) output projection of X: one grouped launch when the three problems have one shape
    {
        echr_gemm_desc q3[3];
        q3[0] = desc_nt(w.X, Df, a->w_q, Df, w.Q, Df, N, Df, Df); q3[0].bias = a->b_q;
        q3[1] = desc_nt(w.X, Df, a->w_k, Df, w.K, Df, N, Df, Df); q3[1].bias = a->b_k;
        q3[2] = desc_nt(w.X, Df, a->w_out, Df, w.XW, Do, N, Do, Df);
        for (int i = 0; i < 3; ++i) { q3[i].split_k = -1; q3[i].beta = 1.f; }
        if (Do == Df) RC(gemm_grouped(q3, 3, st));
        else for (int i = 0; i < 3; ++i) RC(gemm(q3[i], st));
    }
    const DropCfg dc = make_drop(drop, drop ? drop->p_tsrm : 0.f);
    if (head_fused_ok(N, Df, Do, G)) {
        // few events: affinities, gated softmax, dropout and the weighted sum in one launch, one wave per (event, head) (:138-160)
        if (fork) RC(aux_join(st));
        hipLaunchKernelGGL(tsrm_rowhead_fwd_kernel, dim3(N, G), dim3(64), 0, st, w.Q, w.K, w.XW, w.GATE, a->b_out, w.AFF, w.WSM, w.WD, a->out,
                           N, Df, Do, G, 1.0f / sqrtf((float)dgq), dc, mode);
        return check_launch("tsrm_rowhead_fwd");
    }
    // per-head scaled affinities AFF[g] = Q_g . K_g^T / sqrt(dgq)   (:138-140)
    d = desc_nt(w.Q, Df, w.K, Df, w.AFF, N, N, N, dgq);
    d.batch = G; d.bsa = dgq; d.bsb = dgq; d.bsc = (long)NN; d.alpha = 1.0f / sqrtf((float)dgq);
    RC(gemm(d, st));
    if (fork) RC(aux_join(st));
    const size_t sm_rows = (size_t)N * (G + 1) * sizeof(float);
    // 0 = not asked yet, 1 = the device grants the row kernel's dynamic LDS, 2 = refused (then the per-(row, head) kernel serves every size)
    static std::atomic<int> rows_attr{0};
    if (N >= 128 && sm_rows <= 150 * 1024 && rows_attr.load(std::memory_order_relaxed) == 0) {
        const bool ok = hipFuncSetAttribute(reinterpret_cast<const void*>(tsrm_softmax_rows_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) == hipSuccess;
        if (!ok) (void)hipGetLastError();
        rows_attr.store(ok ? 1 : 2, std::memory_order_relaxed);
    }
    if (N >= 128 && sm_rows <= 150 * 1024 && rows_attr.load(std::memory_order_relaxed) == 1) {
        hipLaunchKernelGGL(tsrm_softmax_rows_kernel, dim3(N), dim3(256), sm_rows, st, w.GATE, w.AFF, w.WSM, w.WD, N, G, dc, mode);
    } else {
        hipLaunchKernelGGL(tsrm_softmax_fwd_kernel, dim3(N, G), dim3(64), 0, st, w.GATE, w.AFF, w.WSM, w.WD, N, G, dc, mode);
    }
    RC(check_launch("tsrm_softmax_fwd"));
    // OUT[:, g] = WD_g . XW_g + b_out_g
    d = desc_nn(w.WD, N, w.XW, Do, a->out, Do, N, dgo, N);
    d.batch = G; d.bsa = (long)NN; d.bsb = dgo; d.bsc = dgo; d.bias = a->b_out; d.bs_bias = dgo;
    RC(gemm(d, st));
    return 0;
}

extern "C" int echr_tsrm_bwd(const echr_tsrm_args* a, const echr_tsrm_grads* g, const echr_dropout* drop, void* stream) {
    return tsrm_bwd_parts(a, g, drop, stream, 0);
}
// part 0: the whole backward.  echr_train_step's joint mode (step.hip) wants d ech -- the gradient the proposal encoder waits for -- as early
// as possible: part 1 = the chain that leads to it (per-head attention backward, d X, d ech) on `stream`; part 2 = every parameter gradient
// (position MLP, projections, embedding, biases), issued later on a helper stream passed as `stream`.  Parts need zeroed gradient buffers.
static bool g_tsrm_nojoin = false;
void echr::tsrm_bwd_defer_join(bool on) { g_tsrm_nojoin = on; }
int echr::tsrm_bwd_parts(const echr_tsrm_args* a, const echr_tsrm_grads* g, const echr_dropout* drop, void* stream, int part) {
    RC(check(a, "tsrm_bwd"));
    ECHR_REQUIRE(g && g->g_out && g->ws_bwd, "tsrm_bwd: missing buffers");
    ECHR_REQUIRE(part == 0 || (g->zeroed && g->g_ech), "tsrm_bwd: the two-piece form needs zeroed gradient buffers and g_ech");
    const bool main_part = part != 2, rest_part = part != 1;
    hipStream_t st = (hipStream_t)stream;
    const int N = a->N, Din = a->Din, Df = a->Df, Do = a->Do, G = a->G;
    const int NN = N * N, dgq = Df / G, dgo = Do / G;
    const float scale = 1.0f / sqrtf((float)dgq);
    TsrmWs w = carve(N, Din, Df, Do, G, a->ws);
    TsrmWsB b = carve_b(N, Din, Df, Do, G, g->ws_bwd);
    echr_gemm_desc d;
    const bool z = g->zeroed != 0;             // gradient buffers pre-zeroed by the caller: accumulate, no fills
    const float zb = z ? 1.f : 0.f;
    if (!z) RC(colsum(g->g_out, Do, N, Do, g->g_b_out, false, st));
    const DropCfg dc = make_drop(drop, drop ? drop->p_tsrm : 0.f);
    const bool heads = head_fused_ok(N, Df, Do, G);
    const int mode = a->fst_mode;
    ECHR_REQUIRE(mode >= 0 && mode <= 4, "tsrm_bwd: fst_mode must be 0..4");
    const bool posit = mode != 4;                 // use_posit = 0: pair_pos_fc1 / fc2 are not part of the graph (their gradients stay untouched)
    if (main_part) {
    {   // accumulated (split-K) outputs zeroed by one launch: dX and, when requested, d ech
        float* zp[2] = {b.DX, g->g_ech};
        const long zn[2] = {(long)N * Df, (long)N * Din};
        RC(fill_zero_multi(zp, zn, g->g_ech ? 2 : 1, st));
    }
    if (heads) {
        // few events: d WD, the softmax backward and d Q per (event, head) wave; then d XW and d K per (column, head) wave
        hipLaunchKernelGGL(tsrm_rowhead_bwd_kernel, dim3(N, G), dim3(64), 0, st, g->g_out, w.K, w.XW, w.GATE, w.AFF, w.WSM, b.DGATE, b.DAFF, b.DQ,
                           N, Df, Do, G, scale, dc, mode);
        hipLaunchKernelGGL(tsrm_colhead_bwd_kernel, dim3(N, G), dim3(64), 0, st, g->g_out, w.Q, w.WD, b.DAFF, b.DXW, b.DK, N, Df, Do, G, scale);
        RC(check_launch("tsrm_rowhead_bwd"));
    } else {
    // dWD_g = dOUT_g . XW_g^T ; dXW_g = WD_g^T . dOUT_g
    d = desc_nt(g->g_out, Do, w.XW, Do, b.DWD, N, N, N, dgo);
    d.batch = G; d.bsa = dgo; d.bsb = dgo; d.bsc = (long)NN;
    RC(gemm(d, st));
    d = desc_tn(w.WD, N, g->g_out, Do, b.DXW, Do, N, dgo, N);
    d.batch = G; d.bsa = (long)NN; d.bsb = dgo; d.bsc = dgo;
    RC(gemm(d, st));
    hipLaunchKernelGGL(tsrm_softmax_bwd_kernel, dim3(N, G), dim3(64), 0, st, w.GATE, w.AFF, w.WSM, b.DWD, b.DGATE, b.DAFF, N, G, dc, mode);
    RC(check_launch("tsrm_softmax_bwd"));
    }
    }
    // The position-MLP gradients (d W_fc2, d P1, d W_fc1: 4.3 GF over the N^2 pairs) depend on d GATE alone: they run on the decoder's
    // prepare stream -- idle during a backward pass -- beside the query / key / embedding chain below (ten dependent small launches)
    // (round 5: off by default.  With the token-embedding chain on the prepare stream behind the LSTM-layer stage (decoder.hip, ECHR_DXT_STREAM)
    // the three streams of the backward tail end together when THIS stream keeps the position MLP: 1.48 vs 1.54 ms per iteration, same box)
    static const bool fork2_off = [] { const char* e = getenv("ECHR_TSRM_FORK2"); return !(e && e[0] == '1'); }();      // A/B switch
    hipStream_t sp = (config().tsrm_fork && !fork2_off && part == 0 && posit) ? aux2_fork(st) : nullptr;
    const bool fork2 = sp != nullptr;
    if (!fork2) sp = st;
    // one streaming pass instead of two latency-bound fp32 products + two column sums (pair_mlp_bwd_kernel; ECHR_TSRM_PAIR_BWD=0: the products)
    static const bool pm_off = [] { const char* e = getenv("ECHR_TSRM_PAIR_BWD"); return e && e[0] == '0'; }();
    const bool pair_fused = posit && z && !pm_off && Df == 512 && G == 16 && ((reinterpret_cast<uintptr_t>(w.P1) | reinterpret_cast<uintptr_t>(a->w_fc2) | reinterpret_cast<uintptr_t>(b.DP1)) & 15) == 0;
    if (rest_part && posit) {
        // position MLP (depends on d GATE only)
        if (pair_fused) {
            hipLaunchKernelGGL(pair_mlp_bwd_kernel, dim3((NN + PM_ROWS - 1) / PM_ROWS), dim3(256), 0, sp, b.DGATE, w.P1, a->w_fc2, b.DP1, b.PMP, NN);
            RC(check_launch("pair_mlp_bwd"));
        } else {
        d = desc_tn(b.DGATE, G, w.P1, Df, g->g_w_fc2, Df, G, Df, NN); d.beta = zb; d.split_k = -1;
        RC(gemm(d, sp));
        d = desc_nn(b.DGATE, G, a->w_fc2, Df, b.DP1, Df, NN, Df, G); d.act = ECHR_ACT_MUL_DTANH; d.aux = w.P1; d.ld_aux = Df;
        RC(gemm(d, sp));
        }
        if (config().gemm_h2 && NN >= 1024) {
            H2PackJob pj[2] = {pack_cols(b.DP1, Df, Df, NN, b.PK_DP1T), pack_cols(w.POS, Df, Df, NN, b.PK_POST)};
            RC(h2_pack_multi(pj, 2, sp));
            d = desc_h2(b.PK_DP1T, b.PK_POST, g->g_w_fc1, Df, Df, Df, NN);
        } else {
            d = desc_tn(b.DP1, Df, w.POS, Df, g->g_w_fc1, Df, Df, Df, NN); d.split_k = -1;
        }
        d.beta = zb;
        RC(gemm(d, sp));
    }
    // dQ_g = scale * dAFF_g . K_g ; dK_g = scale * dAFF_g^T . Q_g
    if (!heads && main_part) {
    d = desc_nn(b.DAFF, N, w.K, Df, b.DQ, Df, N, dgq, N);
    d.batch = G; d.bsa = (long)NN; d.bsb = dgq; d.bsc = dgq; d.alpha = scale;
    RC(gemm(d, st));
    d = desc_tn(b.DAFF, N, w.Q, Df, b.DK, Df, N, dgq, N);
    d.batch = G; d.bsa = (long)NN; d.bsb = dgq; d.bsc = dgq; d.alpha = scale;
    RC(gemm(d, st));
    }
    // dX = dQ . Wq + dK . Wk + dXW . Wout: three problems adding into one output -> one grouped launch
    if (main_part) {
        echr_gemm_desc x3[3];
        x3[0] = desc_nn(b.DQ, Df, a->w_q, Df, b.DX, Df, N, Df, Df);
        x3[1] = desc_nn(b.DK, Df, a->w_k, Df, b.DX, Df, N, Df, Df);
        x3[2] = desc_nn(b.DXW, Do, a->w_out, Df, b.DX, Df, N, Df, Do);
        for (int i = 0; i < 3; ++i) { x3[i].split_k = -1; x3[i].beta = 1.f; }      // dX zeroed above
        if (Do == Df && Df > 32) RC(gemm_grouped(x3, 3, st));
        else for (int i = 0; i < 3; ++i) RC(gemm(x3[i], st));
        if (part == 1) {          // d ech right behind d X: nothing else of this call is on the way
            d = desc_nn(b.DX, Df, a->w_emb, Din, g->g_ech, Din, N, Din, Df); d.split_k = -1; d.beta = 1.f;
            return gemm(d, st);
        }
    }
    // projection weights (three same-shaped products of X^T)
    {
        echr_gemm_desc w3[3];
        w3[0] = desc_tn(b.DQ, Df, w.X, Df, g->g_w_q, Df, Df, Df, N);
        w3[1] = desc_tn(b.DK, Df, w.X, Df, g->g_w_k, Df, Df, Df, N);
        w3[2] = desc_tn(b.DXW, Do, w.X, Df, g->g_w_out, Df, Do, Df, N);
        for (int i = 0; i < 3; ++i) { w3[i].beta = zb; w3[i].split_k = -1; }
        if (Do == Df) RC(gemm_grouped(w3, 3, st));
        else for (int i = 0; i < 3; ++i) RC(gemm(w3[i], st));
    }
    // event embedding
    d = desc_tn(b.DX, Df, a->ech, Din, g->g_w_emb, Din, Df, Din, N); d.beta = zb; d.split_k = -1;
    RC(gemm(d, st));
    // bias gradients: all six column sums in one launch when the gradient buffers accumulate.  With the position branch forked onto the
    // prepare stream its two sums (d b_fc2, d b_fc1: over the N^2 pair rows it just produced) run THERE, behind the products, and this stream
    // does not join: nothing later on it reads the branch's outputs, and echr_stream_join / the next library call wait for the prepare
    // stream's event -- the caller's chain then ends with its own last product instead of with the branch's (it used to: join, six sums, d ech).
    // Only for callers that keep the workspaces alive until that join (echr_train_step: tsrm_bwd_defer_join); the plain entry joins here
    if (z && fork2 && g_tsrm_nojoin) {
        const ColsumJob cp[2] = {{b.DGATE, G, NN, G, g->g_b_fc2, nullptr, nullptr}, {b.DP1, Df, NN, Df, g->g_b_fc1, nullptr, nullptr}};
        const int nblk = (NN + PM_ROWS - 1) / PM_ROWS;
        const ColsumJob cq[3] = {{b.PMP, PM_LD, nblk, G * Df, g->g_w_fc2, nullptr, nullptr}, {b.PMP + G * Df, PM_LD, nblk, Df, g->g_b_fc1, nullptr, nullptr},
                                 {b.PMP + G * Df + Df, PM_LD, nblk, G, g->g_b_fc2, nullptr, nullptr}};
        if (pair_fused) RC(colsum_multi(cq, 3, sp)); else RC(colsum_multi(cp, 2, sp));
        RC(aux2_publish());
        const ColsumJob cj[4] = {{g->g_out, Do, N, Do, g->g_b_out, nullptr, nullptr}, {b.DQ, Df, N, Df, g->g_b_q, nullptr, nullptr},
                                 {b.DK, Df, N, Df, g->g_b_k, nullptr, nullptr},        {b.DX, Df, N, Df, g->g_b_emb, nullptr, nullptr}};
        RC(colsum_multi(cj, 4, st));
    } else {
    if (fork2) RC(aux2_join(st));
    if (z) {
        const ColsumJob cj[6] = {{g->g_out, Do, N, Do, g->g_b_out, nullptr, nullptr}, {b.DQ, Df, N, Df, g->g_b_q, nullptr, nullptr},
                                 {b.DK, Df, N, Df, g->g_b_k, nullptr, nullptr},        {b.DX, Df, N, Df, g->g_b_emb, nullptr, nullptr},
                                 {b.DGATE, G, NN, G, g->g_b_fc2, nullptr, nullptr},    {b.DP1, Df, NN, Df, g->g_b_fc1, nullptr, nullptr}};
        ColsumJob cf[7] = {cj[0], cj[1], cj[2], cj[3]};
        const int nblk = (NN + PM_ROWS - 1) / PM_ROWS;          // pair_fused: the per-block partials of pair_mlp_bwd_kernel -> d W_fc2, d b_fc1, d b_fc2
        cf[4] = ColsumJob{b.PMP, PM_LD, nblk, G * Df, g->g_w_fc2, nullptr, nullptr};
        cf[5] = ColsumJob{b.PMP + G * Df, PM_LD, nblk, Df, g->g_b_fc1, nullptr, nullptr};
        cf[6] = ColsumJob{b.PMP + G * Df + Df, PM_LD, nblk, G, g->g_b_fc2, nullptr, nullptr};
        if (pair_fused) RC(colsum_multi(cf, 7, st)); else RC(colsum_multi(cj, posit ? 6 : 4, st));
    } else {
        RC(colsum(b.DQ, Df, N, Df, g->g_b_q, false, st));
        RC(colsum(b.DK, Df, N, Df, g->g_b_k, false, st));
        if (posit) {
        RC(colsum(b.DGATE, G, NN, G, g->g_b_fc2, false, st));
        RC(colsum(b.DP1, Df, NN, Df, g->g_b_fc1, false, st));
        }
        RC(colsum(b.DX, Df, N, Df, g->g_b_emb, false, st));
    }
    }
    if (g->g_ech && part == 0) {
        d = desc_nn(b.DX, Df, a->w_emb, Din, g->g_ech, Din, N, Din, Df); d.split_k = -1; d.beta = 1.f;
        RC(gemm(d, st));
    }
    return 0;
}
