// fp32 GEMM on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32), wave64, LDS-tiled.
//
// One kernel template serves every dense projection of the ECHR hot path: NT (x . W^T,
// nn.Linear forward), NN (dY . W, data gradients) and TN (dY^T . X, weight gradients) through
// per-operand (row, k) strides, plus strided batches (the 16 TSRM heads), split-K with fp32 atomics
// for skinny/deep shapes, and a fused epilogue (two biases, a row-broadcast addend, tanh,
// tanh-derivative multiply, output row remap for the [t,n] -> [n,t] log-prob layout).
//
// MFMA 32x32x2 f32 operand maps (cdna_hip_programming.md section 3): lane l supplies A[i=l&31][k=l>>5] and
// B[k=l>>5][j=l&31]; accumulator reg r of lane l is C[(r&3) + 8*(r>>2) + 4*(l>>5)][l&31].
// LDS tiles are stored k-major ([BK][BM+pad]) so the 32 lanes of a half-wave read 32 consecutive
// floats (conflict-free ds_read_b32); pad = 1 when the tile is filled by transposing float4 loads
// along k (scatter of b32 writes, conflict-free at stride BM+1), pad = 4 when filled along m (b128 writes).
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include "echr_common.h"
#include "echr_internal.h"

namespace echr {

static thread_local int g_det_depth = 0;
bool deterministic_gemm() { return g_det_depth > 0 || det_mode(); }
DeterministicScope::DeterministicScope() { ++g_det_depth; }
DeterministicScope::~DeterministicScope() { --g_det_depth; }

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BK = 32;

constexpr int GEMM_MAXG = 8;
struct GemmParams {
    const float* A; const float* B; float* C;
    int M, N, K;
    long sam, sak, sbk, sbn, ldc;
    long bsa, bsb, bsc;
    float alpha, beta;
    const float* bias; long bs_bias; const float* bias2;
    const float* addend; int add_mod; long ld_add;
    int act; const float* aux; long ld_aux;
    int rowmap_mod, rowmap_mul;
    const int* rowidx; int rowidx_max;      // optional output-row scatter: row i accumulates into C[rowidx[i]] (atomic adds, duplicates allowed)
    int split_k, k_tiles_per_split;
    int tiles_m, tiles_n;
    int xcd_n, xr_m, xr_n;      // h2 kernel: XCD grid columns, tiles per XCD rectangle (rows, cols)
    int vecA, vecB;
    int diag;                  // diagnostic ablation of the h2 kernels (config diag_skip bits 64 / 128; results WRONG): 1 = loads only, 2 = compute only
    // grouped launch: up to GEMM_MAXG problems in one grid (blockIdx.z = group * split_k + k-slice); per-group operands.  The problems share
    // K and the epilogue mode; h2 problems may differ in M and N (the grid is sized for the largest, smaller ones retire their spare tiles)
    int ngroup;
    const float* gA[GEMM_MAXG]; const float* gB[GEMM_MAXG]; float* gC[GEMM_MAXG];
    long gsbk[GEMM_MAXG], gsbn[GEMM_MAXG], gldc[GEMM_MAXG];
    int gN[GEMM_MAXG], gM[GEMM_MAXG];
    const float* gbias[GEMM_MAXG]; const float* gbias2[GEMM_MAXG]; const float* gaddend[GEMM_MAXG];
};

__device__ __forceinline__ GemmParams select_group(const GemmParams& pin, int& z) {
    GemmParams p = pin;
    if (pin.ngroup > 1) {
        const int per = pin.split_k;           // batch == 1 in grouped launches
        const int gi = z / per;
        z = z % per;
        p.A = pin.gA[gi]; p.B = pin.gB[gi]; p.C = pin.gC[gi];
        p.sbk = pin.gsbk[gi]; p.sbn = pin.gsbn[gi]; p.ldc = pin.gldc[gi]; p.N = pin.gN[gi]; p.M = pin.gM[gi];
        p.bias = pin.gbias[gi]; p.bias2 = pin.gbias2[gi]; p.addend = pin.gaddend[gi];
    }
    return p;
}

// Shared epilogue: C = act(alpha*acc + beta*C + biases + addend) with optional output row remap; split-K slices add atomically.
__device__ __forceinline__ void epilogue_elem(const GemmParams& p, float* __restrict__ C, bool first_split, float cb, int row, int col, float a) {
    float v = p.alpha * a;
    int orow = row;
    if (p.rowmap_mod > 0) orow = (row % p.rowmap_mod) * p.rowmap_mul + row / p.rowmap_mod;
    if (p.rowidx) orow = min(max(p.rowidx[row], 0), p.rowidx_max);
    float* dst = C + (long)orow * p.ldc + col;
    if (p.split_k > 1 || p.rowidx) {
        if (first_split) {
            v += cb;
            if (p.addend) v += p.addend[(long)(row % p.add_mod) * p.ld_add + col];
        }
        atomicAdd(dst, v);
    } else {
        v += cb;
        if (p.addend) v += p.addend[(long)(row % p.add_mod) * p.ld_add + col];
        if (p.beta != 0.f) v += p.beta * *dst;
        if (p.act == ECHR_ACT_TANH) v = tanhf(v);
        else if (p.act == ECHR_ACT_MUL_DTANH) {
            float t = p.aux[(long)row * p.ld_aux + col];
            v *= (1.f - t * t);
        }
        *dst = v;
    }
}
template <int TM, int TN>
__device__ __forceinline__ void epilogue(const GemmParams& p, f32x16 (&acc)[TM][TN], float* __restrict__ C, int b, int ks, int m0,
                                         int n0, int wm, int wn, int lane) {
    const int khalf = lane >> 5, l31 = lane & 31;
    const bool first_split = (ks == 0);
    const float* bias = p.bias ? p.bias + (long)b * p.bs_bias : nullptr;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = n0 + wn + j * 32 + l31;
            if (col >= p.N) continue;
            float cb = 0.f;
            if (first_split) {
                if (bias) cb += bias[col];
                if (p.bias2) cb += p.bias2[col];
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
                if (row >= p.M) continue;
                epilogue_elem(p, C, first_split, cb, row, col, acc[i][j][r]);
            }
        }
}
// the same for 16x16 MFMA accumulators: register r of tile (i, j) = row 16 i + 4 (lane >> 4) + r, column 16 j + (lane & 15)
typedef float f32x4v __attribute__((ext_vector_type(4)));
template <int TM, int TN>
__device__ __forceinline__ void epilogue16(const GemmParams& p, f32x4v (&acc)[TM][TN], float* __restrict__ C, int b, int ks, int m0,
                                           int n0, int wm, int wn, int lane) {
    const int l15 = lane & 15, rq = lane >> 4;
    const bool first_split = (ks == 0);
    const float* bias = p.bias ? p.bias + (long)b * p.bs_bias : nullptr;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = n0 + wn + j * 16 + l15;
        if (col >= p.N) continue;
        float cb = 0.f;
        if (first_split) {
            if (bias) cb += bias[col];
            if (p.bias2) cb += p.bias2[col];
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + wm + i * 16 + 4 * rq + r;
                if (row >= p.M) continue;
                epilogue_elem(p, C, first_split, cb, row, col, acc[i][j][r]);
            }
    }
}

// Fill registers with one [BMN x BK] operand tile.  KC: k is the contiguous axis of the source.
template <int BMN, bool KC, int NT>
__device__ __forceinline__ void load_tile(float4 (&r)[BMN * 8 / NT], const float* __restrict__ P, long s_mn, long s_k,
                                          int mn0, int k0, int MN, int K, int kend, bool vec, int tid) {
    if (vec && mn0 + BMN <= MN && k0 + BK <= kend) {      // interior tile: unconditional 16-byte loads, all in flight together
#pragma unroll
        for (int p = 0; p < BMN * 8 / NT; ++p) {
            const int f = tid + p * NT;
            if (KC) r[p] = *reinterpret_cast<const float4*>(P + (long)(mn0 + f / (BK / 4)) * s_mn + k0 + 4 * (f % (BK / 4)));
            else    r[p] = *reinterpret_cast<const float4*>(P + (long)(k0 + f / (BMN / 4)) * s_k + mn0 + 4 * (f % (BMN / 4)));
        }
        return;
    }
#pragma unroll
    for (int p = 0; p < BMN * 8 / NT; ++p) {
        int f = tid + p * NT;
        int mn, k;
        if (KC) { mn = mn0 + f / (BK / 4); k = k0 + 4 * (f % (BK / 4)); }
        else    { k = k0 + f / (BMN / 4); mn = mn0 + 4 * (f % (BMN / 4)); }
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (KC) {
            if (mn < MN) {
                const float* src = P + (long)mn * s_mn + k;
                if (vec && k + 3 < kend) v = *reinterpret_cast<const float4*>(src);
                else {
                    if (k < kend) v.x = src[0];
                    if (k + 1 < kend) v.y = src[1];
                    if (k + 2 < kend) v.z = src[2];
                    if (k + 3 < kend) v.w = src[3];
                }
            }
        } else {
            if (k < kend) {
                const float* src = P + (long)k * s_k + mn;
                if (vec && mn + 3 < MN) v = *reinterpret_cast<const float4*>(src);
                else {
                    if (mn < MN) v.x = src[0];
                    if (mn + 1 < MN) v.y = src[1];
                    if (mn + 2 < MN) v.z = src[2];
                    if (mn + 3 < MN) v.w = src[3];
                }
            }
        }
        r[p] = v;
    }
}

template <int BMN, bool KC, int NT>
__device__ __forceinline__ void store_tile(const float4 (&r)[BMN * 8 / NT], float* __restrict__ S, int tid) {
    constexpr int LD = BMN + (KC ? 1 : 4);
#pragma unroll
    for (int p = 0; p < BMN * 8 / NT; ++p) {
        int f = tid + p * NT;
        if (KC) {
            int mn = f / (BK / 4), k = 4 * (f % (BK / 4));
            S[(k + 0) * LD + mn] = r[p].x;
            S[(k + 1) * LD + mn] = r[p].y;
            S[(k + 2) * LD + mn] = r[p].z;
            S[(k + 3) * LD + mn] = r[p].w;
        } else {
            int k = f / (BMN / 4), mn = 4 * (f % (BMN / 4));
            *reinterpret_cast<float4*>(&S[k * LD + mn]) = r[p];
        }
    }
}

template <int BM, int BN, int WM, int WN, bool AKC, bool BKC>
__global__ __launch_bounds__((BM / WM) * (BN / WN) * 64) void gemm_f32_kernel(GemmParams pin) {
    constexpr int NT = (BM / WM) * (BN / WN) * 64;
    int z = blockIdx.z;
    const GemmParams p = select_group(pin, z);
    constexpr int LDA = BM + (AKC ? 1 : 4);
    constexpr int LDB = BN + (BKC ? 1 : 4);
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int WAVES_N = BN / WN;
    static_assert(NT == 256 || NT == 512, "4 or 8 waves per workgroup");
    __shared__ __attribute__((aligned(16))) float smem[BK * LDA + BK * LDB];
    float* As = smem;
    float* Bs = smem + BK * LDA;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = (wave / WAVES_N) * WM;
    const int wn = (wave % WAVES_N) * WN;

    // XCD-aware tile order: workgroups b and b+8 share an XCD (round-robin dispatch), so give each XCD a
    // contiguous run of tiles (neighbouring tiles share an A panel -> L2 hits).  Bijective for any grid.
    const int nwg = p.tiles_m * p.tiles_n;
    int bid = blockIdx.x;
    {
        int q = nwg / 8, r = nwg % 8, x = bid % 8;
        bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + bid / 8;
    }
    const int m0 = (bid / p.tiles_n) * BM;
    const int n0 = (bid % p.tiles_n) * BN;
    const int b = z / p.split_k;
    const int ks = z % p.split_k;

    const float* A = p.A + (long)b * p.bsa;
    const float* B = p.B + (long)b * p.bsb;
    float* C = p.C + (long)b * p.bsc;

    const int kt_total = (p.K + BK - 1) / BK;
    const int kt0 = ks * p.k_tiles_per_split;
    const int kt1 = min(kt_total, kt0 + p.k_tiles_per_split);
    const int kend = min(p.K, kt1 * BK);

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // two register sets: the tile written to LDS at step kt was requested two compute phases earlier
    float4 ra0[BM * 8 / NT], rb0[BN * 8 / NT], ra1[BM * 8 / NT], rb1[BN * 8 / NT];
    if (kt0 < kt1) {
        load_tile<BM, AKC, NT>(ra0, A, p.sam, p.sak, m0, kt0 * BK, p.M, p.K, kend, p.vecA, tid);
        load_tile<BN, BKC, NT>(rb0, B, p.sbn, p.sbk, n0, kt0 * BK, p.N, p.K, kend, p.vecB, tid);
    }
    if (kt0 + 1 < kt1) {
        load_tile<BM, AKC, NT>(ra1, A, p.sam, p.sak, m0, (kt0 + 1) * BK, p.M, p.K, kend, p.vecA, tid);
        load_tile<BN, BKC, NT>(rb1, B, p.sbn, p.sbk, n0, (kt0 + 1) * BK, p.N, p.K, kend, p.vecB, tid);
    }
    const int khalf = lane >> 5, l31 = lane & 31;
    auto compute = [&]() {
#pragma unroll
        for (int kk = 0; kk < BK / 2; ++kk) {
            float a[TM], bb[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = As[(2 * kk + khalf) * LDA + wm + i * 32 + l31];
#pragma unroll
            for (int j = 0; j < TN; ++j) bb[j] = Bs[(2 * kk + khalf) * LDB + wn + j * 32 + l31];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], bb[j], acc[i][j], 0, 0, 0);
        }
    };
    for (int kt = kt0; kt < kt1; kt += 2) {
        store_tile<BM, AKC, NT>(ra0, As, tid);
        store_tile<BN, BKC, NT>(rb0, Bs, tid);
        __syncthreads();
        if (kt + 2 < kt1) {
            load_tile<BM, AKC, NT>(ra0, A, p.sam, p.sak, m0, (kt + 2) * BK, p.M, p.K, kend, p.vecA, tid);
            load_tile<BN, BKC, NT>(rb0, B, p.sbn, p.sbk, n0, (kt + 2) * BK, p.N, p.K, kend, p.vecB, tid);
        }
        compute();
        __syncthreads();
        if (kt + 1 < kt1) {
            store_tile<BM, AKC, NT>(ra1, As, tid);
            store_tile<BN, BKC, NT>(rb1, Bs, tid);
            __syncthreads();
            if (kt + 3 < kt1) {
                load_tile<BM, AKC, NT>(ra1, A, p.sam, p.sak, m0, (kt + 3) * BK, p.M, p.K, kend, p.vecA, tid);
                load_tile<BN, BKC, NT>(rb1, B, p.sbn, p.sbk, n0, (kt + 3) * BK, p.N, p.K, kend, p.vecB, tid);
            }
            compute();
            __syncthreads();
        }
    }

    epilogue<TM, TN>(p, acc, C, b, ks, m0, n0, wm, wn, lane);
}

// ------------------------------------------------------------------------------------------------------
// fp32 GEMM through the bf16 matrix cores: each fp32 operand is split EXACTLY into three bf16 planes
// (x = hi + mid + lo, 8 significand bits each, by truncation: every residual is exactly representable), and the six
// largest of the nine plane products (hh, hm, mh, mm, hl, lh) are accumulated in fp32 by v_mfma_f32_32x32x16_bf16.
// Every bf16 x bf16 product is exact in fp32; the three dropped products are below 2^-23 of |x.y|, i.e. at the level of
// one fp32 rounding of the product -- the result is fp32-accurate (measured against float64 in tests/test_gpu_parity.py)
// at 16/6 = 2.7x the rate of the native fp32 MFMA.
// NT only (both operands k-contiguous, 16-byte aligned, K % 4 == 0): 128x128 tile, BK = 32, 4 waves x (2x2) 32x32 tiles.
// LDS holds the three planes of both operands as [row][32 + 8 pad] bf16 (80-byte rows: conflict-free ds_read_b128
// fragment reads, lane l reads row l&31, k = 8*(l>>5)..+7 -- the 32x32x16 A/B operand map).
// ------------------------------------------------------------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int SLD = 40;     // bf16 elements per LDS row

__device__ __forceinline__ void split_store(unsigned short* __restrict__ base, int plane_stride, int off, float4 v) {
    // hi / mid / lo planes of 4 consecutive k values -> one 8-byte LDS store per plane.  Truncation split: every residual
    // is exact.  (Non-finite inputs turn into NaN in the lower planes, i.e. a non-finite result, like the fp32 product.)
    const unsigned x0 = __float_as_uint(v.x), x1 = __float_as_uint(v.y), x2 = __float_as_uint(v.z), x3 = __float_as_uint(v.w);
    const uint2 h = make_uint2(__builtin_amdgcn_perm(x1, x0, 0x07060302u), __builtin_amdgcn_perm(x3, x2, 0x07060302u));
    const float r0 = v.x - __uint_as_float(x0 & 0xFFFF0000u), r1 = v.y - __uint_as_float(x1 & 0xFFFF0000u);
    const float r2 = v.z - __uint_as_float(x2 & 0xFFFF0000u), r3 = v.w - __uint_as_float(x3 & 0xFFFF0000u);
    const unsigned y0 = __float_as_uint(r0), y1 = __float_as_uint(r1), y2 = __float_as_uint(r2), y3 = __float_as_uint(r3);
    const uint2 m = make_uint2(__builtin_amdgcn_perm(y1, y0, 0x07060302u), __builtin_amdgcn_perm(y3, y2, 0x07060302u));
    const float s0 = r0 - __uint_as_float(y0 & 0xFFFF0000u), s1 = r1 - __uint_as_float(y1 & 0xFFFF0000u);
    const float s2 = r2 - __uint_as_float(y2 & 0xFFFF0000u), s3 = r3 - __uint_as_float(y3 & 0xFFFF0000u);
    const uint2 l = make_uint2(__builtin_amdgcn_perm(__float_as_uint(s1), __float_as_uint(s0), 0x07060302u),
                               __builtin_amdgcn_perm(__float_as_uint(s3), __float_as_uint(s2), 0x07060302u));
    *reinterpret_cast<uint2*>(base + off) = h;
    *reinterpret_cast<uint2*>(base + plane_stride + off) = m;
    *reinterpret_cast<uint2*>(base + 2 * plane_stride + off) = l;
}

// one [ROWS x 32] fp32 tile, ROWS*8/NT float4 per thread; interior tiles load unconditionally, edge tiles use clamped
// addresses + masks (both branch-free per lane; the tile class is workgroup-uniform)
template <int ROWS, int NT>
__device__ __forceinline__ void split_load(float4 (&r)[ROWS * 8 / NT], const float* __restrict__ P, long ld, int mn0, int k0, int MN,
                                           int K, int kend, int tid) {
    constexpr int L = ROWS * 8 / NT;
    if (mn0 + ROWS <= MN && k0 + BK <= kend) {
#pragma unroll
        for (int i = 0; i < L; ++i) {
            const int f = tid + i * NT;
            r[i] = *reinterpret_cast<const float4*>(P + (long)(mn0 + (f >> 3)) * ld + k0 + 4 * (f & 7));
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < L; ++i) {
        const int f = tid + i * NT;
        const int row = mn0 + (f >> 3), k = k0 + 4 * (f & 7);
        const float4 v = *reinterpret_cast<const float4*>(P + (long)min(row, MN - 1) * ld + min(k, K - 4));
        const float msk = (row < MN && k < kend) ? 1.f : 0.f;
        r[i] = make_float4(v.x * msk, v.y * msk, v.z * msk, v.w * msk);
    }
}

// 128 x BN tile, 2 x (BN/32) waves of 64x32 wave tiles (8 waves for BN = 128, 4 for BN = 64); <= 128 VGPRs so that several
// workgroups (4 waves per SIMD) share a CU and one wave's operand splitting (VALU) overlaps the other waves' MFMAs.
template <int BN>
__global__ __launch_bounds__(BN * 4, BN == 128 ? 4 : 2) void gemm_split_kernel(GemmParams pin) {
    constexpr int BM = 128, NT = BN * 4, WN_CNT = BN / 32;
    int z = blockIdx.z;
    const GemmParams p = select_group(pin, z);
    constexpr int PSA = BM * SLD, PSB = BN * SLD;                     // plane strides (bf16 elements)
    constexpr int LA = BM * 8 / NT, LB = BN * 8 / NT;
    __shared__ __attribute__((aligned(16))) unsigned short sm[3 * PSA + 3 * PSB];   // 61,440 B (BN=128) / 46,080 B (BN=64)
    unsigned short* As = sm;
    unsigned short* Bs = sm + 3 * PSA;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = (wave / WN_CNT) * 64, wn = (wave % WN_CNT) * 32;
    const int l31 = lane & 31, h = lane >> 5;
    const int nwg = p.tiles_m * p.tiles_n;
    int bid = blockIdx.x;
    {
        int q = nwg / 8, r = nwg % 8, x = bid % 8;
        bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + bid / 8;
    }
    const int m0 = (bid / p.tiles_n) * BM, n0 = (bid % p.tiles_n) * BN;
    const int b = z / p.split_k, ks = z % p.split_k;
    const float* A = p.A + (long)b * p.bsa;
    const float* B = p.B + (long)b * p.bsb;
    float* C = p.C + (long)b * p.bsc;
    const int kt_total = (p.K + BK - 1) / BK;
    const int kt0 = ks * p.k_tiles_per_split, kt1 = min(kt_total, kt0 + p.k_tiles_per_split);
    const int kend = min(p.K, kt1 * BK);

    f32x16 acc[2][1];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][0][r] = 0.f;

    float4 ra0[LA], rb0[LB], ra1[LA], rb1[LB];
    auto fetch = [&](float4 (&ra)[LA], float4 (&rb)[LB], int kt) {
        split_load<BM, NT>(ra, A, p.sam, m0, kt * BK, p.M, p.K, kend, tid);
        split_load<BN, NT>(rb, B, p.sbn, n0, kt * BK, p.N, p.K, kend, tid);
    };
    if (kt0 < kt1) fetch(ra0, rb0, kt0);
    if (kt0 + 1 < kt1) fetch(ra1, rb1, kt0 + 1);

    auto stage = [&](const float4 (&ra)[LA], const float4 (&rb)[LB]) {
#pragma unroll
        for (int i = 0; i < LA; ++i) { const int f = tid + i * NT; split_store(As, PSA, (f >> 3) * SLD + 4 * (f & 7), ra[i]); }
#pragma unroll
        for (int i = 0; i < LB; ++i) { const int f = tid + i * NT; split_store(Bs, PSB, (f >> 3) * SLD + 4 * (f & 7), rb[i]); }
    };
    auto compute = [&]() {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            bf16x8 a[2][3], bb[3];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
                a[0][pl] = *reinterpret_cast<const bf16x8*>(As + pl * PSA + (wm + l31) * SLD + 16 * s + 8 * h);
                a[1][pl] = *reinterpret_cast<const bf16x8*>(As + pl * PSA + (wm + 32 + l31) * SLD + 16 * s + 8 * h);
                bb[pl] = *reinterpret_cast<const bf16x8*>(Bs + pl * PSB + (wn + l31) * SLD + 16 * s + 8 * h);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                f32x16 c = acc[i][0];       // smallest terms first
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][2], bb[0], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], bb[2], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], bb[1], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], bb[0], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], bb[1], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], bb[0], c, 0, 0, 0);
                acc[i][0] = c;
            }
        }
    };
    for (int kt = kt0; kt < kt1; kt += 2) {
        stage(ra0, rb0);
        __syncthreads();
        if (kt + 2 < kt1) fetch(ra0, rb0, kt + 2);
        compute();
        __syncthreads();
        if (kt + 1 < kt1) {
            stage(ra1, rb1);
            __syncthreads();
            if (kt + 3 < kt1) fetch(ra1, rb1, kt + 3);
            compute();
            __syncthreads();
        }
    }
    epilogue<2, 1>(p, acc, C, b, ks, m0, n0, wm, wn, lane);
}

// ------------------------------------------------------------------------------------------------------
// "h2" operands: fp32-grade products at twice the bf16x3 rate and at fp32's byte count.  h2_pack_kernel rewrites an operand
// ONCE as two fp16 planes with a shared power-of-two scale per row and 256-wide k segment (H2_SEG blocks of 32; the block-exponent idea of the MX
// formats, applied to fp16 pairs):
//     xs = x * 2^(14 - floor(log2(max_k |x|)))      (exact; the segment maximum lands in [2^14, 2^15))
//     h1 = fp16_rne(xs),  h2 = fp16_rne(xs - h1)    (xs - h1 is exact in fp32; |xs - h1 - h2| <= 2^-24 |xs| while h2 is a
//                                                    normal fp16, i.e. for every element within 2^-17 of its segment maximum,
//                                                    and <= 2^-25 absolute (2^-39 of the segment maximum) below that)
// and the GEMM accumulates, per segment, the three fp16 MFMA products h1.h1' + h1.h2' + h2.h1' (each product exact in the
// fp32 accumulator; the dropped h2.h2' is < 2^-22 of a term) into a scratch accumulator that is then folded into the result with
// the two block scales: acc += tmp * 2^-(eA[row] + eB[col]).  Error against float64: a few 2^-24 . sum|a||b|, measured in
// tests/test_gpu_parity.py at or below the native fp32 MFMA path -- with the exponent range of fp32, because no value is ever
// held in fp16 unscaled.
//
// Packed image of a logical [R x K] operand (k = the contraction axis): chunks [ceil(R/128)][ceil(K/32)], each chunk = 2 planes x
// 128 rows x 32 k fp16 = 16,384 contiguous bytes (zero-padded in both directions), followed by the inverse scales
// [ceil(R/128)][ceil(K/32)][128] fp32 (the blocks of one segment carry the same values; the GEMM folds once per segment).  Inside a plane a row is 64 B = four 16-byte slots; logical slot s (k = 8s..8s+7) of
// row r sits at physical slot s ^ swz(r), swz = the permutation {0,2,3,1} of (r>>2)&3: a wave's ds_read_b128 fragment read (16 rows, one logical slot) then touches 16
// distinct 16-byte bank groups.  The swizzle is baked into the image, so the GEMM stages tiles with direct-to-LDS loads
// (global_load_lds_dwordx4: linear source, linear destination, no VGPR round trip, no VALU).
// ------------------------------------------------------------------------------------------------------
constexpr int H2_ROWS = 128;
constexpr int H2_PLANE = H2_ROWS * BK * 2;       // 8,192 B
constexpr int H2_CHUNK = 2 * H2_PLANE;           // 16,384 B
constexpr int H2_SCALES = H2_ROWS * 4;           // 512 B of inverse scales per chunk
constexpr int H2_SEG = 8;                        // k blocks that share one scale per row (256 k)

// slot swizzle of a row: the permutation {0,2,3,1} of (row / 4) % 4.  A ds_read_b128 is served in four groups of 16 lanes
// ({0-3,12-15,20-27}, {4-11,16-19,28-31}, ...; MI355X_MICROARCH.md LDS table); with this permutation both fragment shapes touch 16
// distinct 16-byte bank groups per lane group: the 32x32x16 one (32 rows, one logical slot per half wave) and the 16x16x32 one (16 rows x
// 4 logical slots)
__device__ __forceinline__ int h2_swz(int row) { return (0x78 >> (2 * ((row >> 2) & 3))) & 3; }

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int H2_MAX_JOBS = 12;
struct H2PackArgs { H2PackJob job[H2_MAX_JOBS]; int start[H2_MAX_JOBS + 1]; int njobs; };

// one workgroup = one scale segment: up to H2_SEG consecutive chunks (128 rows x 32 k each) of one row block, which share a scale per row.
// src element (r, k) = src[r * s_row + k * s_col]; one of the strides is 1.  k-contiguous sources need no staging: a thread owns
// (row, 16-byte slot) = 8 consecutive k of every chunk and reads them as two float4.  Row-contiguous sources (a transposing pack) go
// through an LDS tile filled by float4 reads along the rows.  The whole segment is held in registers (2 x 8 x H2_SEG floats per thread)
// so the source is read once.
struct __attribute__((packed, aligned(4))) F4U { float x, y, z, w; };
__device__ __forceinline__ float h2_rowmax(const float (&v)[8], float mx) {
#pragma unroll
    for (int j = 0; j < 8; ++j) mx = fmaxf(mx, fabsf(v[j]));
    return mx;
}
__device__ __forceinline__ void h2_emit(const float (&v)[8], float mx, unsigned char* __restrict__ chunk, float* __restrict__ inv_scales, int q, int row, int p) {
    // block exponent e = floor(log2 mx) from the bit pattern; zero / subnormal / non-finite rows keep scale 1
    const int ex = (int)((__float_as_uint(mx) >> 23) & 0xFFu);
    int e = (ex == 0 || ex == 255) ? 14 : ex - 127;
    e = max(e, 14 - 126);                        // keeps 2^(14-e) a normal fp32 (rows below 2^-112 lose bits they do not need)
    const float sc = __uint_as_float((unsigned)(127 + 14 - e) << 23);
    unsigned hw[8], lw[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float xs = v[j] * sc;
        const _Float16 h1 = (_Float16)xs;
        const _Float16 h2 = (_Float16)(xs - (float)h1);
        hw[j] = (unsigned)__builtin_bit_cast(unsigned short, h1);
        lw[j] = (unsigned)__builtin_bit_cast(unsigned short, h2);
    }
    *reinterpret_cast<uint4*>(chunk + q * 16) = make_uint4(hw[0] | (hw[1] << 16), hw[2] | (hw[3] << 16), hw[4] | (hw[5] << 16), hw[6] | (hw[7] << 16));
    *reinterpret_cast<uint4*>(chunk + H2_PLANE + q * 16) = make_uint4(lw[0] | (lw[1] << 16), lw[2] | (lw[3] << 16), lw[4] | (lw[5] << 16), lw[6] | (lw[7] << 16));
    if (p == 0) inv_scales[row] = ldexpf(1.f, e - 14);
}

__global__ __launch_bounds__(256) void h2_pack_kernel(H2PackArgs args) {
    int ji = 0;
    while (ji + 1 < args.njobs && (int)blockIdx.x >= args.start[ji + 1]) ++ji;      // work items of all jobs share one linear grid
    const H2PackJob jb = args.job[ji];
    const int KT = (jb.K + BK - 1) / BK, RB = (jb.R + H2_ROWS - 1) / H2_ROWS, KS = (KT + H2_SEG - 1) / H2_SEG;
    int ci = (int)blockIdx.x - args.start[ji];
    const int half = ci & 1;                      // rows [64 half, 64 half + 64) of the row block
    ci >>= 1;
    const int sg = ci % KS, rb = ci / KS, tid = threadIdx.x;
    const int r0 = rb * H2_ROWS, kt0 = sg * H2_SEG, nch = min(H2_SEG, KT - kt0);
    const float* __restrict__ src = jb.src;
    unsigned char* chunk0 = jb.dst + ((long)rb * KT + kt0) * H2_CHUNK;
    float* inv0 = reinterpret_cast<float*>(jb.dst + (long)RB * KT * H2_CHUNK) + ((long)rb * KT + kt0) * H2_ROWS;
    const bool aligned = ((reinterpret_cast<uintptr_t>(src) & 15) == 0);
    const int q = half * 256 + tid, row = q >> 2, pp = q & 3, sl = pp ^ h2_swz(row);      // this thread's (row, 16-byte slot)
    float v[H2_SEG][8];
    float mx = 0.f;
    if (jb.s_col == 1) {
        const bool vec = aligned && (jb.s_row & 3) == 0;
        const int r = r0 + row;
        const int rs = min(r, jb.R - 1);
        const float* srow = src + (long)(jb.gather ? jb.gather[rs] : rs) * jb.s_row;
#pragma unroll
        for (int c = 0; c < H2_SEG; ++c) {
            const int k = (kt0 + c) * BK + 8 * sl;
            if (c < nch && r < jb.R) {
                if (vec && k + 8 <= jb.K) {
                    const float4 x0 = *reinterpret_cast<const float4*>(srow + k), x1 = *reinterpret_cast<const float4*>(srow + k + 4);
                    v[c][0] = x0.x; v[c][1] = x0.y; v[c][2] = x0.z; v[c][3] = x0.w; v[c][4] = x1.x; v[c][5] = x1.y; v[c][6] = x1.z; v[c][7] = x1.w;
                } else if (k + 8 <= jb.K) {
                    // rows that are only 4-byte aligned (K = 5001): still two 16-byte loads where the target allows dword-aligned
                    // dwordx4 access, four dword loads each otherwise -- the compiler decides from the declared alignment
                    const F4U x0 = *reinterpret_cast<const F4U*>(srow + k), x1 = *reinterpret_cast<const F4U*>(srow + k + 4);
                    v[c][0] = x0.x; v[c][1] = x0.y; v[c][2] = x0.z; v[c][3] = x0.w; v[c][4] = x1.x; v[c][5] = x1.y; v[c][6] = x1.z; v[c][7] = x1.w;
                } else {
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[c][j] = (k + j < jb.K) ? srow[k + j] : 0.f;
                }
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[c][j] = 0.f;
            }
            mx = h2_rowmax(v[c], mx);
        }
    } else {
        __shared__ float tile[2][H2_ROWS / 2][BK + 1];       // two tiles: chunk c + 1 is written while chunk c is read
        const int rh = r0 + 64 * half;
        const bool vec = aligned && (jb.s_col & 3) == 0 && rh + 64 <= jb.R;
        // vec (float4 = 4 consecutive rows of one k): ALL of the segment's loads are requested up front -- a workgroup then has 16 KB in
        // flight instead of 2 (with ~2 workgroups per CU on the weight-gradient packs the kernel was bound by load latency: 2 TB/s);
        // gathered k indices are fetched first, one dependent round for the whole segment.  Otherwise: the next chunk's loads are in
        // flight while this one is transposed
        float4 xa[H2_SEG][2];
        float xs[8];
        if (vec) {
            int gk[H2_SEG][2];
#pragma unroll
            for (int c = 0; c < H2_SEG; ++c)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int k = (kt0 + c) * BK + ((tid + i * 256) >> 4);
                    gk[c][i] = (c < nch && k < jb.K) ? (jb.gather ? jb.gather[k] : k) : -1;
                }
#pragma unroll
            for (int c = 0; c < H2_SEG; ++c)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int rr = ((tid + i * 256) & 15) * 4;
                    xa[c][i] = gk[c][i] >= 0 ? *reinterpret_cast<const float4*>(src + (long)gk[c][i] * jb.s_col + rh + rr) : make_float4(0.f, 0.f, 0.f, 0.f);
                }
        }
        auto fetch = [&](int c) {
            const int k0 = (kt0 + c) * BK;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int f = tid + i * 256, k = f >> 6, rr = f & 63;
                xs[i] = (rh + rr < jb.R && k0 + k < jb.K) ? src[(long)(jb.gather ? jb.gather[k0 + k] : k0 + k) * jb.s_col + rh + rr] : 0.f;
            }
        };
        if (!vec) fetch(0);
#pragma unroll
        for (int c = 0; c < H2_SEG; ++c) {
            if (c < nch) {                 // nch is uniform over the workgroup
                float (*tl)[BK + 1] = tile[c & 1];
                if (vec) {
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        const int f = tid + i * 256, k = f >> 4, rr = (f & 15) * 4;
                        tl[rr][k] = xa[c][i].x; tl[rr + 1][k] = xa[c][i].y; tl[rr + 2][k] = xa[c][i].z; tl[rr + 3][k] = xa[c][i].w;
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const int f = tid + i * 256;
                        tl[f & 63][f >> 6] = xs[i];
                    }
                    if (c + 1 < nch) fetch(c + 1);
                }
                __syncthreads();           // tile c complete; tile (c - 1) & 1 == (c + 1) & 1 was last read before this barrier
#pragma unroll
                for (int j = 0; j < 8; ++j) v[c][j] = tl[row & 63][8 * sl + j];
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[c][j] = 0.f;
            }
            mx = h2_rowmax(v[c], mx);
        }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 1));          // the row's four slots sit in adjacent lanes
    mx = fmaxf(mx, __shfl_xor(mx, 2));
#pragma unroll
    for (int c = 0; c < H2_SEG; ++c)
        if (c < nch) h2_emit(v[c], mx, chunk0 + (long)c * H2_CHUNK, inv0 + (long)c * H2_ROWS, q, row, pp);
}

typedef __attribute__((address_space(3))) void* lds_vptr;
typedef const __attribute__((address_space(1))) void* glb_vptr;

// C[M,N] = A . B^T on h2-packed operands.  BM x 128 tile (BM = 128: 4 waves, 256: 8 waves), each wave a 64x64 block of 2x2 MFMA
// tiles; BK = 32: per k block a wave reads 8 fragments (2 row blocks x 2 planes of A and of B), issues 24 MFMAs into the scratch
// accumulators and folds them into the result with the block scales (128 VALU FMAs/MULs).
// Two LDS stages: the global_load_lds pieces of k block t+1 are in flight while block t computes; one barrier per k block (its
// vmcnt(0) retires the stage that is read next).  A CU pulls at most ~28 B/clk from its L2 (14 from the Infinity Cache), about what
// a 128x128 stage needs at full MFMA rate, so the workgroup -> tile map gives every XCD a compact rectangle of tiles (xm x xn XCD
// grid): co-running tiles then share their A and B panels through that XCD's L2.
template <int BM, int WN, int NS>
__global__ __launch_bounds__((BM / 64) * (128 / WN) * 64, (NS == 2 && BM == 128) ? 2 : 1) void gemm_h2_kernel(GemmParams pin) {
    constexpr int WAVES_N = 128 / WN, TN = WN / 32, NT = (BM / 64) * WAVES_N * 64, ACH = BM / 128;
    constexpr int PLANES = (ACH + 1) * H2_CHUNK;          // plane bytes per stage
    constexpr int STAGE = PLANES + (ACH + 1) * H2_SCALES;
    constexpr int PIECES = PLANES / (NT * 16);            // 8 (BM = 128) / 6 (BM = 256) 16-byte pieces per thread and stage
    constexpr int PPC = H2_CHUNK / (NT * 16);             // pieces per chunk: 4 / 2
    int z = blockIdx.z;
    const GemmParams p = select_group(pin, z);
    // NS LDS stages (dynamic LDS: NS * STAGE bytes): the loads of k blocks t+1 .. t+NS-1 are in flight while block t computes.
    extern __shared__ __attribute__((aligned(1024))) unsigned char sm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = (wave / WAVES_N) * 64, wn = (wave % WAVES_N) * WN;
    const int l31 = lane & 31, h = lane >> 5, sw = h2_swz(l31);
    // XCD-rectangle tile map: workgroup b runs on XCD b % 8 (round-robin dispatch)
    const int x = blockIdx.x & 7, sl = blockIdx.x >> 3;
    const int mb = (x / p.xcd_n) * p.xr_m + sl / p.xr_n, nb = (x % p.xcd_n) * p.xr_n + sl % p.xr_n;
    if (mb >= p.tiles_m || nb >= p.tiles_n) return;
    const int ks = z;
    const int KT = (p.K + BK - 1) / BK;
    const int kt0 = ks * p.k_tiles_per_split, kt1 = min(KT, kt0 + p.k_tiles_per_split);
    const int chunks_m = (p.M + H2_ROWS - 1) / H2_ROWS, chunks_n = (p.N + H2_ROWS - 1) / H2_ROWS;
    const unsigned char* Ap = reinterpret_cast<const unsigned char*>(p.A);
    const unsigned char* Bp = reinterpret_cast<const unsigned char*>(p.B);
    const unsigned char* Ag[ACH];
#pragma unroll
    for (int c = 0; c < ACH; ++c) Ag[c] = Ap + ((long)min(mb * ACH + c, chunks_m - 1) * KT + kt0) * H2_CHUNK + tid * 16;
    const unsigned char* Bg = Bp + ((long)nb * KT + kt0) * H2_CHUNK + tid * 16;
    // inverse scales: waves 0..ACH-1 fetch an A chunk's, wave ACH the B chunk's (32 lanes x 16 B = 512 B each)
    const unsigned char* Sg = nullptr;
    if (wave < ACH) Sg = Ap + (long)chunks_m * KT * H2_CHUNK + ((long)min(mb * ACH + wave, chunks_m - 1) * KT + kt0) * H2_SCALES + l31 * 16;
    else if (wave == ACH) Sg = Bp + (long)chunks_n * KT * H2_CHUNK + ((long)nb * KT + kt0) * H2_SCALES + l31 * 16;
    f32x16 acc[2][TN];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    auto stage = [&](int buf) {
        unsigned char* dst = sm + buf * STAGE + wave * 1024;
#pragma unroll
        for (int j = 0; j < PIECES; ++j) {
            const int c = j / PPC, jj = j % PPC;
            const unsigned char* src = (c < ACH ? Ag[c < ACH ? c : 0] : Bg) + jj * (NT * 16);
            __builtin_amdgcn_global_load_lds((glb_vptr)src, (lds_vptr)(dst + j * (NT * 16)), 16, 0, 0);
        }
        if (wave <= ACH && lane < 32)
            __builtin_amdgcn_global_load_lds((glb_vptr)Sg, (lds_vptr)(sm + buf * STAGE + PLANES + wave * H2_SCALES), 16, 0, 0);
#pragma unroll
        for (int c = 0; c < ACH; ++c) Ag[c] += H2_CHUNK;
        Bg += H2_CHUNK;
        Sg += H2_SCALES;
    };
    // fragment byte offsets (k half 0 / 1) inside the stage image [A chunk(s) | B chunk | A scales | B scales]
    const int rowA = (wm & 127) + l31;
    const int baseA = (wm >> 7) * H2_CHUNK + rowA * 64, baseB = ACH * H2_CHUNK + (wn + l31) * 64;
    const int o0 = ((0 + h) ^ sw) * 16, o1 = ((2 + h) ^ sw) * 16;
    const int scA = PLANES + (wm >> 7) * H2_SCALES + ((wm & 127) + 4 * h) * 4;     // + (i*32 + 8*g) * 4: rows (r&3) + 8g + 4h of block i
    const int scB = PLANES + ACH * H2_SCALES + (wn + l31) * 4;                    // + j*32*4

#pragma unroll
    for (int s0 = 0; s0 < NS - 1; ++s0)
        if (kt0 + s0 < kt1) stage(s0);
    int cur = 0;
    f32x16 tmp[2][TN];          // products of the current scale segment, in the segment's scaled units
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) tmp[i][j][r] = 0.f;
    for (int kt = kt0; kt < kt1; ++kt) {
        // this wave's loads of stage `cur` have landed once at most the NS-2 younger stages' loads are outstanding (loads retire in
        // issue order; a wave issues PIECES (+1 for the waves that fetch scales) load instructions per stage); after the barrier stage
        // `cur` is visible everywhere and the stage read during the previous k block is free again
        if (NS > 2 && kt + NS - 2 < kt1) {
            if (wave <= ACH) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * (PIECES + 1)) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * PIECES) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        if (kt + NS - 1 < kt1) stage(cur == 0 ? NS - 1 : cur - 1);
        const unsigned char* sb = sm + cur * STAGE;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int o = s ? o1 : o0;
            f16x8 a[2][2], bb[TN][2];
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) {
                a[0][pl] = *reinterpret_cast<const f16x8*>(sb + baseA + pl * H2_PLANE + o);
                a[1][pl] = *reinterpret_cast<const f16x8*>(sb + baseA + pl * H2_PLANE + o + 32 * 64);
#pragma unroll
                for (int j = 0; j < TN; ++j) bb[j][pl] = *reinterpret_cast<const f16x8*>(sb + baseB + pl * H2_PLANE + o + j * 32 * 64);
            }
            // small terms first
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) tmp[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i][1], bb[j][0], tmp[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) tmp[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i][0], bb[j][1], tmp[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) tmp[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i][0], bb[j][0], tmp[i][j], 0, 0, 0);
        }
        // at the end of a scale segment (H2_SEG k blocks, absolute k block index, so any split-K slice boundary works) fold the segment's
        // products into the result: acc += tmp * invA[row] * invB[col].  The scales are read with inline-asm ds_reads: hipcc orders a
        // plain LDS read of the glds-written scale area behind vmcnt(0), which would drain the next stage's loads (the barrier above
        // already ordered this stage's DMA).
        if (((kt + 1) & (H2_SEG - 1)) == 0 || kt + 1 == kt1) {
            float cb[TN];
            float4 ca[2][4];
            {
                const unsigned sbase = (unsigned)(size_t)(sm) + cur * STAGE;       // LDS byte address
                const unsigned aB = sbase + scB, aA = sbase + scA;
                asm volatile("ds_read_b32 %0, %1" : "=v"(cb[0]) : "v"(aB));
                if (TN == 2) asm volatile("ds_read_b32 %0, %1 offset:128" : "=v"(cb[TN - 1]) : "v"(aB));
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        asm volatile("ds_read_b128 %0, %1" : "=v"(ca[i][g]) : "v"(aA + (i * 32 + 8 * g) * 4));
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float cav[4] = {ca[i][g].x, ca[i][g].y, ca[i][g].z, ca[i][g].w};
#pragma unroll
                    for (int j = 0; j < TN; ++j)
#pragma unroll
                        for (int r4 = 0; r4 < 4; ++r4) {
                            acc[i][j][4 * g + r4] = fmaf(tmp[i][j][4 * g + r4], cav[r4] * cb[j], acc[i][j][4 * g + r4]);
                            tmp[i][j][4 * g + r4] = 0.f;
                        }
                }
        }
        cur = (cur + 1 == NS) ? 0 : cur + 1;
    }
    epilogue<2, TN>(p, acc, p.C, 0, ks, mb * BM, nb * 128, wm, wn, lane);
}

// The same product on v_mfma_f32_16x16x32_f16: 128 x 128 tile, 8 waves, each a 64 x 32 block of 4 x 2 MFMA tiles.  Same packed image
// (its swizzle was laid out for 16-row fragment reads), same staging; per k block a wave reads 12 fragments and issues 24 MFMAs of 16
// cycles.  Under sustained MFMA load the chip holds a higher clock on this shape than on 32x32x16 (MI355X_MICROARCH.md, DVFS item 7).
template <int NS>
__global__ __launch_bounds__(512, 2) void gemm_h2m16_kernel(GemmParams pin) {
    constexpr int NT = 512, STAGE = 2 * H2_CHUNK + 2 * H2_SCALES, PLANES = 2 * H2_CHUNK, PIECES = PLANES / (NT * 16), PPC = H2_CHUNK / (NT * 16);
    int z = blockIdx.z;
    const GemmParams p = select_group(pin, z);
    extern __shared__ __attribute__((aligned(1024))) unsigned char sm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = (wave >> 2) * 64, wn = (wave & 3) * 32;
    const int l15 = lane & 15, kq = lane >> 4;
    const int x = blockIdx.x & 7, sl = blockIdx.x >> 3;
    const int mb = (x / p.xcd_n) * p.xr_m + sl / p.xr_n, nb = (x % p.xcd_n) * p.xr_n + sl % p.xr_n;
    if (mb >= p.tiles_m || nb >= p.tiles_n) return;
    const int ks = z;
    const int KT = (p.K + BK - 1) / BK;
    const int kt0 = ks * p.k_tiles_per_split, kt1 = min(KT, kt0 + p.k_tiles_per_split);
    const int chunks_m = (p.M + H2_ROWS - 1) / H2_ROWS, chunks_n = (p.N + H2_ROWS - 1) / H2_ROWS;
    const unsigned char* Ap = reinterpret_cast<const unsigned char*>(p.A);
    const unsigned char* Bp = reinterpret_cast<const unsigned char*>(p.B);
    const unsigned char* Ag = Ap + ((long)min(mb, chunks_m - 1) * KT + kt0) * H2_CHUNK + tid * 16;
    const unsigned char* Bg = Bp + ((long)nb * KT + kt0) * H2_CHUNK + tid * 16;
    const unsigned char* Sg = nullptr;
    if (wave == 0) Sg = Ap + (long)chunks_m * KT * H2_CHUNK + ((long)min(mb, chunks_m - 1) * KT + kt0) * H2_SCALES + (lane & 31) * 16;
    else if (wave == 1) Sg = Bp + (long)chunks_n * KT * H2_CHUNK + ((long)nb * KT + kt0) * H2_SCALES + (lane & 31) * 16;
    f32x4v acc[4][2], tmp[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) { acc[i][j][r] = 0.f; tmp[i][j][r] = 0.f; }

    auto stage = [&](int buf) {
        unsigned char* dst = sm + buf * STAGE + wave * 1024;
#pragma unroll
        for (int j = 0; j < PIECES; ++j) {
            const unsigned char* src = (j < PPC ? Ag : Bg) + (j % PPC) * (NT * 16);
            __builtin_amdgcn_global_load_lds((glb_vptr)src, (lds_vptr)(dst + j * (NT * 16)), 16, 0, 0);
        }
        if (wave <= 1 && lane < 32)
            __builtin_amdgcn_global_load_lds((glb_vptr)Sg, (lds_vptr)(sm + buf * STAGE + PLANES + wave * H2_SCALES), 16, 0, 0);
        Ag += H2_CHUNK;
        Bg += H2_CHUNK;
        Sg += H2_SCALES;
    };
    // fragment byte offsets inside the stage image [A chunk | B chunk | A scales | B scales]: lane (row l15 of a 16-row block, k group kq)
    const int so = (kq ^ h2_swz(l15)) * 16;
    const int baseA = (wm + l15) * 64 + so, baseB = H2_CHUNK + (wn + l15) * 64 + so;
    const int scA = PLANES + (wm + 4 * kq) * 4;          // + 16 i * 4: rows 16 i + 4 kq .. + 3
    const int scB = PLANES + H2_SCALES + (wn + l15) * 4;  // + 16 j * 4

    const bool do_load = p.diag != 2, do_math = p.diag != 1;
#pragma unroll
    for (int s0 = 0; s0 < NS - 1; ++s0)
        if (kt0 + s0 < kt1 && do_load) stage(s0);
    int cur = 0;
    for (int kt = kt0; kt < kt1; ++kt) {
        if (NS > 2 && kt + NS - 2 < kt1) {
            if (wave <= 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * (PIECES + 1)) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * PIECES) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        if (kt + NS - 1 < kt1 && do_load) stage(cur == 0 ? NS - 1 : cur - 1);
        if (!do_math) { cur = (cur + 1 == NS) ? 0 : cur + 1; continue; }
        const unsigned char* sb = sm + cur * STAGE;
        f16x8 a[4][2], bb[2][2];
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) {
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i][pl] = *reinterpret_cast<const f16x8*>(sb + baseA + pl * H2_PLANE + i * 16 * 64);
#pragma unroll
            for (int j = 0; j < 2; ++j) bb[j][pl] = *reinterpret_cast<const f16x8*>(sb + baseB + pl * H2_PLANE + j * 16 * 64);
        }
        // small terms first
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) tmp[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i][1], bb[j][0], tmp[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) tmp[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i][0], bb[j][1], tmp[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) tmp[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i][0], bb[j][0], tmp[i][j], 0, 0, 0);
        if (((kt + 1) & (H2_SEG - 1)) == 0 || kt + 1 == kt1) {
            float cb[2];
            float4 ca[4];
            {
                const unsigned sbase = (unsigned)(size_t)(sm) + cur * STAGE;
                const unsigned aB = sbase + scB, aA = sbase + scA;
                asm volatile("ds_read_b32 %0, %1" : "=v"(cb[0]) : "v"(aB));
                asm volatile("ds_read_b32 %0, %1 offset:64" : "=v"(cb[1]) : "v"(aB));
#pragma unroll
                for (int i = 0; i < 4; ++i) asm volatile("ds_read_b128 %0, %1" : "=v"(ca[i]) : "v"(aA + i * 64));
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float cav[4] = {ca[i].x, ca[i].y, ca[i].z, ca[i].w};
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        acc[i][j][r] = fmaf(tmp[i][j][r], cav[r] * cb[j], acc[i][j][r]);
                        tmp[i][j][r] = 0.f;
                    }
            }
        }
        cur = (cur + 1 == NS) ? 0 : cur + 1;
    }
    epilogue16<4, 2>(p, acc, p.C, 0, ks, mb * 128, nb * 128, wm, wn, lane);
}

long h2_bytes(int rows, int cols) {
    return (long)((rows + H2_ROWS - 1) / H2_ROWS) * ((cols + BK - 1) / BK) * (H2_CHUNK + H2_SCALES);
}

// up to 12 operands packed by one launch (one workgroup per 128 x 32 chunk, all jobs in one linear grid)
int h2_pack_multi(const H2PackJob* jobs, int n, hipStream_t st) {
    ECHR_REQUIRE(jobs && n >= 1 && n <= H2_MAX_JOBS, "h2_pack: 1..%d jobs", H2_MAX_JOBS);
    H2PackArgs a;
    int total = 0;
    double bytes = 0.0;
    for (int i = 0; i < n; ++i) {
        const H2PackJob& j = jobs[i];
        ECHR_REQUIRE(j.src && j.dst && j.R > 0 && j.K > 0, "h2_pack: bad job %d", i);
        ECHR_REQUIRE(j.s_row == 1 || j.s_col == 1, "h2_pack: one stride must be 1 (s_row=%ld s_col=%ld)", j.s_row, j.s_col);
        ECHR_REQUIRE((reinterpret_cast<uintptr_t>(j.dst) & 15) == 0, "h2_pack: dst must be 16-byte aligned");
        a.job[i] = j;
        a.start[i] = total;
        total += 2 * ((((j.K + BK - 1) / BK) + H2_SEG - 1) / H2_SEG) * ((j.R + H2_ROWS - 1) / H2_ROWS);
        bytes += 4.0 * j.R * j.K + (double)h2_bytes(j.R, j.K);
    }
    a.start[n] = total;
    for (int i = n; i < H2_MAX_JOBS; ++i) { a.job[i] = a.job[0]; a.start[i + 1] = total; }
    a.njobs = n;
    if (config().diag_skip & 1) return 0;
    ProfScope prof(PROF_PACK, 0.0, bytes, st);
    hipLaunchKernelGGL(h2_pack_kernel, dim3(total), dim3(256), 0, st, a);
    return check_launch("h2_pack");
}

int h2_pack(const float* src, int rows, int cols, long s_row, long s_col, void* dst, hipStream_t st) {
    H2PackJob j{src, static_cast<unsigned char*>(dst), rows, cols, s_row, s_col};
    return h2_pack_multi(&j, 1, st);
}

constexpr int N128_LD = BK + 4;                         // floats per LDS row of a k-contiguous operand: 16-byte aligned, conflict-free b128 fragment reads (as rec_gemm)
// ------------------------------------------------------------------------------------------------------
// Exact-fp32 128 x 128 tile for EVERY layout (round 6; round 3's NT-only gemm_f32_nt128_kernel generalised and replaced): operands whose
// contiguous axis is k (NT) or the row / column axis (NN data gradients: B = W [K][N]; TN weight gradients: A = dY^T [K][M], B = X [K][N]),
// aligned or not, ragged in every direction.  4 waves of 64 x 64 (2 x 2 v_mfma_f32_32x32x2_f32 tiles), BK = 32, two LDS stages, ONE barrier per k block,
// two workgroups per CU (a wave of the other workgroup issues MFMAs while this one stores its stage and waits at the barrier).
//   k-contiguous operand (KC):  LDS [128 rows][32 + 4]; a lane's fragment of four consecutive MFMAs is one ds_read_b128
//   row-contiguous operand:     LDS [32 k][128 + 4];    filled by float4 loads along the rows, fragments are ds_read_b32 of 32 consecutive
//                               floats per half wave (conflict-free); the k pairing of the MFMAs is the same as for KC, so any mix works
// The 64 x 64 tile of gemm_f32_kernel stages 32 KB of operand per 0.26 MFLOP and has ONE MFMA per fragment pair; this tile halves the bytes
// per flop and issues four MFMAs per pair.  Same epilogue (biases, addend, tanh, row remap / scatter, split-K atomics), same grouped form.
// ------------------------------------------------------------------------------------------------------
constexpr int T128_OP = 128 * N128_LD;                  // floats per operand and stage (the row-contiguous form needs 32 * 132 = 4224 <= 4608)
constexpr int T128_STAGE = 2 * T128_OP;
constexpr int T128_LDM = 128 + 4;
template <bool KC, int NT = 256>
__device__ __forceinline__ void t128_fetch(float4 (&r)[1024 / NT], const float* __restrict__ P, long s_mn, long s_k, int mn0, int k0, int MN, int K,
                                           int kend, bool vec, int tid) {
#pragma unroll
    for (int i = 0; i < 1024 / NT; ++i) {
        const int f = tid + NT * i;
        if (KC) {
            const int row = f >> 3, k = k0 + 4 * (f & 7);
            const float* src = P + (long)min(mn0 + row, MN - 1) * s_mn;          // rows beyond MN are clamped: their products land in rows / columns the epilogue does not store
            float4 v;
            if (vec) {                                                       // (K % 4 == 0 here: a float4 never straddles kend)
                v = *reinterpret_cast<const float4*>(src + min(k, K - 4));
                if (k >= kend) v = make_float4(0.f, 0.f, 0.f, 0.f);
            } else {
                v.x = k < kend ? src[k] : 0.f;
                v.y = k + 1 < kend ? src[k + 1] : 0.f;
                v.z = k + 2 < kend ? src[k + 2] : 0.f;
                v.w = k + 3 < kend ? src[k + 3] : 0.f;
            }
            r[i] = v;
        } else {
            const int k = k0 + (f >> 5), mn = mn0 + 4 * (f & 31);
            const float* src = P + (long)min(k, K - 1) * s_k;
            float4 v;
            if (vec) {                                                       // (MN % 4 == 0 here)
                v = *reinterpret_cast<const float4*>(src + min(mn, MN - 4));
            } else {
                v.x = src[min(mn, MN - 1)]; v.y = src[min(mn + 1, MN - 1)]; v.z = src[min(mn + 2, MN - 1)]; v.w = src[min(mn + 3, MN - 1)];
            }
            if (k >= kend) v = make_float4(0.f, 0.f, 0.f, 0.f);
            r[i] = v;
        }
    }
}
template <bool KC, int NT = 256>
__device__ __forceinline__ void t128_stash(const float4 (&r)[1024 / NT], float* __restrict__ S, int tid) {
#pragma unroll
    for (int i = 0; i < 1024 / NT; ++i) {
        const int f = tid + NT * i;
        if (KC) *reinterpret_cast<float4*>(S + (f >> 3) * N128_LD + 4 * (f & 7)) = r[i];
        else    *reinterpret_cast<float4*>(S + (f >> 5) * T128_LDM + 4 * (f & 31)) = r[i];
    }
}
// fragment of chunk c (k = 8c .. 8c + 7) for the 32-row block at `row`: element j feeds MFMA j, which contracts k = 8c + j (lanes 0-31) and 8c + 4 + j (lanes 32-63)
template <bool KC>
__device__ __forceinline__ float4 t128_frag(const float* __restrict__ S, int row, int c, int kh) {
    if (KC) return *reinterpret_cast<const float4*>(S + row * N128_LD + 8 * c + 4 * kh);
    const float* q = S + (8 * c + 4 * kh) * T128_LDM + row;
    return make_float4(q[0], q[T128_LDM], q[2 * T128_LDM], q[3 * T128_LDM]);
}
// W8: eight waves of 64 x 32 instead of four of 64 x 64 -- for launches of at most one workgroup per CU (the 240-tile logits product), where
// the four-wave form leaves every SIMD with ONE wave and nothing to issue while that wave fetches, stores its stage or waits at the barrier
template <bool AKC, bool BKC, bool W8 = false>
__global__ __launch_bounds__(W8 ? 512 : 256, W8 ? 1 : 2) void gemm_f32_t128_kernel(GemmParams pin) {
    constexpr int NT = W8 ? 512 : 256, TN = W8 ? 1 : 2;
    extern __shared__ __attribute__((aligned(16))) float nsm[];
    int z = blockIdx.z;
    const GemmParams p = select_group(pin, z);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = W8 ? (wave >> 2) * 64 : (wave >> 1) * 64, wn = W8 ? (wave & 3) * 32 : (wave & 1) * 64;
    const int nwg = p.tiles_m * p.tiles_n;
    int bid = blockIdx.x;
    {   // XCD-aware tile order (see gemm_f32_kernel)
        int q = nwg / 8, r = nwg % 8, x = bid % 8;
        bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + bid / 8;
    }
    const int m0 = (bid / p.tiles_n) * 128, n0 = (bid % p.tiles_n) * 128;
    if (m0 >= p.M || n0 >= p.N) return;          // grouped problems narrower than the widest of the launch
    const int b = z / p.split_k, ks = z % p.split_k;
    const float* A = p.A + (long)b * p.bsa;
    const float* B = p.B + (long)b * p.bsb;
    float* C = p.C + (long)b * p.bsc;
    const int kt_total = (p.K + BK - 1) / BK;
    const int kt0 = ks * p.k_tiles_per_split, kt1 = min(kt_total, kt0 + p.k_tiles_per_split);
    const int kend = min(p.K, kt1 * BK);
    f32x16 acc[2][TN];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    float4 ra[1024 / NT], rb[1024 / NT];
    const long sa_mn = AKC ? p.sam : 1, sa_k = AKC ? 1 : p.sak, sb_mn = BKC ? p.sbn : 1, sb_k = BKC ? 1 : p.sbk;
    auto fetch = [&](int kt) {
        t128_fetch<AKC, NT>(ra, A, sa_mn, sa_k, m0, kt * BK, p.M, p.K, kend, p.vecA, tid);
        t128_fetch<BKC, NT>(rb, B, sb_mn, sb_k, n0, kt * BK, p.N, p.K, kend, p.vecB, tid);
    };
    auto stash = [&](int stage) {
        t128_stash<AKC, NT>(ra, nsm + stage * T128_STAGE, tid);
        t128_stash<BKC, NT>(rb, nsm + stage * T128_STAGE + T128_OP, tid);
    };
    const int l31 = lane & 31, kh = lane >> 5;
    if (kt0 < kt1) { fetch(kt0); stash(0); }
    __syncthreads();
    for (int kt = kt0; kt < kt1; ++kt) {
        const int st = (kt - kt0) & 1;
        if (kt + 1 < kt1) fetch(kt + 1);                         // in flight during the MFMAs below
        const float* As = nsm + st * T128_STAGE;
        const float* Bs = As + T128_OP;
#pragma unroll
        for (int c = 0; c < BK / 8; ++c) {
            float4 a4[2], b4[TN];
#pragma unroll
            for (int i = 0; i < 2; ++i) a4[i] = t128_frag<AKC>(As, wm + i * 32 + l31, c, kh);
#pragma unroll
            for (int j = 0; j < TN; ++j) b4[j] = t128_frag<BKC>(Bs, wn + j * 32 + l31, c, kh);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[i].x, b4[j].x, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[i].y, b4[j].y, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[i].z, b4[j].z, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[i].w, b4[j].w, acc[i][j], 0, 0, 0);
                }
        }
        if (kt + 1 < kt1) stash(st ^ 1);                         // the other stage was last read before the previous barrier
        __syncthreads();
    }
    epilogue<2, TN>(p, acc, C, b, ks, m0, n0, wm, wn, lane);
}

template <int BM, int BN, int WM, int WN>
static void launch_cfg(const GemmParams& p, bool akc, bool bkc, dim3 grid, hipStream_t st) {
    constexpr int NT = (BM / WM) * (BN / WN) * 64;
    if (akc && bkc) hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, WM, WN, true, true>), grid, dim3(NT), 0, st, p);
    else if (akc && !bkc) hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, WM, WN, true, false>), grid, dim3(NT), 0, st, p);
    else if (!akc && bkc) hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, WM, WN, false, true>), grid, dim3(NT), 0, st, p);
    else hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, WM, WN, false, false>), grid, dim3(NT), 0, st, p);
}

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

static int gemm_impl(const echr_gemm_desc* ds, int ng, hipStream_t st);
int gemm(const echr_gemm_desc& d, hipStream_t st) { return gemm_impl(&d, 1, st); }

// C[M, Nc <= 16] = A[M, K] . W[Nc, K]^T + bias for very tall A (the event encoder's fc2 over N*N pairs: 1 M rows x 512 -> 16): a pure
// stream over A.  One wave per 16-row tile, v_mfma_f32_16x16x4_f32 (exact fp32); W lives in registers for the whole launch (K / 16 x 4
// floats per lane); a lane's A fragment of four consecutive MFMAs is ONE float4 (lane (r, kk) holds A[r][16 s + 4 kk + j], j = MFMA
// index: a fixed permutation of k that W's registers follow), so a row's 64-byte pieces are read once, in order.
template <int KS>          // K / 16
__global__ __launch_bounds__(256) void skinny_nt_kernel(const float* __restrict__ A, long lda, const float* __restrict__ W, long ldw,
                                                        const float* __restrict__ bias, float* __restrict__ C, long ldc, int M, int Nc, int tiles) {
    typedef float f32x4s __attribute__((ext_vector_type(4)));
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, r = lane & 15, kk = lane >> 4;
    float4 bw[KS];
#pragma unroll
    for (int s_ = 0; s_ < KS; ++s_)
        bw[s_] = r < Nc ? *reinterpret_cast<const float4*>(W + (long)r * ldw + 16 * s_ + 4 * kk) : make_float4(0.f, 0.f, 0.f, 0.f);
    const float bv = (bias && r < Nc) ? bias[r] : 0.f;
    for (int tile = blockIdx.x * 4 + w; tile < tiles; tile += gridDim.x * 4) {
        const long r0 = (long)tile * 16;
        const float* ap = A + (r0 + r < M ? r0 + r : (long)M - 1) * lda + 4 * kk;
        float4 av[KS];
#pragma unroll
        for (int s_ = 0; s_ < KS; ++s_) av[s_] = *reinterpret_cast<const float4*>(ap + 16 * s_);
        f32x4s acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s_ = 0; s_ < KS; ++s_) {
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s_].x, bw[s_].x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s_].y, bw[s_].y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s_].z, bw[s_].z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s_].w, bw[s_].w, acc, 0, 0, 0);
        }
        if (r < Nc) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const long row = r0 + 4 * kk + i;
                if (row < M) C[row * ldc + r] = acc[i] + bv;
            }
        }
    }
}

bool gemm_skinny_ok(int M, int Nc, int K, long lda, long ldw, const float* A, const float* W) {
    return config().gemm_skinny && M >= 4096 && Nc >= 1 && Nc <= 16 && (K == 512 || K == 256) && lda % 4 == 0 && ldw % 4 == 0 && aligned16(A) && aligned16(W);
}
int gemm_skinny_nt(const float* A, long lda, const float* W, long ldw, const float* bias, float* C, long ldc, int M, int Nc, int K, hipStream_t st) {
    ECHR_REQUIRE(gemm_skinny_ok(M, Nc, K, lda, ldw, A, W), "gemm_skinny_nt: unsupported shape");
    const int tiles = (M + 15) / 16;
    const int grid = tiles / 4 < 2048 ? (tiles + 3) / 4 : 2048;
    ProfScope prof(PROF_GEMM, 2.0 * M * Nc * K, 4.0 * ((double)M * K + (double)Nc * K + (double)M * Nc), st);
    if (K == 512) hipLaunchKernelGGL(skinny_nt_kernel<32>, dim3(grid), dim3(256), 0, st, A, lda, W, ldw, bias, C, ldc, M, Nc, tiles);
    else hipLaunchKernelGGL(skinny_nt_kernel<16>, dim3(grid), dim3(256), 0, st, A, lda, W, ldw, bias, C, ldc, M, Nc, tiles);
    return check_launch("skinny_nt");
}

// Up to 4 problems of identical shape/layout/epilogue mode in ONE launch (the three streams' W_ih / W_hh products): fills the
// chip better than three 640-workgroup grids and pays one launch ramp.  Problems may share C when they accumulate (beta = 1).
int gemm_grouped(const echr_gemm_desc* ds, int ng, hipStream_t st) {
    ECHR_REQUIRE(ds && ng >= 1 && ng <= GEMM_MAXG, "gemm_grouped: 1..%d problems", GEMM_MAXG);
    for (int i = 1; i < ng; ++i) {
        const echr_gemm_desc &a = ds[0], &b = ds[i];
        ECHR_REQUIRE(((a.M == b.M && a.N == b.N) || (a.algo == ECHR_GEMM_H2 && b.algo == ECHR_GEMM_H2)) && a.K == b.K && a.sam == b.sam && a.sak == b.sak && (a.sbk == 1) == (b.sbk == 1) &&
                     (a.sbn == 1) == (b.sbn == 1) && a.batch == 1 && b.batch == 1 && a.alpha == b.alpha && a.beta == b.beta &&
                     a.act == b.act && a.split_k == b.split_k && a.algo == b.algo && a.rowmap_mod == b.rowmap_mod &&
                     a.add_mod == b.add_mod && a.ld_add == b.ld_add && a.act == ECHR_ACT_NONE,
                     "gemm_grouped: problems must share shape, layout and epilogue mode");
    }
    if (det_mode() && ng > 1) {
        // fixed-order mode: problems that share an output cannot add atomically -- one launch per problem, in group order, each a single k loop
        // per tile that updates C in place (beta = 1, accumulate mode)
        bool shared_c = false;
        for (int gi = 1; gi < ng; ++gi) for (int gj = 0; gj < gi; ++gj) shared_c = shared_c || ds[gi].C == ds[gj].C;
        if (shared_c) {
            ECHR_REQUIRE(ds[0].split_k < 0 && ds[0].beta == 1.f, "gemm_grouped: problems sharing C need auto split-K in accumulate mode");
            for (int gi = 0; gi < ng; ++gi)
                if (int rc = gemm_impl(&ds[gi], 1, st)) return rc;
            return 0;
        }
    }
    return gemm_impl(ds, ng, st);
}

static int gemm_impl(const echr_gemm_desc* ds, int ng, hipStream_t st) {
    const echr_gemm_desc& d = ds[0];
    ECHR_REQUIRE(d.A && d.B && d.C, "gemm: null operand");
    ECHR_REQUIRE(d.M > 0 && d.N > 0 && d.K > 0 && d.batch >= 1, "gemm: bad shape M=%d N=%d K=%d batch=%d", d.M, d.N, d.K, d.batch);
    ECHR_REQUIRE(d.algo == ECHR_GEMM_H2 || d.sam == 1 || d.sak == 1, "gemm: A needs a unit stride (sam=%ld sak=%ld)", (long)d.sam, (long)d.sak);
    ECHR_REQUIRE(d.algo == ECHR_GEMM_H2 || d.sbk == 1 || d.sbn == 1, "gemm: B needs a unit stride (sbk=%ld sbn=%ld)", (long)d.sbk, (long)d.sbn);
    ECHR_REQUIRE(d.act == ECHR_ACT_NONE || d.split_k <= 1, "gemm: split-K cannot carry an activation");
    ECHR_REQUIRE(d.act != ECHR_ACT_MUL_DTANH || d.aux, "gemm: MUL_DTANH needs aux");
    ECHR_REQUIRE(!d.addend || d.add_mod > 0, "gemm: addend needs add_mod");
    GemmParams p;
    p.A = d.A; p.B = d.B; p.C = d.C; p.M = d.M; p.N = d.N; p.K = d.K;
    p.sam = d.sam; p.sak = d.sak; p.sbk = d.sbk; p.sbn = d.sbn; p.ldc = d.ldc;
    p.bsa = d.bsa; p.bsb = d.bsb; p.bsc = d.bsc;
    p.alpha = d.alpha; p.beta = d.beta;
    p.bias = d.bias; p.bs_bias = d.bs_bias; p.bias2 = d.bias2;
    p.addend = d.addend; p.add_mod = d.add_mod; p.ld_add = d.ld_add;
    p.act = d.act; p.aux = d.aux; p.ld_aux = d.ld_aux;
    p.rowmap_mod = d.rowmap_mod; p.rowmap_mul = d.rowmap_mul;
    p.rowidx = d.row_index; p.rowidx_max = d.row_index_max;
    ECHR_REQUIRE(!d.row_index || (d.beta == 1.f && d.act == ECHR_ACT_NONE && d.rowmap_mod == 0 && d.row_index_max >= 0),
                 "gemm: row_index scatters with atomic adds: needs beta = 1 (accumulate), no activation, no row remap");
    p.ngroup = ng;
    p.diag = (config().diag_skip >> 6) & 3;
    int maxN = d.N, maxM = d.M;
    for (int i = 0; i < GEMM_MAXG; ++i) {
        const echr_gemm_desc& g = ds[i < ng ? i : 0];
        maxN = g.N > maxN ? g.N : maxN;
        maxM = g.M > maxM ? g.M : maxM;
        p.gM[i] = g.M;
        p.gA[i] = g.A; p.gB[i] = g.B; p.gC[i] = g.C; p.gsbk[i] = g.sbk; p.gsbn[i] = g.sbn; p.gldc[i] = g.ldc; p.gN[i] = g.N;
        p.gbias[i] = g.bias; p.gbias2[i] = g.bias2; p.gaddend[i] = g.addend;
    }
    const bool h2 = d.algo == ECHR_GEMM_H2;
    ECHR_REQUIRE(!h2 || d.batch == 1, "gemm: h2 operands take batch == 1");
    const bool akc = (d.sak == 1);
    const bool bkc = (d.sbk == 1);
    const bool use_split = !h2 && d.algo == ECHR_GEMM_BF16X3 && config().gemm_bf16x3 && akc && bkc && d.K % 4 == 0 && d.K >= 4 && d.sam % 4 == 0 &&
                           d.sbn % 4 == 0 && aligned16(d.A) && aligned16(d.B) && d.bsa % 4 == 0 && d.bsb % 4 == 0 &&
                           (long)d.M * d.N >= 128L * 128L && ng == 1;
    p.vecA = akc ? (d.sam % 4 == 0 && aligned16(d.A) && d.bsa % 4 == 0) : (d.sak % 4 == 0 && aligned16(d.A) && d.bsa % 4 == 0);
    p.vecB = bkc ? (d.sbn % 4 == 0 && aligned16(d.B) && d.bsb % 4 == 0) : (d.sbk % 4 == 0 && aligned16(d.B) && d.bsb % 4 == 0);
    for (int i = 1; i < ng; ++i) {
        p.vecA = p.vecA && aligned16(ds[i].A);
        p.vecB = p.vecB && aligned16(ds[i].B) && (bkc ? ds[i].sbn % 4 == 0 : ds[i].sbk % 4 == 0);
    }

    // tile choice.  Measured on the c3 shapes (tools/gemm_bench.py): the 64x64 tile (7 waves/SIMD resident, latency hidden by
    // occupancy) matches or beats 128x128 (2 waves/SIMD) everywhere at these sizes, including the 1280 x 5001 x 1536 logit
    // products (tile quantisation: 400 big tiles on 256 CUs), so it is the default; 128x128 stays selectable for tuning.
    int BMs = 64, BNs = 64;
    bool w8 = false;
    // tuning knobs are read from the environment ONCE per process (tools/gemm_bench.py sets them before loading the library); a stray
    // variable can therefore not change numerics or split order call by call
    static const int env_h2_bm = getenv("ECHR_H2_BM") ? atoi(getenv("ECHR_H2_BM")) : 128;
    const char tile_code = (char)config().gemm_tile;          // 0 = heuristics; set from ECHR_GEMM_TILE at load or by echr_config_set
    const int env_split = config().gemm_split;
    if (h2) { BMs = env_h2_bm; BNs = 128; }
    if (use_split) { BMs = 128; BNs = ((long)((d.M + 127) / 128) * ((d.N + 127) / 128) * d.batch >= 96) ? 128 : 64; }
    // large exact-fp32 products: the 128 x 128 double-buffered tile (gemm_f32_t128_kernel) once its tiles fill most of the chip -- measured stand-alone
    // against the 64 x 64 tile (tools/t128_bench.py, us): logits 762 x 5001 x 1536 (240 tiles) 133 vs 155, d W_logit 5001 x 1536 x 764 (480) 118 vs
    // 134, 4096^3 NT 1083 vs 1286; with fewer tiles the 64 x 64 tile's finer grain (7 resident waves per SIMD, 2.5 x more tiles to balance) wins
    // whatever the k split: d OUTD 762 x 1536 x 5004 (72 tiles) 179 vs 143, gin 1280 x 2048 x 512 (160) 48 vs 38, P_all 8192 x 512 x 500 (256) 58 vs 50
    static const int t128_on = getenv("ECHR_GEMM_T128") ? atoi(getenv("ECHR_GEMM_T128")) : 1;
    static const int t128_min_tiles = getenv("ECHR_GEMM_T128_TILES") ? atoi(getenv("ECHR_GEMM_T128_TILES")) : 200;
    static const int t128_min_k = getenv("ECHR_GEMM_T128_K") ? atoi(getenv("ECHR_GEMM_T128_K")) : 1024;
    const long tiles128 = (long)((maxM + 127) / 128) * ((maxN + 127) / 128) * d.batch * ng;
    // ... and K-heavy NT products of 32-128 such tiles (d OUTD 762 x 1536 x 5004: 72 tiles): the eight-wave form on floor(256 / tiles) k slices,
    // one workgroup per CU -- 132 us against 146 on the 64 x 64 tile's own split (tools/t128_w8.py)
    static const int t128_ks_on = getenv("ECHR_GEMM_T128_KS") ? atoi(getenv("ECHR_GEMM_T128_KS")) : 1;
    const bool t128ks = t128_on && t128_ks_on && !h2 && !use_split && !tile_code && akc && bkc && ng == 1 && d.batch == 1 && d.split_k < 0 && d.rowmap_mod == 0 &&
                        d.act == ECHR_ACT_NONE && (d.beta == 0.f || d.beta == 1.f) && tiles128 >= 32 && tiles128 <= 128 && d.K >= 2048 && !deterministic_gemm();
    const bool t128 = !h2 && !use_split && d.K >= 64 && d.rowmap_mod == 0 &&
                      ((tile_code == 't') || t128ks || (t128_on && !tile_code && tiles128 >= t128_min_tiles && d.K >= t128_min_k && (akc || bkc)));
    if (t128) { BMs = 128; BNs = 128; }
    if (tile_code && tile_code != 't') {          // tuning override (tools/gemm_bench.py); never set in production
        const char e0 = tile_code;
        if (e0 == '1') { BMs = 128; BNs = 128; } else if (e0 == '6') { BMs = 64; BNs = 64; }
        else if (e0 == 'a') { BMs = 128; BNs = 64; } else if (e0 == 'b') { BMs = 64; BNs = 128; }
        else if (e0 == 'c') { BMs = 128; BNs = 128; w8 = true; }
        if (use_split) { BMs = 128; BNs = (e0 == 's') ? 64 : 128; }
    }
    p.tiles_m = (maxM + BMs - 1) / BMs;
    p.tiles_n = (maxN + BNs - 1) / BNs;          // grouped h2 problems of different widths: sized for the widest
    const int kt_total = (d.K + BK - 1) / BK;
    int split = d.split_k;
    const bool accumulate = (d.split_k > 1 || d.split_k < 0);   // caller promises C already holds its base value
    if (split < 0) {  // auto: fill ~2 waves of workgroups over the chip when the output grid is small
        long wgs = (long)p.tiles_m * p.tiles_n * d.batch * ng;
        split = 1;
        // latency-bound regime: fewer than 2 workgroups per CU.  Split K so that ~1024 workgroups overlap each other's
        // load latency, keeping at least 4 k-tiles (128 deep) per split (measured optimum on the weight-gradient shapes).
        if (h2) {
            // two 8-wave workgroups per CU = 512 slots.  Cost model fitted to tools/h2_bench.py / h2_ksweep.py on the c3 shapes (us): a k
            // block costs 0.8 per workgroup while every CU holds at most one, 1.4 per pair once CUs hold two; a launch costs 9 for
            // prologue + epilogue; a k-split adds its atomic epilogue (the output written `split` times at ~2.5 TB/s) and keeps at
            // least 8 k blocks per slice.  (Round 1 split whenever the grid had fewer than 200 tiles: 192 tiles x 3 slices ran 1.35x
            // slower than the unsplit product.)
            if (d.act == ECHR_ACT_NONE && (d.beta == 0.f || d.beta == 1.f) && d.rowmap_mod == 0) {
                const double out_mb = 4e-6 * (double)d.M * d.N * ng;
                double best = 1e30;
                for (int sp = 1; sp <= 32 && (sp == 1 || kt_total / sp >= 8); ++sp) {
                    const long W = wgs * sp;
                    const int kb = (kt_total + sp - 1) / sp;
                    double t;
                    if (W <= 256) t = 0.8 * kb;
                    else if (W <= 512) t = 1.4 * kb;
                    else t = 1.4 * kb * (double)(W / 512) + ((W % 512) > 256 ? 1.4 : ((W % 512) ? 0.8 : 0.0)) * kb;
                    t += 9.0 + (sp > 1 ? out_mb * sp / 2.5 + (d.beta == 0.f ? 3.0 : 0.0) : 0.0);
                    if (t < best) { best = t; split = sp; }
                }
            }
        } else if (d.act == ECHR_ACT_NONE && wgs < (use_split ? 200 : (t128 ? 0 : 512)) && (d.beta == 0.f || d.beta == 1.f) && d.rowmap_mod == 0) {
            split = (int)min((long)kt_total, max(1L, ((use_split ? 400 : 1024) + wgs - 1) / max(wgs, 1L)));
            if (split > 1 && kt_total / split < 4) split = max(1, kt_total / 4);
        }
    }
    if (t128ks) split = (int)(256 / tiles128);
    if (env_split > 0 && d.split_k < 0) split = env_split;
    // reproducible mode (greedy sampler: index outputs must be bit-exact run to run): no k-slices that add with fp32 atomics
    if (deterministic_gemm() && d.split_k < 0) split = 1;
    // grouped problems that share an output must add atomically even when K would not be split: force two k-slices
    bool shared_c = false;
    for (int gi = 1; gi < ng; ++gi) for (int gj = 0; gj < gi; ++gj) shared_c = shared_c || ds[gi].C == ds[gj].C;
    ECHR_REQUIRE(!shared_c || (d.split_k < 0 && kt_total >= 2), "gemm_grouped: problems sharing C need auto split-K (accumulate mode) and K > 32");
    if (shared_c && split < 2) split = 2;
    if (split < 1) split = 1;
    if (split > kt_total) split = kt_total;
    p.k_tiles_per_split = (kt_total + split - 1) / split;
    split = (kt_total + p.k_tiles_per_split - 1) / p.k_tiles_per_split;
    p.split_k = split;
    if (d.split_k < 0) {
        // auto mode: beta == 0 -> the library zero-fills C itself before splitting; beta == 1 -> accumulate
        ECHR_REQUIRE(d.beta == 0.f || d.beta == 1.f, "gemm: auto split needs beta in {0,1}");
        if (split > 1 && d.beta == 0.f) {
            ECHR_REQUIRE(d.rowmap_mod == 0, "gemm: auto split cannot zero a row-remapped output");
            for (int gi = 0; gi < ng; ++gi) {
                bool seen = false;
                for (int gj = 0; gj < gi; ++gj) seen = seen || ds[gj].C == ds[gi].C;
                if (seen) continue;
                for (int bb = 0; bb < d.batch; ++bb) {
                    int rc = fill_zero_2d(ds[gi].C + (long)bb * d.bsc, ds[gi].M, ds[gi].N, ds[gi].ldc, st);
                    if (rc) return rc;
                }
            }
        }
        if (split == 1) p.beta = d.beta;
    } else if (accumulate && split == 1) p.beta = 1.f;
    dim3 grid(p.tiles_m * p.tiles_n, 1, d.batch * split * ng);
    if (config().diag_skip & (h2 ? 32 : 16)) return 0;          // diagnostic (tools/skip_bounds.py): the product is not launched, results are wrong
    static const bool log_on = getenv("ECHR_GEMM_LOG") != nullptr;
    if (log_on) fprintf(stderr, "[gemm] M=%d N=%d K=%d batch=%d %s%s tile=%dx%d split=%d algo=%s wgs=%d\n", d.M, d.N, d.K, d.batch, akc ? "N" : "T",
                        bkc ? "T" : "N", BMs, BNs, split, h2 ? "h2" : use_split ? "bf16x3" : t128 ? "f32-t128" : "f32", (int)(grid.x * grid.z));
    // algorithmic work of this launch: 2MNK flops; one read of A and B, one write of C
    ProfScope prof(h2 ? PROF_GEMM_H2 : use_split ? PROF_GEMM_SPLIT : PROF_GEMM, 2.0 * d.M * d.N * d.K * d.batch * ng, 4.0 * ((double)d.M * d.K + (double)d.K * d.N + (double)d.M * d.N) * d.batch * ng, st);
    if (h2) {
        // XCD rectangles: xm x xn = 8, least padded tiles first, then fewest panels per XCD
        int best = -1; long best_cost = 0;
        for (int xm = 1; xm <= 8; xm *= 2) {
            const int xn = 8 / xm, rm = (p.tiles_m + xm - 1) / xm, rn = (p.tiles_n + xn - 1) / xn;
            const long cost = ((long)rm * rn * 8) * 100000L + (long)rm * BMs + (long)rn * BNs;
            if (best < 0 || cost < best_cost) { best = xm; best_cost = cost; }
        }
        p.xcd_n = 8 / best; p.xr_m = (p.tiles_m + best - 1) / best; p.xr_n = (p.tiles_n + p.xcd_n - 1) / p.xcd_n;
        grid.x = 8 * p.xr_m * p.xr_n;
        static const int wn_sel = getenv("ECHR_H2_WN") ? atoi(getenv("ECHR_H2_WN")) : 32;
        // measured (tools/h2_bench.py): 2 stages x 2 workgroups per CU beats 3-4 stages x 1 workgroup per CU on every c3 shape (270 vs 245
        // TF/s at 4096^3): the second resident workgroup hides more latency than a deeper ring does; the deeper rings stay selectable
        static const int ns_sel = getenv("ECHR_H2_STAGES") ? atoi(getenv("ECHR_H2_STAGES")) : 2;
        constexpr int ST128 = 2 * H2_CHUNK + 2 * H2_SCALES, ST256 = 3 * H2_CHUNK + 3 * H2_SCALES;
        static bool attr_done = false;
        if (!attr_done) {          // LDS beyond 64 KB needs the opt-in attribute, once per kernel
            attr_done = true;
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_h2_kernel<128, 32, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * ST128);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_h2_kernel<128, 32, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * ST128);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_h2_kernel<128, 64, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * ST128);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_h2_kernel<128, 32, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * ST128);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_h2_kernel<128, 64, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * ST128);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_h2_kernel<256, 64, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * ST256);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_h2_kernel<256, 64, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * ST256);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_h2m16_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * ST128);
        }
        // the 16x16x32 form is faster wherever the epilogue is not a many-way atomic split (its stores are 64-byte row pieces)
        static const int m16_sel = getenv("ECHR_H2_M16") ? atoi(getenv("ECHR_H2_M16")) : 1;
        if (m16_sel && BMs == 128 && split < 8) hipLaunchKernelGGL((gemm_h2m16_kernel<2>), grid, dim3(512), 2 * ST128, st, p);
        else if (BMs == 256 && ns_sel >= 3) hipLaunchKernelGGL((gemm_h2_kernel<256, 64, 3>), grid, dim3(512), 3 * ST256, st, p);
        else if (BMs == 256) hipLaunchKernelGGL((gemm_h2_kernel<256, 64, 2>), grid, dim3(512), 2 * ST256, st, p);
        else if (wn_sel == 64 && ns_sel >= 4) hipLaunchKernelGGL((gemm_h2_kernel<128, 64, 4>), grid, dim3(256), 4 * ST128, st, p);
        else if (wn_sel == 64) hipLaunchKernelGGL((gemm_h2_kernel<128, 64, 2>), grid, dim3(256), 2 * ST128, st, p);
        else if (ns_sel >= 4) hipLaunchKernelGGL((gemm_h2_kernel<128, 32, 4>), grid, dim3(512), 4 * ST128, st, p);
        else if (ns_sel == 3) hipLaunchKernelGGL((gemm_h2_kernel<128, 32, 3>), grid, dim3(512), 3 * ST128, st, p);
        else hipLaunchKernelGGL((gemm_h2_kernel<128, 32, 2>), grid, dim3(512), 2 * ST128, st, p);
    }
    else if (t128) {
        static bool attr_t = false;
        if (!attr_t) {
            attr_t = true;
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f32_t128_kernel<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * T128_STAGE * (int)sizeof(float));
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f32_t128_kernel<true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * T128_STAGE * (int)sizeof(float));
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f32_t128_kernel<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * T128_STAGE * (int)sizeof(float));
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f32_t128_kernel<false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * T128_STAGE * (int)sizeof(float));
        }
        // float4 staging needs the vector axis to be a multiple of 4 for EVERY problem of the launch (a float4 never straddles the edge)
        bool va = p.vecA && (akc ? d.K % 4 == 0 : true), vb = p.vecB && (bkc ? d.K % 4 == 0 : true);
        for (int gi = 0; gi < ng; ++gi) { if (!akc) va = va && ds[gi].M % 4 == 0; if (!bkc) vb = vb && ds[gi].N % 4 == 0; }
        p.vecA = va; p.vecB = vb;
        const size_t lds = 2 * T128_STAGE * sizeof(float);
        static const int w8_sel = getenv("ECHR_T128_W8") ? atoi(getenv("ECHR_T128_W8")) : 1;
        static const long w8_max = getenv("ECHR_T128_W8_MAX") ? atol(getenv("ECHR_T128_W8_MAX")) : 256;
        if (w8_sel && akc && bkc && (long)grid.x * grid.z <= w8_max) {          // at most one workgroup per CU: eight waves per workgroup
            static bool attr_w8 = false;
            if (!attr_w8) { attr_w8 = true; (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f32_t128_kernel<true, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); }
            hipLaunchKernelGGL((gemm_f32_t128_kernel<true, true, true>), grid, dim3(512), lds, st, p);
        }
        else if (akc && bkc) hipLaunchKernelGGL((gemm_f32_t128_kernel<true, true>), grid, dim3(256), lds, st, p);
        else if (akc) hipLaunchKernelGGL((gemm_f32_t128_kernel<true, false>), grid, dim3(256), lds, st, p);
        else if (bkc) hipLaunchKernelGGL((gemm_f32_t128_kernel<false, true>), grid, dim3(256), lds, st, p);
        else hipLaunchKernelGGL((gemm_f32_t128_kernel<false, false>), grid, dim3(256), lds, st, p);
    }
    else if (use_split && BNs == 128) hipLaunchKernelGGL(gemm_split_kernel<128>, grid, dim3(512), 0, st, p);
    else if (use_split) hipLaunchKernelGGL(gemm_split_kernel<64>, grid, dim3(256), 0, st, p);
    else if (BMs == 128 && BNs == 128 && w8) launch_cfg<128, 128, 64, 32>(p, akc, bkc, grid, st);
    else if (BMs == 128 && BNs == 128) launch_cfg<128, 128, 64, 64>(p, akc, bkc, grid, st);
    else if (BMs == 128) launch_cfg<128, 64, 64, 32>(p, akc, bkc, grid, st);
    else if (BNs == 128) launch_cfg<64, 128, 32, 64>(p, akc, bkc, grid, st);
    else launch_cfg<64, 64, 32, 32>(p, akc, bkc, grid, st);
    return check_launch("gemm_f32");
}

}  // namespace echr

extern "C" int64_t echr_h2_bytes(int32_t rows, int32_t cols) { return echr::h2_bytes(rows, cols); }

extern "C" int echr_h2_pack(const float* src, int32_t rows, int32_t cols, int64_t s_row, int64_t s_col, void* dst, void* stream) {
    ECHR_REQUIRE(src && dst && rows > 0 && cols > 0, "h2_pack: bad arguments");
    return echr::h2_pack(src, rows, cols, (long)s_row, (long)s_col, dst, static_cast<hipStream_t>(stream));
}

extern "C" int echr_gemm_f32(const echr_gemm_desc* d, void* stream) {
    if (!d) { echr::set_error("echr_gemm_f32: null descriptor"); return -22; }
    return echr::gemm(*d, static_cast<hipStream_t>(stream));
}
