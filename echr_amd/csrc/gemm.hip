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
#include "echr_common.h"
#include "echr_internal.h"

namespace echr {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BK = 32;

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
    int split_k, k_tiles_per_split;
    int tiles_m, tiles_n;
    int vecA, vecB;
    // grouped launch: up to 4 same-shaped problems in one grid (blockIdx.z = group * split_k + k-slice); per-group operands
    int ngroup;
    const float* gA[4]; const float* gB[4]; float* gC[4];
    long gsbk[4], gsbn[4], gldc[4];
    const float* gbias[4]; const float* gbias2[4]; const float* gaddend[4];
};

__device__ __forceinline__ GemmParams select_group(const GemmParams& pin, int& z) {
    GemmParams p = pin;
    if (pin.ngroup > 1) {
        const int per = pin.split_k;           // batch == 1 in grouped launches
        const int gi = z / per;
        z = z % per;
        p.A = pin.gA[gi]; p.B = pin.gB[gi]; p.C = pin.gC[gi];
        p.sbk = pin.gsbk[gi]; p.sbn = pin.gsbn[gi]; p.ldc = pin.gldc[gi];
        p.bias = pin.gbias[gi]; p.bias2 = pin.gbias2[gi]; p.addend = pin.gaddend[gi];
    }
    return p;
}

// Shared epilogue: C = act(alpha*acc + beta*C + biases + addend) with optional output row remap; split-K slices add atomically.
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
                float v = p.alpha * acc[i][j][r];
                int orow = row;
                if (p.rowmap_mod > 0) orow = (row % p.rowmap_mod) * p.rowmap_mul + row / p.rowmap_mod;
                float* dst = C + (long)orow * p.ldc + col;
                if (p.split_k > 1) {
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
__global__ __launch_bounds__(BN * 4, 4) void gemm_split_kernel(GemmParams pin) {
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

// Up to 4 problems of identical shape/layout/epilogue mode in ONE launch (the three streams' W_ih / W_hh products): fills the
// chip better than three 640-workgroup grids and pays one launch ramp.  Problems may share C when they accumulate (beta = 1).
int gemm_grouped(const echr_gemm_desc* ds, int ng, hipStream_t st) {
    ECHR_REQUIRE(ds && ng >= 1 && ng <= 4, "gemm_grouped: 1..4 problems");
    for (int i = 1; i < ng; ++i) {
        const echr_gemm_desc &a = ds[0], &b = ds[i];
        ECHR_REQUIRE(a.M == b.M && a.N == b.N && a.K == b.K && a.sam == b.sam && a.sak == b.sak && (a.sbk == 1) == (b.sbk == 1) &&
                     (a.sbn == 1) == (b.sbn == 1) && a.batch == 1 && b.batch == 1 && a.alpha == b.alpha && a.beta == b.beta &&
                     a.act == b.act && a.split_k == b.split_k && a.algo == b.algo && a.rowmap_mod == b.rowmap_mod &&
                     a.add_mod == b.add_mod && a.ld_add == b.ld_add && a.act == ECHR_ACT_NONE,
                     "gemm_grouped: problems must share shape, layout and epilogue mode");
    }
    return gemm_impl(ds, ng, st);
}

static int gemm_impl(const echr_gemm_desc* ds, int ng, hipStream_t st) {
    const echr_gemm_desc& d = ds[0];
    ECHR_REQUIRE(d.A && d.B && d.C, "gemm: null operand");
    ECHR_REQUIRE(d.M > 0 && d.N > 0 && d.K > 0 && d.batch >= 1, "gemm: bad shape M=%d N=%d K=%d batch=%d", d.M, d.N, d.K, d.batch);
    ECHR_REQUIRE(d.sam == 1 || d.sak == 1, "gemm: A needs a unit stride (sam=%ld sak=%ld)", (long)d.sam, (long)d.sak);
    ECHR_REQUIRE(d.sbk == 1 || d.sbn == 1, "gemm: B needs a unit stride (sbk=%ld sbn=%ld)", (long)d.sbk, (long)d.sbn);
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
    p.ngroup = ng;
    for (int i = 0; i < 4; ++i) {
        const echr_gemm_desc& g = ds[i < ng ? i : 0];
        p.gA[i] = g.A; p.gB[i] = g.B; p.gC[i] = g.C; p.gsbk[i] = g.sbk; p.gsbn[i] = g.sbn; p.gldc[i] = g.ldc;
        p.gbias[i] = g.bias; p.gbias2[i] = g.bias2; p.gaddend[i] = g.addend;
    }
    const bool akc = (d.sak == 1);
    const bool bkc = (d.sbk == 1);
    const bool use_split = d.algo == ECHR_GEMM_BF16X3 && config().gemm_bf16x3 && akc && bkc && d.K % 4 == 0 && d.K >= 4 && d.sam % 4 == 0 &&
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
    if (use_split) { BMs = 128; BNs = ((long)((d.M + 127) / 128) * ((d.N + 127) / 128) * d.batch >= 96) ? 128 : 64; }
    if (const char* e = getenv("ECHR_GEMM_TILE")) {          // tuning override (tools/gemm_bench.py); never set in production
        if (e[0] == '1') { BMs = 128; BNs = 128; } else if (e[0] == '6') { BMs = 64; BNs = 64; }
        else if (e[0] == 'a') { BMs = 128; BNs = 64; } else if (e[0] == 'b') { BMs = 64; BNs = 128; }
        else if (e[0] == 'c') { BMs = 128; BNs = 128; w8 = true; }
        if (use_split) { BMs = 128; BNs = (e[0] == 's') ? 64 : 128; }
    }
    p.tiles_m = (d.M + BMs - 1) / BMs;
    p.tiles_n = (d.N + BNs - 1) / BNs;
    const int kt_total = (d.K + BK - 1) / BK;
    int split = d.split_k;
    const bool accumulate = (d.split_k > 1 || d.split_k < 0);   // caller promises C already holds its base value
    if (split < 0) {  // auto: fill ~2 waves of workgroups over the chip when the output grid is small
        long wgs = (long)p.tiles_m * p.tiles_n * d.batch * ng;
        split = 1;
        // latency-bound regime: fewer than 2 workgroups per CU.  Split K so that ~1024 workgroups overlap each other's
        // load latency, keeping at least 4 k-tiles (128 deep) per split (measured optimum on the weight-gradient shapes).
        if (d.act == ECHR_ACT_NONE && wgs < (use_split ? 200 : 512) && (d.beta == 0.f || d.beta == 1.f) && d.rowmap_mod == 0) {
            split = (int)min((long)kt_total, max(1L, ((use_split ? 400 : 1024) + wgs - 1) / max(wgs, 1L)));
            if (split > 1 && kt_total / split < 4) split = max(1, kt_total / 4);
        }
    }
    if (const char* e = getenv("ECHR_GEMM_SPLIT")) { if (d.split_k < 0 && atoi(e) > 0) split = atoi(e); }
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
                    int rc = fill_zero_2d(ds[gi].C + (long)bb * d.bsc, d.M, d.N, ds[gi].ldc, st);
                    if (rc) return rc;
                }
            }
        }
        if (split == 1) p.beta = d.beta;
    } else if (accumulate && split == 1) p.beta = 1.f;
    dim3 grid(p.tiles_m * p.tiles_n, 1, d.batch * split * ng);
    static const bool log_on = getenv("ECHR_GEMM_LOG") != nullptr;
    if (log_on) fprintf(stderr, "[gemm] M=%d N=%d K=%d batch=%d %s%s tile=%dx%d split=%d algo=%s wgs=%d\n", d.M, d.N, d.K, d.batch, akc ? "N" : "T",
                        bkc ? "T" : "N", BMs, BNs, split, use_split ? "bf16x3" : "f32", (int)(grid.x * grid.z));
    // algorithmic work of this launch: 2MNK flops; one read of A and B, one write of C
    ProfScope prof(use_split ? PROF_GEMM_SPLIT : PROF_GEMM, 2.0 * d.M * d.N * d.K * d.batch * ng, 4.0 * ((double)d.M * d.K + (double)d.K * d.N + (double)d.M * d.N) * d.batch * ng, st);
    if (use_split && BNs == 128) hipLaunchKernelGGL(gemm_split_kernel<128>, grid, dim3(512), 0, st, p);
    else if (use_split) hipLaunchKernelGGL(gemm_split_kernel<64>, grid, dim3(256), 0, st, p);
    else if (BMs == 128 && BNs == 128 && w8) launch_cfg<128, 128, 64, 32>(p, akc, bkc, grid, st);
    else if (BMs == 128 && BNs == 128) launch_cfg<128, 128, 64, 64>(p, akc, bkc, grid, st);
    else if (BMs == 128) launch_cfg<128, 64, 64, 32>(p, akc, bkc, grid, st);
    else if (BNs == 128) launch_cfg<64, 128, 32, 64>(p, akc, bkc, grid, st);
    else launch_cfg<64, 64, 32, 32>(p, akc, bkc, grid, st);
    return check_launch("gemm_f32");
}

}  // namespace echr

extern "C" int echr_gemm_f32(const echr_gemm_desc* d, void* stream) {
    if (!d) { echr::set_error("echr_gemm_f32: null descriptor"); return -22; }
    return echr::gemm(*d, static_cast<hipStream_t>(stream));
}
