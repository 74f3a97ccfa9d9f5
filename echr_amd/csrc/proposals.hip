// Proposal selection on device: the index outputs of eval_utils.gettop1000 (eval_utils.py:259-287).
//   masked = scores * mask;  thr = the topN-th largest masked value (ties included);  keep (n,k) with n >= k and
//   masked[n,k] >= max(thr, val_thres), enumerated n-major / k-minor:  ind = n, feat = [n-k, n+1], conf = masked[n,k].
// One 1024-thread workgroup: an exact 4-pass (8 bits each) radix select over order-preserving uint32 keys finds thr, then
// an ordered compaction (block prefix sums over 1024-element chunks) writes the lists.  Integer outputs are bit-exact.
#include "echr_common.h"
#include "echr_internal.h"

namespace echr {

__device__ __forceinline__ unsigned order_key(float f) {          // monotone float -> uint map
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__global__ __launch_bounds__(1024) void top_proposals_kernel(const float* __restrict__ scores, const float* __restrict__ mask, int T, int K,
                                                             int topN, float val_thres, int* __restrict__ out_ind,
                                                             int* __restrict__ out_feat, float* __restrict__ out_conf,
                                                             int* __restrict__ out_count) {
    __shared__ unsigned hist[256];
    __shared__ unsigned s_prefix, s_rank;
    __shared__ int s_scan[1024];
    __shared__ int s_base;
    const long n = (long)T * K;
    const int tid = threadIdx.x;
    // ---- radix select of the r-th largest key, r = min(n, topN) (1-based) ----
    if (tid == 0) { s_prefix = 0u; s_rank = (unsigned)min((long)topN, n); }
    __syncthreads();
    for (int pass = 0; pass < 4; ++pass) {
        const int shift = 24 - 8 * pass;
        const unsigned himask = pass == 0 ? 0u : (0xFFFFFFFFu << (shift + 8));
        if (tid < 256) hist[tid] = 0u;
        __syncthreads();
        const unsigned prefix = s_prefix;
        for (long i = tid; i < n; i += 1024) {
            const unsigned key = order_key(scores[i] * mask[i]);
            if ((key & himask) == prefix) atomicAdd(&hist[(key >> shift) & 255u], 1u);
        }
        __syncthreads();
        if (tid == 0) {
            unsigned r = s_rank, b = 255;
            for (;; --b) {                          // walk from the largest digit down
                if (hist[b] >= r) break;
                r -= hist[b];
                if (b == 0) break;
            }
            s_rank = r;
            s_prefix = prefix | (b << shift);
        }
        __syncthreads();
    }
    const unsigned thr_key = max(s_prefix, order_key(val_thres));
    // ---- ordered compaction ----
    if (tid == 0) s_base = 0;
    __syncthreads();
    for (long c0 = 0; c0 < n; c0 += 1024) {
        const long i = c0 + tid;
        int flag = 0, row = 0, col = 0;
        float v = 0.f;
        if (i < n) {
            row = (int)(i / K); col = (int)(i % K);
            v = scores[i] * mask[i];
            flag = (row >= col && order_key(v) >= thr_key) ? 1 : 0;
        }
        s_scan[tid] = flag;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {      // inclusive Hillis-Steele scan
            const int add = tid >= off ? s_scan[tid - off] : 0;
            __syncthreads();
            s_scan[tid] += add;
            __syncthreads();
        }
        if (flag) {
            const int pos = s_base + s_scan[tid] - 1;
            out_ind[pos] = row;
            out_feat[2 * pos] = row - col;
            out_feat[2 * pos + 1] = row + 1;
            out_conf[pos] = v;
        }
        __syncthreads();
        if (tid == 1023) s_base += s_scan[1023];
        __syncthreads();
    }
    if (tid == 0) out_count[0] = s_base;
}

// Greedy 1-D non-maximum suppression (eval_utils.gettop1000_nms, eval_utils.py:290-331).  Candidates (n, k < min(n, K)) = segments
// [n-k, n+1]; repeat up to topN times: pick the best live candidate (ties -> the later one in n-major order, i.e. the last element
// of a stable ascending sort), kill every live candidate whose inclusive temporal IoU with it exceeds `overlap`.  IoU in float64
// with the reference's operation order (wh / (area_i + area_j - wh)), so the comparison against `overlap` is bit-identical.
// `live` is a [T*K] float scratch: the score while the candidate is live, -inf otherwise.
__global__ __launch_bounds__(1024) void top_proposals_nms_kernel(const float* __restrict__ scores, int T, int K, int topN, double overlap,
                                                                 float* __restrict__ live, int* __restrict__ out_feat,
                                                                 float* __restrict__ out_conf, int* __restrict__ out_count) {
    __shared__ float s_val[16];
    __shared__ int s_idx[16];
    __shared__ int s_pick;
    const long n = (long)T * K;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (long i = tid; i < n; i += 1024) {
        const int row = (int)(i / K), col = (int)(i % K);
        live[i] = col < min(row, K) ? scores[i] : -INFINITY;
    }
    __syncthreads();
    int picked = 0;
    for (; picked < topN; ++picked) {
        float bv = -INFINITY;
        int bi = -1;
        for (long i = tid; i < n; i += 1024) {
            const float v = live[i];
            if (v > bv || (v == bv && v != -INFINITY)) { bv = v; bi = (int)i; }        // ascending i: ties keep the larger index
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const float ov = __shfl_xor(bv, off, 64);
            const int oi = __shfl_xor(bi, off, 64);
            if (ov > bv || (ov == bv && oi > bi)) { bv = ov; bi = oi; }
        }
        if (lane == 0) { s_val[wave] = bv; s_idx[wave] = bi; }
        __syncthreads();
        if (tid == 0) {
            float v = s_val[0]; int ix = s_idx[0];
            for (int w = 1; w < 16; ++w) if (s_val[w] > v || (s_val[w] == v && s_idx[w] > ix)) { v = s_val[w]; ix = s_idx[w]; }
            s_pick = (v == -INFINITY) ? -1 : ix;
            if (s_pick >= 0) {
                const int row = ix / K, col = ix % K;
                out_feat[2 * picked] = row - col;
                out_feat[2 * picked + 1] = row + 1;
                out_conf[picked] = v;
            }
        }
        __syncthreads();
        const int pk = s_pick;
        if (pk < 0) break;
        const int prow = pk / K, pcol = pk % K;
        const double pt1 = prow - pcol, pt2 = prow + 1, parea = pt2 - pt1 + 1.0;
        for (long i = tid; i < n; i += 1024) {
            if (live[i] == -INFINITY) continue;
            const int row = (int)(i / K), col = (int)(i % K);
            const double t1 = row - col, t2 = row + 1;
            const double wh = fmax(0.0, fmin(pt2, t2) - fmax(pt1, t1) + 1.0);
            const double o = wh / (parea + (t2 - t1 + 1.0) - wh);
            if (i == pk || !(o <= overlap)) live[i] = -INFINITY;
        }
        __syncthreads();
    }
    if (tid == 0) out_count[0] = picked;
}

}  // namespace echr

using namespace echr;

extern "C" int echr_top_proposals_nms(const float* scores, int32_t T, int32_t K, int32_t topN, double overlap, float* scratch,
                                      int32_t* out_feat, float* out_conf, int32_t* out_count, void* stream) {
    ECHR_REQUIRE(scores && scratch && out_feat && out_conf && out_count && T > 0 && K > 0 && topN > 0, "top_proposals_nms: bad arguments");
    hipLaunchKernelGGL(top_proposals_nms_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, scores, T, K, topN, overlap, scratch, out_feat,
                       out_conf, out_count);
    return check_launch("top_proposals_nms");
}


extern "C" int echr_top_proposals(const float* scores, const float* mask, int32_t T, int32_t K, int32_t topN, float val_thres,
                                  int32_t* out_ind, int32_t* out_feat, float* out_conf, int32_t* out_count, void* stream) {
    ECHR_REQUIRE(scores && mask && out_ind && out_feat && out_conf && out_count && T > 0 && K > 0 && topN > 0, "top_proposals: bad arguments");
    hipLaunchKernelGGL(top_proposals_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, scores, mask, T, K, topN, val_thres, out_ind, out_feat,
                       out_conf, out_count);
    return check_launch("top_proposals");
}
