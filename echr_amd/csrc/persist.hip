// Weights-stationary PERSISTENT recurrence of the three-stream decoder (forward): all S timesteps in ONE launch.
//
// Reference semantics: models/OldModel_NEW.py:801-823 (ThreeStream_Core.forward), :376-401 (Attention.forward), :105-130 (the
// teacher-forced loop).  The launch-per-phase path (decoder.hip: step_fwd) runs five dependent launches per timestep, each of
// which re-streams the 17.7 MB of time-invariant recurrent weights and 33 MB of attention operands; at 3-9 us of work per launch
// the ~2 us launch boundaries and the re-streaming dominate.  Here:
//   * 256 workgroups (one per CU, 256 threads = one wave per SIMD) stay resident for the whole sequence;
//   * the recurrent weights live in LDS for all S steps, pre-arranged as MFMA B fragments (W_hh1 + W_ih1[:,E:] on 128 "gate"
//     workgroups of 4 hidden units each, W_h2a on 32 "q" workgroups of 16 columns, W_hh0 / W_hh2 on 32 workgroups of 16 units each);
//   * the attention operands live in REGISTERS for all S steps: event n's P_all and C3D rows are split over 3 workgroups
//     (43 slots each, 11 per wave, 8 features per lane: 176 VGPRs);
//   * per timestep the attention chain makes three all-to-all hand-offs (h1 -> q -> context -> h1), the two plain LSTM streams
//     one each, on their own clock.  A hand-off is: write-through (sc1) stores -> every storing wave s_waitcnt vmcnt(0) ->
//     workgroup barrier -> ONE lane's agent-scope atomic add on a per-(edge, timestep) counter sharded 8 ways; the consumer polls
//     the shards with sc1 loads, joins a workgroup barrier and reads the payload with 16-byte sc1 loads (MI355X_MICROARCH.md,
//     inter-workgroup visibility, table row 1; measured 2.65 us per hop + 2.5 us per 128 KB ingest, tools/micro/hop_bench.hip).
//     Every exchanged address is written once per launch and read only after its counter is complete; counters are never reused
//     inside a launch and are zeroed by a memset node ahead of it.  Every spin is bounded: a timeout raises an abort word that all
//     pollers watch, the grid drains, and the host sees a sticky error at its next library call.
//   * the softmax over an event's slots is split over 3 workgroups without an exchange: scores are alpha . tanh(.) (+ b_alpha, which
//     cancels), so exp(score) needs no max-shift while sum|alpha| is moderate; the unnormalised context and the sum of exponentials
//     are added atomically per event and normalised by the consumer.  When sum|alpha| > 40 the three workgroups of an event first
//     exchange their local maxima through 8-byte {tag, value} granules (exact max-shifted softmax).
// Products are exact fp32 (v_mfma_f32_16x16x4_f32).  Saved activations (GATES, CS, HS, OUTD, QS, WT, ATT) are written in the layouts
// the backward pass and the batched projections expect, off the critical path (after the hand-off is published).
#include <atomic>
#include <functional>
#include <mutex>
#include "echr_common.h"
#include "echr_internal.h"

namespace echr {

typedef unsigned u32;
typedef u32 u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define ECHR_AGENT __HIP_MEMORY_SCOPE_AGENT

namespace {

constexpr int PH = 512;                 // H = Ha = K of every recurrent product (D is zero-padded to it)
constexpr int PROWS = 64;               // events (MFMA rows)
constexpr int PSL = 43, PSG = 3;        // attention slots per workgroup (3 workgroups per event: A <= 129) / per 16-lane row (16 rows)
constexpr int NG1 = 128, NQ = 32, NATT = 192, NS = 32;
constexpr int B_Q0 = NG1, NWG = 256;
constexpr int SHARDS = 8, SHSTRIDE = 32;                 // one 128-byte line per counter shard
constexpr int CNT_LINE = SHARDS * SHSTRIDE;              // u32 per counter
enum { C_H1 = 0, C_Q = 1, C_C = 2, C_H0 = 3, C_H2 = 4, C_KINDS = 5 };
constexpr u32 SPIN_LIMIT = 4000000;                      // ~ seconds
constexpr int WU_LD = 264;                               // unnormalised attention weights per event: up to 2 x 129 slots (second slot set, below)
constexpr int NCB = 2, NCBB = 3;                         // row buffers of the BIG kernels' C3D stream: forward / reverse (what their registers hold without spilling)
constexpr int PSET2 = 3 * PSL;                           // first slot of an event's SECOND slot set (events longer than 129 segments)
constexpr int LDS_W = 128 * 1024, LDS_WA = 64 * 1024, LDS_RED = 16 * 1024, LDS_RED_ATT = 32 * 1024 + 2048 + 2048 + 256;
constexpr int LDS_BYTES_LSTM = LDS_W + LDS_RED + 256, LDS_BYTES_ATT = LDS_WA + LDS_RED_ATT + 256;
constexpr float ALPHA_SAFE = 40.f;

}  // namespace

struct PersistLayout { long cnt, xc, xs, gran, zero_end, xh1, xh0, xh2, xq, wu, total; };
static PersistLayout persist_layout(int S) {
    PersistLayout L;
    long off = 0;
    auto take = [&](long n) { long o = off; off += (n + 63) / 64 * 64; return o; };
    L.cnt = take((long)C_KINDS * (S + 1) * CNT_LINE);
    L.xc = take((long)S * PROWS * PH);
    L.xs = take((long)S * PROWS);
    L.gran = take((long)S * PROWS * 3 * 2);
    L.zero_end = off;
    L.xh1 = take((long)S * PROWS * PH);
    L.xh0 = take((long)S * PROWS * PH);
    L.xh2 = take((long)S * PROWS * PH);
    L.xq = take((long)S * PROWS * PH);
    L.wu = take((long)S * PROWS * WU_LD);
    L.total = off;
    return L;
}
long persist_fwd_ws_floats(int S);

struct PersistK {
    const float* evb0;             // optional [N][4H]: event part of stream 0's gate pre-activations, added by the LSTM role itself (nullptr: GATES[0] already holds it)
    int N, A, D, S, ld_att;
    const float* w_hh[3]; const float* w_h2a; const float* b_h2a; const float* w_att; const float* w_alpha;
    const float* PALL; const float* c3d; const int* ev_start; const int* ev_len;
    float* GATES[3]; float* CS[3]; float* HS; float* OUTD; float* QS; float* WT; float* ATT;
    float *XH1, *XH0, *XH2, *XQ, *XC, *XS, *WU;
    unsigned long long* GRAN;
    u32* cnt; u32* abort_word; u32* host_flag;
    u32 spin_limit, inject;          // bound of every hand-off spin; diagnostic: the wait whose code equals `inject` never completes (0 = none)
    unsigned long long* stamps;      // diagnostic: [4 roles][S][16] s_memrealtime stamps (null = off)
    DropCfg dh, dout;
    int nt_saved;                    // saved activations (gate activations, cell states) leave with non-temporal stores
};
#define STAMP(role, i) do { if (P.stamps && tid == 0) P.stamps[((role) * S + t) * 16 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)

// ---- hand-off primitives -------------------------------------------------------------------------------------------------
__device__ __forceinline__ __amdgpu_buffer_rsrc_t mk_rsrc(const void* p, u32 bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
__device__ __forceinline__ float4 ld16_sc1(__amdgpu_buffer_rsrc_t r, u32 off) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 16);      // aux 16 = sc1
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
// bulk operand fragments of a hand-off (tens of KB per workgroup): sc1 like every other load of handed-off bytes (plain loads were timed
// too: no faster, the consumers of one operand do not share its fetch through L2)
__device__ __forceinline__ float4 ld16_bulk(__amdgpu_buffer_rsrc_t r, u32 off) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 16);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
// read-only operands (C3D rows): default cache policy
__device__ __forceinline__ float4 ld16_plain(__amdgpu_buffer_rsrc_t r, u32 off) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
__device__ __forceinline__ void st16_sc1(__amdgpu_buffer_rsrc_t r, u32 off, float4 v) {
    u32x4 u = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
    __builtin_amdgcn_raw_buffer_store_b128(u, r, off, 0, 16);
}
__device__ __forceinline__ void st4_sc1(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, ECHR_AGENT); }
__device__ __forceinline__ float ld4_sc1(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, ECHR_AGENT); }

// sum over the 16 lanes of a DPP row, result in every lane: quad xor 1, quad xor 2, row_half_mirror, row_mirror
template <int CTRL> __device__ __forceinline__ float dpp_mov(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float row16_sum(float v) {
    v += dpp_mov<0xB1>(v); v += dpp_mov<0x4E>(v); v += dpp_mov<0x141>(v); v += dpp_mov<0x140>(v);
    return v;
}
// sum over the 4 DPP rows (16-lane groups) of the wave, lane by lane, result in every row: v_permlane16_swap / v_permlane32_swap with
// both operands the same register return {x[l], x[l ^ 16]} resp. {x[l], x[l ^ 32]} (checked on gfx950: tools/micro/permlane_swap.hip)
__device__ __forceinline__ float xrow_sum4(float v) {
    const unsigned u = __float_as_uint(v);
    const auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    const float s = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    const unsigned u2 = __float_as_uint(s);
    const auto b = __builtin_amdgcn_permlane32_swap(u2, u2, false, false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

// which of a counter's SHARDS lines a workgroup adds to: x + slot of its (XCD x, slot) dispatch position, so that the producers of one edge
// spread evenly over the lines under BOTH role placements (persist_role_index: with XCD-aware roles a half machine's workgroups share three
// values of blockIdx.x % 8, which alone would put 32 adders on each of three lines instead of 12 on each of eight)
__device__ __forceinline__ int publish_shard() { return (int)((blockIdx.x + (blockIdx.x >> 3)) % SHARDS); }
// every thread of the workgroup, after its write-through stores / atomics
__device__ __forceinline__ void publish(u32* line) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(line + publish_shard() * SHSTRIDE, 1u, __ATOMIC_RELAXED, ECHR_AGENT);
}
// the same, leaving the wave's KEEP youngest vector-memory operations in flight: loads issued AFTER the stores / atomics being published
// (the counter retires in issue order, so everything older than those loads has completed)
template <int KEEP>
__device__ __forceinline__ void publish_keep(u32* line) {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(KEEP) : "memory");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(line + publish_shard() * SHSTRIDE, 1u, __ATOMIC_RELAXED, ECHR_AGENT);
}

// every thread; false = the launch is being aborted (timeout somewhere): the caller returns
template <typename PK>
__device__ __forceinline__ bool wait_total(const PK& P, u32* line, u32 target, int* flag, u32 code) {
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        u32 spins = 0;
        bool ok = false;
        for (;;) {
            u32 v = lane < SHARDS ? __hip_atomic_load(line + lane * SHSTRIDE, __ATOMIC_RELAXED, ECHR_AGENT) : 0u;
            v += __shfl_xor(v, 4); v += __shfl_xor(v, 2); v += __shfl_xor(v, 1);
            v = __shfl(v, 0);
            if (v >= target && code != P.inject) { ok = true; break; }
            if ((++spins & 31) == 0) {
                if (__hip_atomic_load(P.abort_word, __ATOMIC_RELAXED, ECHR_AGENT)) break;
                if (spins > P.spin_limit) {
                    if (lane == 0) {
                        __hip_atomic_store(P.abort_word, code, __ATOMIC_RELAXED, ECHR_AGENT);
                        __hip_atomic_store(P.host_flag, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    }
                    break;
                }
            }
            // (no s_sleep between polls: a poll is one dependent round trip, and 64 idle clocks per poll measured +8 us per iteration -- round 6)
        }
        if (lane == 0) *flag = ok ? 1 : 0;
    }
    __syncthreads();
    const bool r = *flag != 0;
    __syncthreads();
    return r;
}

// two counters in one polling loop (an already satisfied wait still costs a round trip to L2 plus two barriers: ~0.7 us each when taken one
// after the other)
template <typename PK>
__device__ __forceinline__ bool wait_total2(const PK& P, u32* line_a, u32 target_a, u32* line_b, u32 target_b, int* flag, u32 code) {
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        u32 spins = 0;
        bool ok = false;
        for (;;) {
            u32 v = lane < 2 * SHARDS ? __hip_atomic_load((lane < SHARDS ? line_a : line_b) + (lane & (SHARDS - 1)) * SHSTRIDE, __ATOMIC_RELAXED, ECHR_AGENT) : 0u;
            v += __shfl_xor(v, 4); v += __shfl_xor(v, 2); v += __shfl_xor(v, 1);
            const u32 va = __shfl(v, 0), vb = __shfl(v, SHARDS);
            if (va >= target_a && vb >= target_b && code != P.inject) { ok = true; break; }
            if ((++spins & 31) == 0) {
                if (__hip_atomic_load(P.abort_word, __ATOMIC_RELAXED, ECHR_AGENT)) break;
                if (spins > P.spin_limit) {
                    if (lane == 0) {
                        __hip_atomic_store(P.abort_word, code, __ATOMIC_RELAXED, ECHR_AGENT);
                        __hip_atomic_store(P.host_flag, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    }
                    break;
                }
            }
            // (no s_sleep between polls: a poll is one dependent round trip, and 64 idle clocks per poll measured +8 us per iteration -- round 6)
        }
        if (lane == 0) *flag = ok ? 1 : 0;
    }
    __syncthreads();
    const bool r = *flag != 0;
    __syncthreads();
    return r;
}

// a wait whose counters were sampled EARLIER (peek_issue: the loads ride behind other work, peek_ready: no memory latency left): the usual
// case -- the producers finished long ago -- then costs two barriers instead of a round trip to L2; otherwise the polling loop takes over
__device__ __forceinline__ u32 peek_issue(u32* line_a, u32* line_b) {
    const int lane = threadIdx.x;
    return lane < 2 * SHARDS ? __hip_atomic_load((lane < SHARDS ? line_a : line_b) + (lane & (SHARDS - 1)) * SHSTRIDE, __ATOMIC_RELAXED, ECHR_AGENT) : 0u;
}
template <typename PK>
__device__ __forceinline__ bool wait_peeked2(const PK& P, u32 peek, u32* line_a, u32 target_a, u32* line_b, u32 target_b, int* flag, u32 code) {
    if (threadIdx.x < 64) {
        u32 v = peek;
        v += __shfl_xor(v, 4); v += __shfl_xor(v, 2); v += __shfl_xor(v, 1);
        const u32 va = __shfl(v, 0), vb = __shfl(v, SHARDS);
        if (threadIdx.x == 0) *flag = (va >= target_a && vb >= target_b && code != P.inject) ? 1 : 0;
    }
    __syncthreads();
    const bool r = *flag != 0;
    __syncthreads();
    return r ? true : wait_total2(P, line_a, target_a, line_b, target_b, flag, code);
}

// ---- MFMA pieces: one wave multiplies its k range [128 w, 128 w + 128) of a [64 x 512] A operand by 16-column tiles ------------
// A fragments straight from the exchange buffer: lane (r = l & 15, kq = l >> 4) holds, for row block rb and k chunk c, the
// float4 A[16 rb + r][128 w + 16 c + 4 kq ..+3]; element j of it feeds MFMA j of the chunk (B uses the same k pairing).
// LAYOUT 0: [k/4][64][4] (h1), 1: [k/16][64][16] (h0, h2, context), 2: row-major [64][512]
template <int LAYOUT>
__device__ __forceinline__ void load_afrag(float4 (&a)[4][8], __amdgpu_buffer_rsrc_t rs, int w, int lane) {
    const int r = lane & 15, kq = lane >> 4;
#pragma unroll
    for (int c = 0; c < 8; ++c)             // chunk-major issue order: the MFMAs of chunk c only wait for the first 4 (c + 1) loads
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) {
            const int n = 16 * rb + r, k = 128 * w + 16 * c + 4 * kq;
            u32 off;
            if (LAYOUT == 0) off = (u32)(((k >> 2) * PROWS + n) * 16);
            else if (LAYOUT == 1) off = (u32)((((k >> 4) * PROWS + n) * 16 + (k & 15)) * 4);
            else off = (u32)((n * PH + k) * 4);
            a[rb][c] = ld16_bulk(rs, off);
        }
}

// acc[rb] += A . B for one 16-column tile whose B image starts at bimg (float4 [8 chunks][64 lanes] of this wave)
__device__ __forceinline__ void mfma_tile(f32x4 (&acc)[4], const float4 (&a)[4][8], const float4* bimg, int lane) {
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const float4 b = bimg[c * 64 + lane];
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) acc[rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[rb][c].x, b.x, acc[rb], 0, 0, 0);
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) acc[rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[rb][c].y, b.y, acc[rb], 0, 0, 0);
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) acc[rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[rb][c].z, b.z, acc[rb], 0, 0, 0);
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) acc[rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[rb][c].w, b.w, acc[rb], 0, 0, 0);
    }
}

// the wave's partial [64 x 16] tile -> red[w][row][col] (C/D map: col = l & 15, row = 16 rb + 4 (l >> 4) + reg)
__device__ __forceinline__ void acc_to_lds(const f32x4 (&acc)[4], float* red, int w, int lane) {
#pragma unroll
    for (int rb = 0; rb < 4; ++rb)
#pragma unroll
        for (int g = 0; g < 4; ++g) red[(w * PROWS + 16 * rb + 4 * (lane >> 4) + g) * 16 + (lane & 15)] = acc[rb][g];
}

// B image of one 16-column tile: float4 index ((w * 8 + c) * 64 + lane) <- W[row(cc)][k .. k+3], zero beyond K
template <typename RowFn>
__device__ __forceinline__ void fill_bimg(float4* img, const float* W, long ld, int K, RowFn row_of, int tid) {
    for (int idx = tid; idx < 4 * 8 * 64; idx += 256) {
        const int lane = idx & 63, c = (idx >> 6) & 7, w = idx >> 9;
        const int cc = lane & 15, kq = lane >> 4;
        const int k = 128 * w + 16 * c + 4 * kq;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (k < K) v = *reinterpret_cast<const float4*>(W + (long)row_of(cc) * ld + k);
        img[idx] = v;
    }
}

struct CellOut { float c, h, hd; float gi, gf, gg, go; };
// nn.LSTMCell gate math (order i, f, g, o) + the two dropouts (OldModel_NEW.py:808-818, :136)
// mh / mo: the recurrent and the late-fusion dropout multipliers of the element (index-only work: computed ahead of the hand-off waits)
__device__ __forceinline__ CellOut lstm_cell(float pi, float pf, float pg, float po, float c_prev, float mh, float mo) {
    CellOut o;
    o.gi = fast_sigmoid(pi); o.gf = fast_sigmoid(pf); o.gg = fast_tanh(pg); o.go = fast_sigmoid(po);
    o.c = o.gf * c_prev + o.gi * o.gg;
    o.h = o.go * fast_tanh(o.c) * mh;
    o.hd = o.h * mo;
    return o;
}
__device__ __forceinline__ float mask_h(const DropCfg& dh, int n, int j, int k, int t) { return drop_mult(dh, (unsigned)(n * PH + j), (unsigned)t, (unsigned)(1 + k)); }      // SITE_H0 + k
__device__ __forceinline__ float mask_o(const DropCfg& dout, int n, int j, int k, int t) { return drop_mult(dout, (unsigned)(n * 3 * PH + k * PH + j), (unsigned)t, 4u); }   // SITE_OUT

// ---- kernel 1: the two plain LSTM streams (0: event context, 2: scene context), 32 workgroups of 16 hidden units each -------------
__device__ __forceinline__ void dec_persist_lstm_body(const PersistK& P, const int bid) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    float4* wimg = reinterpret_cast<float4*>(lds);
    float* red = reinterpret_cast<float*>(lds + LDS_W);
    int* flag = reinterpret_cast<int*>(lds + LDS_W + LDS_RED);
    const int b = bid, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const bool is_s0 = b < NS;
    const int N = P.N, S = P.S;
    const int k = is_s0 ? 0 : 2, bs = is_s0 ? b : b - NS, ck = is_s0 ? C_H0 : C_H2;
    float* XH = is_s0 ? P.XH0 : P.XH2;
    auto cnt = [&](int kind, int t) { return P.cnt + ((long)kind * (S + 1) + t) * CNT_LINE; };
    {
        const float* W = P.w_hh[k];
        for (int ct = 0; ct < 4; ++ct) {
            auto row = [&](int cc) { return (cc >> 2) * PH + 16 * bs + 4 * ct + (cc & 3); };       // tile column cc = gate * 4 + unit
            fill_bimg(wimg + ct * 2048, W, PH, PH, row, tid);
        }
    }
    __syncthreads();
    const int gn = tid >> 2, gu = tid & 3;          // gate-math ownership: thread (event n, unit u)
    float cs[4] = {0.f, 0.f, 0.f, 0.f};
    const u32 XB = PROWS * PH * 4;
    const bool st_on = b == 0;
    for (int t = 0; t < S; ++t) {
        if (st_on) STAMP(3, 0);
        float pre[4][4];
        const float* grow = P.GATES[k] + ((long)t * N + min(gn, N - 1)) * 4 * PH + 16 * bs + gu;
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
#pragma unroll
            for (int g = 0; g < 4; ++g) pre[ct][g] = grow[g * PH + 4 * ct];
        float mh[4], mo[4];
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
            mh[ct] = mask_h(P.dh, gn, 16 * bs + 4 * ct + gu, k, t);
            mo[ct] = mask_o(P.dout, gn, 16 * bs + 4 * ct + gu, k, t);
        }
        f32x4 acc[4][4];
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
#pragma unroll
            for (int rb = 0; rb < 4; ++rb) acc[ct][rb] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (t > 0) {
            if (!wait_total(P, cnt(ck, t - 1), NS, flag, 1000u * (ck + 1) + t)) return;
            if (st_on) STAMP(3, 1);
            float4 a[4][8];
            load_afrag<1>(a, mk_rsrc(XH + (long)(t - 1) * PROWS * PH, XB), w, lane);
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) mfma_tile(acc[ct], a, wimg + ct * 2048 + w * 512, lane);
            if (st_on) STAMP(3, 2);
        }
        CellOut co[4];
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
            if (t > 0) {
                acc_to_lds(acc[ct], red, w, lane);
                __syncthreads();
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int o = gn * 16 + 4 * g + gu;
                    pre[ct][g] += red[o] + red[PROWS * 16 + o] + red[2 * PROWS * 16 + o] + red[3 * PROWS * 16 + o];
                }
                __syncthreads();
            }
            co[ct] = lstm_cell(pre[ct][0], pre[ct][1], pre[ct][2], pre[ct][3], cs[ct], mh[ct], mo[ct]);
            cs[ct] = co[ct].c;
        }
        // exchange layout [bs][n][16]: unit 4 ct + u
        float* xo = XH + (long)t * PROWS * PH + (bs * PROWS + gn) * 16 + gu;
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) st4_sc1(xo + 4 * ct, co[ct].h);
        if (st_on) STAMP(3, 3);
        publish(cnt(ck, t));
        if (st_on) STAMP(3, 4);
        if (gn < N) {          // saved activations, off the critical path
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) {
                const int j = 16 * bs + 4 * ct + gu;
                float* go = P.GATES[k] + ((long)t * N + gn) * 4 * PH + j;
                go[0] = co[ct].gi; go[PH] = co[ct].gf; go[2 * PH] = co[ct].gg; go[3 * PH] = co[ct].go;
                P.CS[k][((long)(t + 1) * N + gn) * PH + j] = co[ct].c;
                const long o = ((long)gn * 3 + k) * PH + j;
                P.HS[(long)(t + 1) * N * 3 * PH + o] = co[ct].h;
                P.OUTD[(long)t * N * 3 * PH + o] = co[ct].hd;
            }
        }
    }
}
__global__ __launch_bounds__(256, 1) void dec_persist_lstm_kernel(PersistK P) { dec_persist_lstm_body(P, blockIdx.x); }

// ---- kernel 2: the attention chain (stream 1): 128 gate + 32 q + 32 attention-only workgroups; all 192 hold attention operands ----
__global__ __launch_bounds__(256, 1) void dec_persist_att_kernel(PersistK P) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    float4* wimg = reinterpret_cast<float4*>(lds);
    float* red = reinterpret_cast<float*>(lds + LDS_WA);
    int* flag = reinterpret_cast<int*>(lds + LDS_WA + LDS_RED_ATT);
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const bool is_g1 = b < NG1, is_q = b >= B_Q0;
    const int N = P.N, D = P.D, S = P.S;
    auto cnt = [&](int kind, int t) { return P.cnt + ((long)kind * (S + 1) + t) * CNT_LINE; };

    // ---- one-time: recurrent weights -> LDS (MFMA B-fragment order) ----
    if (is_g1) {
        auto row = [&](int cc) { return (cc >> 2) * PH + 4 * b + (cc & 3); };       // tile column cc = gate * 4 + unit
        fill_bimg(wimg, P.w_hh[1], PH, PH, row, tid);
        fill_bimg(wimg + 2048, P.w_att, P.ld_att, D, row, tid);
    } else if (b < B_Q0 + NQ) {
        auto row = [&](int cc) { return 16 * (b - B_Q0) + cc; };
        fill_bimg(wimg, P.w_h2a, PH, PH, row, tid);
    }
    const bool is_qw = is_q && b < B_Q0 + NQ;

    // ---- one-time: attention operands -> registers: DPP row g = 4 w + (lane >> 4) owns slots g, g + 16, g + 32 of this
    //      workgroup's third of the event; lane r = lane & 15 of the row holds features [32 r, 32 r + 32) of P_all and C3D ----
    float* red2 = red;                                  // [16 rows][512] cross-row context partials (32 KB)
    float* sal = red + 16 * PH;                         // [512] alpha
    float* sx = sal + PH;                               // small scratch: [0..15] row sums, [16..31] row maxima, [32] combined
    const int an = b / 3, ap = b - 3 * an;              // event, third
    const bool att_live = an < N;
    const int grow_ = 4 * w + (lane >> 4), lr = lane & 15;
    int alen = 0;
    float4 Pr[PSG][8], Cr[PSG][8];
    bool use_max = false;
    {
        float asum = 0.f;
        for (int j = tid; j < PH; j += 256) { const float av = P.w_alpha[j]; sal[j] = av; asum += fabsf(av); }
        asum = wave_sum(asum);
        if (lane == 0) sx[w] = asum;
        __syncthreads();
        use_max = (sx[0] + sx[1] + sx[2] + sx[3]) > ALPHA_SAFE;
        __syncthreads();
    }
    if (att_live) {
        alen = P.ev_len[an];
        const long row0 = P.ev_start[an];
#pragma unroll
        for (int i = 0; i < PSG; ++i) {
            const int sl = grow_ + 16 * i;
            const int a = min(PSL * ap + min(sl, PSL - 1), alen - 1);
            const float* pr = P.PALL + (row0 + a) * PH + 32 * lr;
            const float* cr = P.c3d + (row0 + a) * D;
#pragma unroll
            for (int h = 0; h < 8; ++h) {
                // tanh(p + q) = 1 - 2 / (e^{2p} e^{2q} + 1): e^{2p} is constant over the S timesteps and kept INSTEAD of p (arguments
                // clamped to +-43 so that neither factor is 0 or inf: the product then saturates to 0 / inf -> tanh = -1 / +1, never NaN)
                const float4 pv = *reinterpret_cast<const float4*>(pr + 4 * h);
                Pr[i][h] = make_float4(__expf(2.f * fminf(fmaxf(pv.x, -43.f), 43.f)), __expf(2.f * fminf(fmaxf(pv.y, -43.f), 43.f)),
                                       __expf(2.f * fminf(fmaxf(pv.z, -43.f), 43.f)), __expf(2.f * fminf(fmaxf(pv.w, -43.f), 43.f)));
                const int d = 32 * lr + 4 * h;
                float4 v = *reinterpret_cast<const float4*>(cr + min(d, D - 4));
                if (d >= D) v = make_float4(0.f, 0.f, 0.f, 0.f);
                Cr[i][h] = v;
            }
        }
    } else {
#pragma unroll
        for (int i = 0; i < PSG; ++i)
#pragma unroll
            for (int h = 0; h < 8; ++h) Pr[i][h] = Cr[i][h] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();

    const int gn = tid >> 2, gu = tid & 3;          // gate-math ownership: thread (event n, unit u)
    float c1 = 0.f;
    const u32 XB = PROWS * PH * 4;       // bytes of one timestep of an exchange buffer

    const int srole = b == 0 ? 0 : (b == B_Q0 ? 1 : (b == B_Q0 + NQ ? 2 : -1));
    for (int t = 0; t < S; ++t) {
        if (srole >= 0) STAMP(srole, 0);
        f32x4 acc[4];
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) acc[rb] = f32x4{0.f, 0.f, 0.f, 0.f};
        float pre[4] = {0.f, 0.f, 0.f, 0.f};
        float mh1 = 1.f, mo1 = 1.f;
        if (is_g1) {
            const float* grow = P.GATES[1] + ((long)t * N + min(gn, N - 1)) * 4 * PH + 4 * b + gu;
#pragma unroll
            for (int g = 0; g < 4; ++g) pre[g] = grow[g * PH];
            mh1 = mask_h(P.dh, gn, 4 * b + gu, 1, t);
            mo1 = mask_o(P.dout, gn, 4 * b + gu, 1, t);
        }
        // ---- phase A: products with h1(t-1): W_hh1 . h1 (gate workgroups, kept in the accumulators), q = W_h2a . h1 + b ----
        if ((is_g1 || is_qw) && t > 0) {
            if (!wait_total(P, cnt(C_H1, t - 1), NG1, flag, 100000u + t)) return;
            if (srole >= 0) STAMP(srole, 1);
            float4 a[4][8];
            load_afrag<0>(a, mk_rsrc(P.XH1 + (long)(t - 1) * PROWS * PH, XB), w, lane);
            mfma_tile(acc, a, wimg + w * 512, lane);
            if (srole >= 0) STAMP(srole, 2);
        }
        if (is_qw) {
            const int cq = b - B_Q0;
            float4 qv = *reinterpret_cast<const float4*>(P.b_h2a + 16 * cq + 4 * gu);
            if (t > 0) {
                acc_to_lds(acc, red, w, lane);
                __syncthreads();
                const float* rp = red + gn * 16 + 4 * gu;
#pragma unroll
                for (int ww = 0; ww < 4; ++ww) {
                    const float4 v = *reinterpret_cast<const float4*>(rp + ww * PROWS * 16);
                    qv.x += v.x; qv.y += v.y; qv.z += v.z; qv.w += v.w;
                }
            }
            st16_sc1(mk_rsrc(P.XQ + (long)t * PROWS * PH, XB), (u32)(((cq * PROWS + gn) * 16 + 4 * gu) * 4), qv);
            if (srole >= 0) STAMP(srole, 3);
            publish(cnt(C_Q, t));
            if (srole >= 0) STAMP(srole, 4);
            if (gn < N) *reinterpret_cast<float4*>(P.QS + ((long)t * N + gn) * PH + 16 * cq + 4 * gu) = qv;
        }
        // ---- attention: scores, (split) softmax, context partial ----
        {
            if (!wait_total(P, cnt(C_Q, t), NQ, flag, 200000u + t)) return;
            if (srole >= 0) STAMP(srole, 5);
            if (att_live) {
                const __amdgpu_buffer_rsrc_t rq = mk_rsrc(P.XQ + (long)t * PROWS * PH, XB);
                // q[n, 32 lr .. +32): two 16-column pieces of the exchange layout [c][n][16]
                float4 q[8];
#pragma unroll
                for (int h = 0; h < 8; ++h) q[h] = ld16_sc1(rq, (u32)((((2 * lr + (h >> 2)) * PROWS + an) * 16 + 4 * (h & 3)) * 4));
                if (P.stamps) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); if (srole >= 0) STAMP(srole, 12); }
                // score = sum_j alpha_j tanh(p_j + q_j) = sum_j alpha_j - 2 sum_j alpha_j / (e^{2 p_j} e^{2 q_j} + 1): one exp per q column
                // and step (shared by the row's 3 slots), one fma + rcp + fma per (slot, feature)
                float asum = 0.f;
#pragma unroll
                for (int h = 0; h < 8; ++h) {
                    q[h] = make_float4(__expf(2.f * fminf(fmaxf(q[h].x, -43.f), 43.f)), __expf(2.f * fminf(fmaxf(q[h].y, -43.f), 43.f)),
                                       __expf(2.f * fminf(fmaxf(q[h].z, -43.f), 43.f)), __expf(2.f * fminf(fmaxf(q[h].w, -43.f), 43.f)));
                    const float4 a4 = *reinterpret_cast<const float4*>(sal + 32 * lr + 4 * h);
                    asum += (a4.x + a4.y) + (a4.z + a4.w);
                }
                float e[PSG];
                float mloc = -INFINITY;
#pragma unroll
                for (int i = 0; i < PSG; ++i) {
                    float v = 0.f;
#pragma unroll
                    for (int h = 0; h < 8; ++h) {
                        const float4 a4 = *reinterpret_cast<const float4*>(sal + 32 * lr + 4 * h);
                        v += a4.x * __builtin_amdgcn_rcpf(fmaf(Pr[i][h].x, q[h].x, 1.f)) + a4.y * __builtin_amdgcn_rcpf(fmaf(Pr[i][h].y, q[h].y, 1.f)) +
                             a4.z * __builtin_amdgcn_rcpf(fmaf(Pr[i][h].z, q[h].z, 1.f)) + a4.w * __builtin_amdgcn_rcpf(fmaf(Pr[i][h].w, q[h].w, 1.f));
                    }
                    v = row16_sum(fmaf(-2.f, v, asum));
                    const int sl = grow_ + 16 * i;
                    const bool valid = sl < PSL && PSL * ap + sl < alen;
                    e[i] = valid ? v : -INFINITY;
                    mloc = fmaxf(mloc, e[i]);
                }
                float shift = 0.f;
                if (use_max) {       // exact max-shifted softmax: the event's three workgroups exchange their local maxima (8-byte granules)
                    if (lr == 0) sx[16 + grow_] = mloc;
                    __syncthreads();
                    unsigned long long* gr = P.GRAN + ((long)t * PROWS + an) * 3;
                    if (tid < 64) {
                        float m16 = lane < 16 ? sx[16 + lane] : -INFINITY;
                        m16 = wave_max(m16);
                        if (lane == 0)
                            __hip_atomic_store(gr + ap, ((unsigned long long)(t + 1) << 32) | __float_as_uint(m16), __ATOMIC_RELAXED, ECHR_AGENT);
                        float mm = -INFINITY;
                        u32 spins = 0;
                        for (;;) {
                            unsigned long long x = lane < 3 ? __hip_atomic_load(gr + lane, __ATOMIC_RELAXED, ECHR_AGENT) : ((unsigned long long)(t + 1) << 32) | 0xff800000u;
                            const bool ok = (u32)(x >> 32) == (u32)(t + 1);
                            if (__all(ok)) { mm = __uint_as_float((u32)x); break; }
                            if ((++spins & 31) == 0) {
                                if (__hip_atomic_load(P.abort_word, __ATOMIC_RELAXED, ECHR_AGENT)) break;
                                if (spins > P.spin_limit) {      // timed out: abort the launch like every other bounded spin (shift = 0 below is never consumed)
                                    if (lane == 0) {
                                        __hip_atomic_store(P.abort_word, 9000u + (u32)t, __ATOMIC_RELAXED, ECHR_AGENT);
                                        __hip_atomic_store(P.host_flag, 9000u + (u32)t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                                    }
                                    break;
                                }
                            }
                            // (no s_sleep between polls: a poll is one dependent round trip, and 64 idle clocks per poll measured +8 us per iteration -- round 6)
                        }
                        mm = wave_max(mm);
                        if (lane == 0) sx[32] = mm;
                    }
                    __syncthreads();
                    shift = sx[32];
                    if (!(shift > -INFINITY)) shift = 0.f;
                }
                if (srole >= 0) STAMP(srole, 13);
                float ssum = 0.f;
                float4 cx[8];
#pragma unroll
                for (int h = 0; h < 8; ++h) cx[h] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int i = 0; i < PSG; ++i) {
                    const float x = __expf(e[i] - shift);          // exp(-inf) = 0 for slots past the event's end
                    e[i] = x;
                    ssum += x;
#pragma unroll
                    for (int h = 0; h < 8; ++h) {
                        cx[h].x += x * Cr[i][h].x; cx[h].y += x * Cr[i][h].y; cx[h].z += x * Cr[i][h].z; cx[h].w += x * Cr[i][h].w;
                    }
                }
#pragma unroll
                for (int h = 0; h < 8; ++h) *reinterpret_cast<float4*>(red2 + grow_ * PH + 32 * lr + 4 * h) = cx[h];
                if (lr == 0) sx[grow_] = ssum;
                // unnormalised weights of this row's slots (lane i of the row stores slot grow_ + 16 i)
                {
                    const float xw = lr == 0 ? e[0] : (lr == 1 ? e[1] : e[2]);
                    const int sl = grow_ + 16 * lr;
                    if (lr < PSG && sl < PSL && PSL * ap + sl < alen) st4_sc1(P.WU + ((long)t * PROWS + an) * WU_LD + PSL * ap + sl, xw);
                }
                __syncthreads();
                if (srole >= 0) STAMP(srole, 14);
                // context partial of this workgroup: exchange layout [d / 16][n][16]
                float* xc = P.XC + (long)t * PROWS * PH;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int d = tid + 256 * h;
                    if (d < D) {
                        float sum = 0.f;
#pragma unroll
                        for (int g = 0; g < 16; ++g) sum += red2[g * PH + d];
                        atomicAdd(xc + ((d >> 4) * PROWS + an) * 16 + (d & 15), sum);
                    }
                }
                if (tid == 0) {
                    float sum = 0.f;
#pragma unroll
                    for (int g = 0; g < 16; ++g) sum += sx[g];
                    atomicAdd(P.XS + (long)t * PROWS + an, sum);
                }
            }
            if (srole >= 0) STAMP(srole, 6);
            publish(cnt(C_C, t));
            if (srole >= 0) STAMP(srole, 7);
        }
        // ---- phase C: attended-context columns of stream 1 + gate math; the new h1 is handed to the next step ----
        if (is_g1) {
            if (!wait_total(P, cnt(C_C, t), NATT, flag, 300000u + t)) return;
            if (srole >= 0) STAMP(srole, 8);
            f32x4 accc[4];
#pragma unroll
            for (int rb = 0; rb < 4; ++rb) accc[rb] = f32x4{0.f, 0.f, 0.f, 0.f};
            float4 inv4[4];
            {
                float4 a[4][8];
                const __amdgpu_buffer_rsrc_t rc = mk_rsrc(P.XC + (long)t * PROWS * PH, XB);
                load_afrag<1>(a, rc, w, lane);
                // 1 / sum of exponentials of the rows this lane's accumulator registers belong to (C/D map: row = 16 rb + 4 (l >> 4) + reg)
                const __amdgpu_buffer_rsrc_t rsum = mk_rsrc(P.XS + (long)t * PROWS, PROWS * 4);
#pragma unroll
                for (int rb = 0; rb < 4; ++rb) inv4[rb] = ld16_sc1(rsum, (u32)((16 * rb + 4 * (lane >> 4)) * 4));
                mfma_tile(accc, a, wimg + 2048 + w * 512, lane);
                // normalised context, saved for backward: this workgroup stores features [4b, 4b+4) of every event (wave b/32, chunk (b%32)/4, kq b%4)
                if (4 * b < D && w == (b >> 5) && (lane >> 4) == (b & 3)) {
#pragma unroll
                    for (int c = 0; c < 8; ++c)
                        if (c == ((b & 31) >> 2)) {
#pragma unroll
                            for (int rb = 0; rb < 4; ++rb) {
                                const int n = 16 * rb + (lane & 15);
                                if (n < N) {
                                    const float is = 1.0f / ld4_sc1(P.XS + (long)t * PROWS + n);
                                    *reinterpret_cast<float4*>(P.ATT + ((long)t * N + n) * D + 4 * b) =
                                        make_float4(a[rb][c].x * is, a[rb][c].y * is, a[rb][c].z * is, a[rb][c].w * is);
                                }
                            }
                        }
                }
            }
#pragma unroll
            for (int rb = 0; rb < 4; ++rb) {
                acc[rb][0] += accc[rb][0] / inv4[rb].x; acc[rb][1] += accc[rb][1] / inv4[rb].y;
                acc[rb][2] += accc[rb][2] / inv4[rb].z; acc[rb][3] += accc[rb][3] / inv4[rb].w;
            }
            if (srole >= 0) STAMP(srole, 9);
            acc_to_lds(acc, red, w, lane);
            __syncthreads();
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int o = gn * 16 + 4 * g + gu;
                pre[g] += red[o] + red[PROWS * 16 + o] + red[2 * PROWS * 16 + o] + red[3 * PROWS * 16 + o];
            }
            const CellOut co = lstm_cell(pre[0], pre[1], pre[2], pre[3], c1, mh1, mo1);
            c1 = co.c;
            st4_sc1(P.XH1 + (long)t * PROWS * PH + (b * PROWS + gn) * 4 + gu, co.h);
            if (srole >= 0) STAMP(srole, 10);
            publish(cnt(C_H1, t));          // (its barrier also protects `red` for the next step)
            if (srole >= 0) STAMP(srole, 11);
            // saved activations, off the critical path
            if (gn < N) {
                const int j = 4 * b + gu;
                float* go = P.GATES[1] + ((long)t * N + gn) * 4 * PH + j;
                go[0] = co.gi; go[PH] = co.gf; go[2 * PH] = co.gg; go[3 * PH] = co.go;
                P.CS[1][((long)(t + 1) * N + gn) * PH + j] = co.c;
                const long o = ((long)gn * 3 + 1) * PH + j;
                P.HS[(long)(t + 1) * N * 3 * PH + o] = co.h;
                P.OUTD[(long)t * N * 3 * PH + o] = co.hd;
            }
            // normalised attention weights: rows of event b/2, slots [65 (b&1), +65)
            {
                const int n = b >> 1, a0 = 65 * (b & 1) + tid;
                if (tid < 65 && n < N && a0 < P.A) {
                    const int len = P.ev_len[n];
                    float wv = 0.f;
                    if (a0 < len) wv = ld4_sc1(P.WU + ((long)t * PROWS + n) * WU_LD + a0) / ld4_sc1(P.XS + (long)t * PROWS + n);
                    P.WT[((long)t * N + n) * P.A + a0] = wv;
                }
            }
        }
    }
}


// ==========================================================================================================================
// Forward attention chain, version 2: TWO HALF-CHIP MACHINES.  Events never interact in the decoder, so rows [0,32) and [32,64) are
// two independent recurrences: workgroups [0,96) serve the first half, [96,192) the second, each with a full copy of stream 1's
// recurrent weights in LDS (64 gate workgroups of 8 hidden units = 32 columns, 16 q workgroups of 32 columns, 16 attention-only; all 96
// hold the attention operands of the half's 32 events).  Against version 1 this halves what a workgroup ingests per hand-off (64 KB),
// the fan-in of every counter and the rows per MFMA tile (v_mfma_f32_32x32x2_f32, still exact fp32), at twice the LDS per workgroup.
// Exchange layout of every [32 rows x 512] operand: [k / 8][32 rows][8 floats]: a wave's fragment load (lane = row, k half) covers 1 KB
// of contiguous memory, and a gate workgroup's 8 units are one contiguous 1-KB group.
// ==========================================================================================================================
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int HR = 32;                        // rows per half machine
constexpr int HWG = 96, HG1 = 64, HQ = 16;    // workgroups per half: gate, q (the remaining 16 are attention-only)
// Which role a workgroup of the merged 256-workgroup launches plays.  Plain: blockIdx.x (workgroups 0..95 = half machine 0, 96..191 = half machine
// 1, 192..255 = the two plain LSTM streams).  Workgroup b of a one-workgroup-per-CU grid runs on XCD b % 8 (round-robin dispatch: speed only, never
// correctness), so in that order EVERY half machine -- and with it every handed-off operand -- is spread over all eight XCDs, each of whose L2s
// fetches the operand across the fabric.  xcd_map = 1 (ECHR_PERSIST_XCD): a half machine's 96 workgroups are exactly the 3 x 32 CUs of three
// XCDs (0-2, 3-5), the 64 LSTM-stream workgroups the two remaining ones: an operand is then fetched into three L2s (LSTM: two) instead of eight.
__host__ __device__ __forceinline__ int persist_role_index(int bx, int xcd_map) {
    if (!xcd_map) return bx;
    const int x = bx & 7, s = bx >> 3;
    if (x >= 6) return 2 * HWG + (x - 6) * 32 + s;
    const int xr = x % 3;
    // xcd_map = 2 (reverse pair, ECHR_PERSIST_XCD_BWD): the 64 product workgroups of a half machine (roles 16..79: four k-slices of 16 column tiles)
    // fill two XCDs, two k-slices each, so that each of them fetches only ITS half of d G1; the 16 gate-gradient and 16 attention-only roles share
    // the third
    if (xcd_map == 2) return (x / 3) * HWG + (xr == 0 ? (s < 16 ? s : 64 + s) : (xr == 1 ? 16 + s : 48 + s));
    return (x / 3) * HWG + xr * 32 + s;
}
constexpr int LDS_W2 = 128 * 1024, LDS_RED2 = 16 * 1024 + 2048 + 1536;
constexpr int LDS_BYTES_ATT2 = LDS_W2 + LDS_RED2 + 256;

struct PersistLayout2 { long cnt, xc, xs, gran, xcmax, zero_begin, xh1, xq, wu, pimg, total; };
constexpr long PIMG_G1 = 2 * 16384 + 64, PIMG_Q = 16384 + 64;          // floats per prebuilt image set: gate workgroup (two images + 2 x 32 scales), q workgroup
static PersistLayout2 persist_layout2(int S) {
    PersistLayout2 L;
    long off = 0;
    auto take = [&](long n) { long o = off; off += (n + 63) / 64 * 64; return o; };
    L.xh1 = take((long)S * PROWS * PH);
    L.xq = take((long)S * PROWS * PH);
    L.wu = take((long)S * PROWS * WU_LD);
    L.pimg = take((long)HG1 * PIMG_G1 + (long)HQ * PIMG_Q);          // prebuilt weight images (persist_fwd_prebuild)
    // the zeroed region comes last: the version-1 layout (whose first region, the counters, is the only part of it the LSTM kernel needs
    // zeroed) follows this one in the workspace, so one memset covers both
    L.zero_begin = off;
    L.cnt = take((long)3 * (S + 1) * 2 * CNT_LINE);
    L.xc = take((long)S * PROWS * PH);
    L.xs = take((long)S * PROWS);
    L.gran = take((long)S * PROWS * 3 * 2);
    L.xcmax = take(PROWS);
    L.total = off;
    return L;
}

// A fragments of the wave's k range [128 w, 128 w + 128): lane (r = l & 31, kh = l >> 5), chunk c: float4 A[r][128 w + 8 c + 4 kh ..+3]
__device__ __forceinline__ void load_afrag32(float4 (&a)[16], __amdgpu_buffer_rsrc_t rs, int w, int lane) {
    const int r = lane & 31, kh = lane >> 5;
#pragma unroll
    for (int c = 0; c < 16; ++c) a[c] = ld16_bulk(rs, (u32)((((16 * w + c) * HR + r) * 8 + 4 * kh) * 4));
}
__device__ __forceinline__ void mfma_tile32(f32x16& acc, const float4 (&a)[16], const float4* bimg, int lane) {
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        const float4 b = bimg[c * 64 + lane];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[c].x, b.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[c].y, b.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[c].z, b.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[c].w, b.w, acc, 0, 0, 0);
    }
}
// the wave's partial [32 x 32] tile -> red[w][row][col]  (C/D map: col = l & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (l >> 5))
__device__ __forceinline__ void acc_to_lds32(const f32x16& acc, float* red, int w, int lane) {
#pragma unroll
    for (int g = 0; g < 16; ++g) red[(w * HR + (g & 3) + 8 * (g >> 2) + 4 * (lane >> 5)) * 32 + (lane & 31)] = acc[g];
}
// B image of one 32-column tile, k-contiguous source: float4 index ((w * 16 + c) * 64 + lane) <- W[row_of(cc)][k..k+3], zero beyond K
template <typename RowFn>
__device__ __forceinline__ void fill_bimg32(float4* img, const float* W, long ld, int K, RowFn row_of, int tid) {
    for (int idx = tid; idx < 4 * 16 * 64; idx += 256) {
        const int lane = idx & 63, c = (idx >> 6) & 15, w = idx >> 10;
        const int cc = lane & 31, kh = lane >> 5;
        const int k = 128 * w + 8 * c + 4 * kh;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (k < K) v = *reinterpret_cast<const float4*>(W + (long)row_of(cc) * ld + k);
        img[idx] = v;
    }
}

long persist_fwd_ws_floats(int S) { return persist_layout(S).total + persist_layout2(S).total; }

typedef _Float16 f16x8p __attribute__((ext_vector_type(8)));
constexpr float H2_SA = 4096.f, H2_INV_SA = 1.f / 4096.f;          // activation scale 2^12 (|h| < 16 keeps hi finite)

__device__ __forceinline__ void split_h2(float x, unsigned short& hi, unsigned short& lo) {
    const _Float16 h1 = (_Float16)x;
    const _Float16 h2 = (_Float16)(x - (float)h1);
    hi = __builtin_bit_cast(unsigned short, h1);
    lo = __builtin_bit_cast(unsigned short, h2);
}
__device__ __forceinline__ f16x8p as_f16x8(float4 v) { return __builtin_bit_cast(f16x8p, v); }

// B planes of `ncb` 32-column blocks over K = 512 for one workgroup: image float4 index ((((w * 8 + s) * ncb + cb) * 2 + plane) * 64 + lane),
// lane = (col & 31) + 32 * kh holds W[row_of(32 cb + col)][128 w + 16 s + 8 kh + j], j < 8, times the column's scale 2^(14 - e).
// inv_scale[col] receives 2^(e - 14).  scratch: ncb * 32 floats of LDS for the column scales.
template <typename RowFn>
__device__ __forceinline__ void fill_bimg_h2(float4* img, float* inv_scale, float* scratch, const float* W, long ld, int K, int ncb, RowFn row_of, int tid) {
    // ONE pass over the slice: 8 threads per column hold its 512 k (64 each, all 16 loads of a thread in flight together; K % 4 == 0, rows
    // 16-byte aligned), agree on the column's scale through three lane exchanges, and each converts its own 8 groups of 8 consecutive k --
    // one 16-byte image slot per plane and group.  (The first cut found the maxima in one pass and re-read the slice in image order in a
    // second: two dependent rounds of global loads per image in front of the first timestep.)
    (void)scratch;
    for (int c0 = 0; c0 < 32 * ncb; c0 += 32) {
        const int cl = tid >> 3, col = c0 + cl, qk = tid & 7, cb = c0 >> 5;
        const float* wp = W + (long)row_of(col) * ld + 64 * qk;
        float4 v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = (64 * qk + 4 * i < K) ? *reinterpret_cast<const float4*>(wp + 4 * i) : make_float4(0.f, 0.f, 0.f, 0.f);
        float mx = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) mx = fmaxf(fmaxf(mx, fmaxf(fabsf(v[i].x), fabsf(v[i].y))), fmaxf(fabsf(v[i].z), fabsf(v[i].w)));
        mx = fmaxf(mx, __shfl_xor(mx, 1));
        mx = fmaxf(mx, __shfl_xor(mx, 2));
        mx = fmaxf(mx, __shfl_xor(mx, 4));
        const int ex = (int)((__float_as_uint(mx) >> 23) & 0xFFu);
        int e = (ex == 0 || ex == 255) ? 14 : ex - 127;
        e = max(e, 14 - 126);
        const float sc = __uint_as_float((unsigned)(127 + 14 - e) << 23);
        if (qk == 0) inv_scale[col] = ldexpf(1.f, e - 14);
#pragma unroll
        for (int i = 0; i < 8; ++i) {          // k = 64 qk + 8 i .. + 7  ->  (w, s, kh) = (k / 128, (k % 128) / 16, (k % 16) / 8)
            const int w = qk >> 1, s_ = (qk & 1) * 4 + (i >> 1), kh = i & 1;
            const float xv[8] = {v[2 * i].x, v[2 * i].y, v[2 * i].z, v[2 * i].w, v[2 * i + 1].x, v[2 * i + 1].y, v[2 * i + 1].z, v[2 * i + 1].w};
            unsigned hw[8], lw[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                unsigned short hi, lo;
                split_h2(xv[j] * sc, hi, lo);
                hw[j] = hi; lw[j] = lo;
            }
            const long base = (((long)(w * 8 + s_) * ncb + cb) * 2) * 64 + cl + 32 * kh;
            reinterpret_cast<uint4*>(img)[base] = make_uint4(hw[0] | (hw[1] << 16), hw[2] | (hw[3] << 16), hw[4] | (hw[5] << 16), hw[6] | (hw[7] << 16));
            reinterpret_cast<uint4*>(img)[base + 64] = make_uint4(lw[0] | (lw[1] << 16), lw[2] | (lw[3] << 16), lw[4] | (lw[5] << 16), lw[6] | (lw[7] << 16));
        }
    }
    __syncthreads();
}


struct PersistK2 {
    unsigned c3d_bytes;            // bytes of the whole [Tv, D] feature tensor (buffer-resource bound of the BIG kernels' C3D row stream)
    int N, A, D, S, ld_att;
    const float* w_hh1; const float* w_h2a; const float* b_h2a; const float* w_att; const float* w_alpha;
    const float* PALL; const float* c3d; const int* ev_start; const int* ev_len;
    float* GATES1; float* CS1; float* HS; float* OUTD; float* QS; float* WT; float* ATT;
    float *XH1, *XQ, *XC, *XS, *WU, *XCMAX;
    unsigned long long* GRAN;
    u32* cnt; u32* abort_word; u32* host_flag;
    u32 spin_limit, inject;          // bound of every hand-off spin; diagnostic: the wait whose code equals `inject` never completes (0 = none)
    unsigned long long* stamps;
    DropCfg dh, dout;
    const float* pimg;             // prebuilt weight images of this launch's workspace (nullptr: the workgroups build their own)
    int nt_saved;                  // saved activations (gate activations, cell states) leave with non-temporal stores
    int xcd_map;                   // 1: role index from (XCD, slot) instead of blockIdx.x, see persist_role_index
};


// persist_fwd_prebuild: block lb < HG1 builds gate workgroup lb's two images, block HG1 + j q workgroup j's (both halves of the chip use the same
// images: the weights do not depend on the event group)
__global__ __launch_bounds__(256) void dec_persist_prebuild_kernel(PersistK2 P, float* __restrict__ pimg) {
    const int lb = blockIdx.x, tid = threadIdx.x;
    if (lb < HG1) {
        float* dst = pimg + (long)lb * PIMG_G1;
        auto row = [&](int cc) { return (cc >> 3) * PH + 8 * lb + (cc & 7); };
        fill_bimg_h2(reinterpret_cast<float4*>(dst), dst + 32768, nullptr, P.w_hh1, PH, PH, 1, row, tid);
        fill_bimg_h2(reinterpret_cast<float4*>(dst) + 4096, dst + 32768 + 32, nullptr, P.w_att, P.ld_att, P.D, 1, row, tid);
    } else {
        float* dst = pimg + (long)HG1 * PIMG_G1 + (long)(lb - HG1) * PIMG_Q;
        auto row = [&](int cc) { return 32 * (lb - HG1) + cc; };
        fill_bimg_h2(reinterpret_cast<float4*>(dst), dst + 16384, nullptr, P.w_h2a, PH, PH, 1, row, tid);
    }
}

// Greedy decoding inside the forward kernels (SAMP instantiations; OldModel.sample, models/OldModel_NEW.py:139-187): the token fed at step
// t + 1 is the arg-max of step t's logits, so the three streams can no longer run ahead of each other.  The 64 workgroups of the two plain
// LSTM streams also compute the logits (80 vocabulary columns each, fp16-pair products against a weight image streamed from L2 / MALL):
// the h0 / h2 two thirds of the contraction while the attention chain is still busy with its step, the h1 third when that arrives; every
// workgroup folds its columns into one 64-bit (value, lowest index) key per event with an atomic max -- order-independent, so `seq` is
// bitwise reproducible -- and the consumers gather the token-side gate pre-activations from a [V1, 4H] table per stream.
constexpr int LCOLS = 80, LCT = 5;             // vocabulary columns / 16-column MFMA tiles per logits workgroup
constexpr int LWG = 2 * NS;                    // logits workgroups
constexpr u32 STOP_ALL_FINISHED = 0x7FFFFFF0u;          // written to the launch's abort word: not an error (see persist_sample_group)
constexpr int LCHMAX = 3;                      // column chunks of LWG x LCOLS a workgroup can take: vocabularies of up to 15 360 words
struct PersistS {
    const float* TG[3];            // [V1][4H] token-side gate pre-activations per stream (stream 1: biases folded in)
    const float* base0;            // [N][4H] event part of stream 0 (+ biases)
    const float* base2;            // [4H] video part of stream 2 (+ biases)
    unsigned long long* KEY;       // [S][64] arg-max keys: ordered(value) << 32 | ~index
    u32* cnt_tok;                  // [S] counters: logits workgroups that have folded step t
    const float4* LIMG;            // logit weights as fp16-pair planes [workgroup 64][stream 3][k step 16][tile 5][plane 2][lane 64] x 16 bytes
    const float* linv;             // [64 * 80] inverse column scales
    const float* lbias;            // [V1]
    float* LSE;                    // [S][64 events][64 workgroups][2]: (local maximum, sum of exp(x - local maximum)) for the log-probability
    float* XC3; float* XS3;        // context partials / exponential sums, one slot per workgroup of an event (summed in a fixed order)
    u32* cnt2;                     // the attention layout's counters (the logits role waits for h1)
    const float* XH1;              // its h1 exchange planes
    int V1, nch;                   // nch = ceil(V1 / 5120) column chunks
    int force_eos;                 // diagnostic (tests): > 0 = the logits of steps >= force_eos - 1 are overridden in favour of <eos> (column 0) for every event
    int stop_early;                // 1 = leave the launch once no event of the group is unfinished.  Only when the call has ONE group: OldModel.sample breaks when
                                   // ALL events have finished and keeps appending the raw max log-prob of finished rows until then (OldModel_NEW.py:179-183), so with
                                   // several groups every group computes every step and the host trims at the first step nobody is unfinished at
};

// BIG: events of up to 258 segments (BASELINE config 5's 256-segment proposals).  An event's first 129 slots live in registers as before;
// slots [129, 258) form a SECOND set of 43 per workgroup whose P_all / C3D rows are re-read from L2 every step (a 256-row video's operands are
// 1 MB: L2-resident), scored with the same q and folded into the same split softmax.  Events of <= 129 segments skip it, and the BIG = false
// instantiation (what shapes with A <= 129 launch) is the previous code unchanged.
template <bool H2, bool BIG, bool SAMP = false>
__device__ __forceinline__ void dec_persist_att2_body(const PersistK2& P, const int bid, const PersistS* Q = nullptr) {
    static_assert(!SAMP || H2, "the sampling form exchanges h1 as fp16 planes");
    if (P.stamps && bid == 0 && threadIdx.x == 0) P.stamps[15] = __builtin_amdgcn_s_memrealtime();          // kernel entry (diagnostic)
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    float4* wimg = reinterpret_cast<float4*>(lds);
    float* red = reinterpret_cast<float*>(lds + LDS_W2);          // 16 KB: cross-wave tile sums / cross-row attention partials (8 rows per pass)
    float* sal = red + 4096;                                      // [512] alpha
    float* sx = sal + PH;                                         // small scratch [64]
    float* invbA = sx + 64;                                       // h2: [32] inverse column scales of the phase-A image, [32] of the phase-C image,
    float* invbC = invbA + 32;                                    //     [32] per-row conversion factor of the context, [32] its inverse scale,
    float* sCt = invbC + 32;                                      //     [32] bound on |context| per row, [64] scratch
    float* invAt = sCt + 32;
    float* cmx = invAt + 32;
    float* scr = cmx + 32;
    int* flag = reinterpret_cast<int*>(lds + LDS_W2 + LDS_RED2);
    const int b = bid, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int m = b / HWG, lb = b - m * HWG;
    const int N = P.N, D = P.D, S = P.S;
    if (HR * m >= N) return;                                     // this half machine has no events
    const bool is_g1 = lb < HG1, is_qw = lb >= HG1 && lb < HG1 + HQ;
    auto cnt = [&](int kind, int t) { return P.cnt + (((long)kind * (S + 1) + t) * 2 + m) * CNT_LINE; };

    // ---- attention operands -> registers (as version 1); the raw loads are issued ahead of the weight-image fills so that the two
    //      set-up phases overlap (both are chains of dependent memory latencies) ----
    const int ar = lb / 3, ap = lb - 3 * ar, an = HR * m + ar;    // row within the half, third, event
    const bool att_live = an < N;
    int grow_ = 4 * w + (lane >> 4), lr = lane & 15;          // (not const: the BIG instantiation re-derives their dependants every step, below)
    int alen = 0;
    long row0 = 0;
    // BIG: e^{2p} of BOTH slot sets stays in registers (Pr, Pr2) and the C3D rows of both sets are streamed from L2 every step through three
    // row buffers -- their addresses do not depend on the step, so a step requests its first rows before it waits for q.  (The first cut kept
    // set 1's P_all and C3D rows resident and re-read set 2's P_all AND C3D rows behind the q hop: two chains of dependent L2 round trips per
    // step and 392 B of scratch per lane.)
    float4 Pr[PSG][8], Cr[BIG ? 1 : PSG][8], Pr2[BIG ? PSG : 1][8];
    bool use_max = false;
    {
        float asum = 0.f;
        for (int j = tid; j < PH; j += 256) { const float av = P.w_alpha[j]; sal[j] = av; asum += fabsf(av); }
        asum = wave_sum(asum);
        if (lane == 0) sx[w] = asum;
        __syncthreads();
        use_max = (sx[0] + sx[1] + sx[2] + sx[3]) > ALPHA_SAFE;
        __syncthreads();
    }
    auto sst_ = [&](int i) { if (P.stamps && b == 0 && tid == 0) P.stamps[(i >> 1) * 16 + 12 + (i & 1)] = __builtin_amdgcn_s_memrealtime(); };
    sst_(0);
    if (att_live) {
        alen = P.ev_len[an];
        row0 = P.ev_start[an];
#pragma unroll
        for (int i = 0; i < PSG; ++i) {
            const int sl = grow_ + 16 * i;
            const int a = min(PSL * ap + min(sl, PSL - 1), alen - 1);
            const float* pr = P.PALL + (row0 + a) * PH + 32 * lr;
            const float* cr = P.c3d + (row0 + a) * D;
#pragma unroll
            for (int h = 0; h < 8; ++h) {
                Pr[i][h] = *reinterpret_cast<const float4*>(pr + 4 * h);
                if constexpr (!BIG) {
                    const int d = 32 * lr + 4 * h;
                    float4 v = *reinterpret_cast<const float4*>(cr + min(d, D - 4));
                    if (d >= D) v = make_float4(0.f, 0.f, 0.f, 0.f);
                    Cr[i][h] = v;
                }
            }
            if constexpr (BIG) {
                const int a2 = min(PSET2 + PSL * ap + min(sl, PSL - 1), alen - 1);          // (clamped: only read by events longer than 129 segments)
                const float* pr2 = P.PALL + (row0 + a2) * PH + 32 * lr;
#pragma unroll
                for (int h = 0; h < 8; ++h) Pr2[i][h] = *reinterpret_cast<const float4*>(pr2 + 4 * h);
            }
        }
    } else {
#pragma unroll
        for (int i = 0; i < PSG; ++i)
#pragma unroll
            for (int h = 0; h < 8; ++h) {
                Pr[i][h] = make_float4(0.f, 0.f, 0.f, 0.f);
                if constexpr (!BIG) Cr[i][h] = make_float4(0.f, 0.f, 0.f, 0.f);
                else Pr2[i][h] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
    }
    sst_(1);
    if (H2 && P.pimg && (is_g1 || is_qw)) {
        // images prebuilt by persist_fwd_prebuild (same code, same bits): a straight copy of 128 / 64 KB from L2
        const float* src = P.pimg + (is_g1 ? (long)lb * PIMG_G1 : (long)HG1 * PIMG_G1 + (long)(lb - HG1) * PIMG_Q);
        const int n4 = is_g1 ? 8192 : 4096;
        const float4* s4 = reinterpret_cast<const float4*>(src);
#pragma unroll 8
        for (int i = tid; i < n4; i += 256) wimg[i] = s4[i];
        if (tid < 64) {
            const float v = src[4 * n4 + tid];
            if (tid < 32) invbA[tid] = v; else if (is_g1) invbC[tid - 32] = v;
        }
        sst_(2);
        sst_(3);
        __syncthreads();
    } else if (is_g1) {
        auto row = [&](int cc) { return (cc >> 3) * PH + 8 * lb + (cc & 7); };       // tile column cc = gate * 8 + unit
        if (H2) {
            fill_bimg_h2(wimg, invbA, scr, P.w_hh1, PH, PH, 1, row, tid);
            sst_(2);
            fill_bimg_h2(wimg + 4096, invbC, scr, P.w_att, P.ld_att, D, 1, row, tid);
            sst_(3);
        } else {
            fill_bimg32(wimg, P.w_hh1, PH, PH, row, tid);
            fill_bimg32(wimg + 4096, P.w_att, P.ld_att, D, row, tid);
        }
    } else if (is_qw) {
        auto row = [&](int cc) { return 32 * (lb - HG1) + cc; };
        if (H2) fill_bimg_h2(wimg, invbA, scr, P.w_h2a, PH, PH, 1, row, tid);
        else fill_bimg32(wimg, P.w_h2a, PH, PH, row, tid);
    }
    auto exp2x = [](const float4 pv) {
        return make_float4(__expf(2.f * fminf(fmaxf(pv.x, -43.f), 43.f)), __expf(2.f * fminf(fmaxf(pv.y, -43.f), 43.f)),
                           __expf(2.f * fminf(fmaxf(pv.z, -43.f), 43.f)), __expf(2.f * fminf(fmaxf(pv.w, -43.f), 43.f)));
    };
    if (att_live) {
#pragma unroll
        for (int i = 0; i < PSG; ++i)
#pragma unroll
            for (int h = 0; h < 8; ++h) {
                Pr[i][h] = exp2x(Pr[i][h]);
                if constexpr (BIG) Pr2[i][h] = exp2x(Pr2[i][h]);
            }
    }
    if (H2 && att_live) {
        // bound on |context| of this event = max |C3D| over its slots: one atomic max per workgroup (non-negative floats order like uints);
        // complete for every reader by the first context hand-off
        float mx = 0.f;
        if constexpr (!BIG) {
#pragma unroll
            for (int i = 0; i < PSG; ++i)
#pragma unroll
                for (int h = 0; h < 8; ++h) mx = fmaxf(fmaxf(mx, fmaxf(fabsf(Cr[i][h].x), fabsf(Cr[i][h].y))), fmaxf(fabsf(Cr[i][h].z), fabsf(Cr[i][h].w)));
        } else {
            for (int set = 0; set < (alen > PSET2 ? 2 : 1); ++set) {
#pragma unroll
                for (int i = 0; i < PSG; ++i) {
                    const int sl = grow_ + 16 * i;
                    const int a = min(PSET2 * set + PSL * ap + min(sl, PSL - 1), alen - 1);
                    const float* cr = P.c3d + (row0 + a) * D;
#pragma unroll
                    for (int h = 0; h < 8; ++h) {
                        const int d = 32 * lr + 4 * h;
                        const float4 v = *reinterpret_cast<const float4*>(cr + min(d, D - 4));
                        if (d < D) mx = fmaxf(fmaxf(mx, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
                    }
                }
            }
        }
        mx = wave_max(mx);
        if (lane == 0) atomicMax(reinterpret_cast<unsigned*>(P.XCMAX) + an, __float_as_uint(mx));
    }
    sst_(4);
    __syncthreads();
    const bool has2 = BIG && att_live && alen > PSET2;           // uniform over the workgroup

    const int gr = tid >> 3, gu = tid & 7, gn = HR * m + gr;      // gate-math ownership: thread (row gr of the half, unit gu)
    float c1 = 0.f;
    const u32 XBH = HR * PH * 4;                                  // bytes of one half's [32 x 512] exchange operand
    const long XHALF = (long)HR * PH;                             // floats
    const int srole = b == 0 ? 0 : (b == HG1 ? 1 : (b == HG1 + HQ ? 2 : -1));
    if (P.stamps && b == 0 && tid == 0) P.stamps[14] = __builtin_amdgcn_s_memrealtime();          // set-up done

    for (int t = 0; t < S; ++t) {
        if constexpr (BIG) {
            // every register counts here (both e^{2p} sets are resident): what is cheap to re-derive from the lane's coordinates -- row
            // offsets, validity predicates, exchange addresses -- must not be hoisted out of the step loop and kept (the compiler then
            // spills them and reloads from scratch every step); redefining the coordinates per step makes their dependants loop-variant
            asm volatile("" : "+v"(grow_), "+v"(lr), "+v"(alen), "+v"(row0));
        }
        if (srole >= 0) STAMP(srole, 0);
        f32x16 acc;
#pragma unroll
        for (int g = 0; g < 16; ++g) acc[g] = 0.f;
        float pre[4] = {0.f, 0.f, 0.f, 0.f};
        float mh1 = 1.f, mo1 = 1.f;
        if (is_g1) {
            if (!SAMP) {
                const float* grow = P.GATES1 + ((long)t * N + min(gn, N - 1)) * 4 * PH + 8 * lb + gu;
#pragma unroll
                for (int g = 0; g < 4; ++g) pre[g] = grow[g * PH];
            }
            mh1 = mask_h(P.dh, gn, 8 * lb + gu, 1, t);
            mo1 = mask_o(P.dout, gn, 8 * lb + gu, 1, t);
        }
        // ---- phase A: W_hh1 . h1(t-1) (gate workgroups) / q = W_h2a . h1(t-1) + b (q workgroups) ----
        // h2: every load of a hand-off operand shares one chip-wide path (measured: 160 workgroups x 64 KB take 4.4 us, the 32 q workgroups alone
        // 1.35 us, copies at other addresses or staggered readers change nothing), so the gate workgroups, whose product is not needed before the
        // context arrives, fetch h1 behind their attention role instead of beside the q workgroups
        const __amdgpu_buffer_rsrc_t rh = mk_rsrc(P.XH1 + ((long)max(t - 1, 0) * 2 + m) * XHALF, XBH);
        float4 ha[H2 ? 8 : 1][2];
        auto h1_fetch = [&]() {
            // two fp16 planes [plane][k / 8][32 rows][8 halves]: lane (r, kh), k step s -> 16 bytes per plane
#pragma unroll
            for (int s_ = 0; s_ < (H2 ? 8 : 1); ++s_)
#pragma unroll
                for (int pl = 0; pl < 2; ++pl)
                    ha[s_][pl] = ld16_bulk(rh, (u32)(((pl * 64 + 16 * w + 2 * s_ + (lane >> 5)) * HR + (lane & 31)) * 16));
        };
        auto h1_product = [&]() {
#pragma unroll
            for (int s_ = 0; s_ < (H2 ? 8 : 1); ++s_) {
                const float4* bp = wimg + (w * 8 + s_) * 128 + lane;
                const f16x8p bh = as_f16x8(bp[0]), bl = as_f16x8(bp[64]), ah = as_f16x8(ha[s_][0]), al = as_f16x8(ha[s_][1]);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0);
            }
            const float ca = H2_INV_SA * invbA[lane & 31];          // back to true scale (column = lane & 31)
#pragma unroll
            for (int g = 0; g < 16; ++g) acc[g] *= ca;
        };
        if (((is_g1 && !H2) || is_qw) && t > 0) {
            if (!wait_total(P, cnt(C_H1, t - 1), HG1, flag, 100000u + t)) return;
            if (srole >= 0) STAMP(srole, 1);
            if (H2) {
                h1_fetch();
                // all 16 fragment loads ahead of the first MFMA: left alone the scheduler interleaves them with the MFMA groups (four in flight:
                // 1.5 us; in the BIG instantiation, short of registers, ONE per group: 16 dependent round trips, 4.3 us; all of them: 1.25 us)
                __builtin_amdgcn_sched_barrier(0);
                h1_product();
            } else {
                float4 a[16];
                load_afrag32(a, rh, w, lane);
                mfma_tile32(acc, a, wimg + w * 1024, lane);
            }
            if (srole >= 0) STAMP(srole, 2);
        }
        if (is_qw) {
            const int cq = lb - HG1, c4 = 4 * gu;                 // thread: row gr, columns 32 cq + c4 .. +3
            float4 qv = *reinterpret_cast<const float4*>(P.b_h2a + 32 * cq + c4);
            if (t > 0) {
                acc_to_lds32(acc, red, w, lane);
                __syncthreads();
                const float* rp = red + gr * 32 + c4;
#pragma unroll
                for (int ww = 0; ww < 4; ++ww) {
                    const float4 v = *reinterpret_cast<const float4*>(rp + ww * HR * 32);
                    qv.x += v.x; qv.y += v.y; qv.z += v.z; qv.w += v.w;
                }
            }
            st16_sc1(mk_rsrc(P.XQ + ((long)t * 2 + m) * XHALF, XBH), (u32)((((4 * cq + (c4 >> 3)) * HR + gr) * 8 + (c4 & 7)) * 4), qv);
            if (srole >= 0) STAMP(srole, 3);
            publish(cnt(C_Q, t));
            if (srole >= 0) STAMP(srole, 4);
            if (!SAMP && gn < N) *reinterpret_cast<float4*>(P.QS + ((long)t * N + gn) * PH + 32 * cq + c4) = qv;
        }
        // ---- attention: scores, (split) softmax, context partial ----
        {
            // BIG: row buffers of the C3D stream (slot group g of 0..5 = set g / 3, rows grow_ + 16 (g % 3))
            float4 cbuf[BIG ? NCB : 1][8];
            // (buffer loads over the whole feature tensor: one 32-bit offset per row + immediates -- with 64-bit global addresses the
            // compiler hoists 96 loop-invariant address registers out of the step loop and spills; reads past a row's end land in the next row
            // and are masked at use, reads past the tensor's end return zero)
            const __amdgpu_buffer_rsrc_t rc3 = mk_rsrc(P.c3d, P.c3d_bytes);
            auto fetch_c = [&](int g, float4 (&dst)[8]) {
                const int sl = grow_ + 16 * (g % PSG);
                const int a = min(PSET2 * (g / PSG) + PSL * ap + min(sl, PSL - 1), alen - 1);
                const u32 off = (u32)(((row0 + a) * D + 32 * lr) * 4);
#pragma unroll
                for (int h = 0; h < 8; ++h) dst[h] = ld16_plain(rc3, off + 16 * h);
            };
            const int ng = has2 ? 2 * PSG : PSG;          // slot groups of this workgroup (uniform)
            if (!wait_total(P, cnt(C_Q, t), HQ, flag, 200000u + t)) return;
            if (srole >= 0) STAMP(srole, 5);
            if (att_live) {
                const __amdgpu_buffer_rsrc_t rq = mk_rsrc(P.XQ + ((long)t * 2 + m) * XHALF, XBH);
                float4 q[8];
#pragma unroll
                for (int h = 0; h < 8; ++h) q[h] = ld16_sc1(rq, (u32)((((4 * lr + (h >> 1)) * HR + ar) * 8 + 4 * (h & 1)) * 4));
                if constexpr (BIG) {
                    // the first C3D rows are requested BEHIND the q loads: they are not needed before every score is formed (pure register work),
                    // which covers their latency.  (Requested ahead of the wait for q they land earlier still, but they then share the
                    // hand-off path with the q workgroups' h1 fetch, which is the critical one: q product 1.5 -> 4.6 us.)
#pragma unroll
                    for (int g = 0; g < NCB; ++g) fetch_c(g, cbuf[g]);
                }
                float asum = 0.f;
#pragma unroll
                for (int h = 0; h < 8; ++h) {
                    q[h] = make_float4(__expf(2.f * fminf(fmaxf(q[h].x, -43.f), 43.f)), __expf(2.f * fminf(fmaxf(q[h].y, -43.f), 43.f)),
                                       __expf(2.f * fminf(fmaxf(q[h].z, -43.f), 43.f)), __expf(2.f * fminf(fmaxf(q[h].w, -43.f), 43.f)));
                    const float4 a4 = *reinterpret_cast<const float4*>(sal + 32 * lr + 4 * h);
                    asum += (a4.x + a4.y) + (a4.z + a4.w);
                }
                float e[PSG];
#pragma unroll
                for (int i = 0; i < PSG; ++i) {
                    float v = 0.f;
#pragma unroll
                    for (int h = 0; h < 8; ++h) {
                        const float4 a4 = *reinterpret_cast<const float4*>(sal + 32 * lr + 4 * h);
                        v += a4.x * __builtin_amdgcn_rcpf(fmaf(Pr[i][h].x, q[h].x, 1.f)) + a4.y * __builtin_amdgcn_rcpf(fmaf(Pr[i][h].y, q[h].y, 1.f)) +
                             a4.z * __builtin_amdgcn_rcpf(fmaf(Pr[i][h].z, q[h].z, 1.f)) + a4.w * __builtin_amdgcn_rcpf(fmaf(Pr[i][h].w, q[h].w, 1.f));
                    }
                    v = row16_sum(fmaf(-2.f, v, asum));
                    const int sl = grow_ + 16 * i;
                    const bool valid = sl < PSL && PSL * ap + sl < alen;
                    e[i] = valid ? v : -INFINITY;
                }
                // second slot set (BIG, events longer than 129 segments): the same scores from its own e^{2p} registers
                float e2[PSG] = {-INFINITY, -INFINITY, -INFINITY};
                if constexpr (BIG) {
                    if (has2) {
#pragma unroll
                        for (int i = 0; i < PSG; ++i) {
                            const int sl = grow_ + 16 * i;
                            float v = 0.f;
#pragma unroll
                            for (int h = 0; h < 8; ++h) {
                                const float4 a4 = *reinterpret_cast<const float4*>(sal + 32 * lr + 4 * h);
                                v += a4.x * __builtin_amdgcn_rcpf(fmaf(Pr2[i][h].x, q[h].x, 1.f)) + a4.y * __builtin_amdgcn_rcpf(fmaf(Pr2[i][h].y, q[h].y, 1.f)) +
                                     a4.z * __builtin_amdgcn_rcpf(fmaf(Pr2[i][h].z, q[h].z, 1.f)) + a4.w * __builtin_amdgcn_rcpf(fmaf(Pr2[i][h].w, q[h].w, 1.f));
                            }
                            v = row16_sum(fmaf(-2.f, v, asum));
                            const bool valid = sl < PSL && PSET2 + PSL * ap + sl < alen;
                            e2[i] = valid ? v : -INFINITY;
                        }
                    }
                }
                float shift = 0.f;
                if (use_max) {       // exact max-shifted softmax: the event's three workgroups exchange their local maxima (8-byte granules)
                    const float mloc = fmaxf(fmaxf(e[0], fmaxf(e[1], e[2])), fmaxf(e2[0], fmaxf(e2[1], e2[2])));
                    if (lr == 0) sx[16 + grow_] = mloc;
                    __syncthreads();
                    unsigned long long* gr_ = P.GRAN + ((long)t * PROWS + an) * 3;
                    if (tid < 64) {
                        float m16 = lane < 16 ? sx[16 + lane] : -INFINITY;
                        m16 = wave_max(m16);
                        if (lane == 0)
                            __hip_atomic_store(gr_ + ap, ((unsigned long long)(t + 1) << 32) | __float_as_uint(m16), __ATOMIC_RELAXED, ECHR_AGENT);
                        float mm = -INFINITY;
                        u32 spins = 0;
                        for (;;) {
                            unsigned long long x = lane < 3 ? __hip_atomic_load(gr_ + lane, __ATOMIC_RELAXED, ECHR_AGENT) : ((unsigned long long)(t + 1) << 32) | 0xff800000u;
                            const bool ok = (u32)(x >> 32) == (u32)(t + 1);
                            if (__all(ok)) { mm = __uint_as_float((u32)x); break; }
                            if ((++spins & 31) == 0) {
                                if (__hip_atomic_load(P.abort_word, __ATOMIC_RELAXED, ECHR_AGENT)) break;
                                if (spins > P.spin_limit) {      // timed out: abort the launch like every other bounded spin (shift = 0 below is never consumed)
                                    if (lane == 0) {
                                        __hip_atomic_store(P.abort_word, 9000u + (u32)t, __ATOMIC_RELAXED, ECHR_AGENT);
                                        __hip_atomic_store(P.host_flag, 9000u + (u32)t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                                    }
                                    break;
                                }
                            }
                            // (no s_sleep between polls: a poll is one dependent round trip, and 64 idle clocks per poll measured +8 us per iteration -- round 6)
                        }
                        mm = wave_max(mm);
                        if (lane == 0) sx[32] = mm;
                    }
                    __syncthreads();
                    shift = sx[32];
                    if (!(shift > -INFINITY)) shift = 0.f;
                }
                float ssum = 0.f;
                float4 cx[8];
#pragma unroll
                for (int h = 0; h < 8; ++h) cx[h] = make_float4(0.f, 0.f, 0.f, 0.f);
                if constexpr (!BIG) {
#pragma unroll
                    for (int i = 0; i < PSG; ++i) {
                        const float x = __expf(e[i] - shift);          // exp(-inf) = 0 for slots past the event's end
                        e[i] = x;
                        ssum += x;
#pragma unroll
                        for (int h = 0; h < 8; ++h) {
                            cx[h].x += x * Cr[i][h].x; cx[h].y += x * Cr[i][h].y; cx[h].z += x * Cr[i][h].z; cx[h].w += x * Cr[i][h].w;
                        }
                    }
                } else {
#pragma unroll
                    for (int g = 0; g < 2 * PSG; ++g) {
                        if (g < ng) {
                            float& eg = g < PSG ? e[g % PSG] : e2[g % PSG];
                            const float x = __expf(eg - shift);
                            eg = x;
                            ssum += x;
#pragma unroll
                            for (int h = 0; h < 8; ++h) {
                                float4 c4 = cbuf[g % NCB][h];
                                if (32 * lr + 4 * h >= D) c4 = make_float4(0.f, 0.f, 0.f, 0.f);
                                cx[h].x += x * c4.x; cx[h].y += x * c4.y; cx[h].z += x * c4.z; cx[h].w += x * c4.w;
                            }
                            if (g + NCB < ng) fetch_c(g + NCB, cbuf[g % NCB]);          // the buffer just consumed takes the row NCB groups ahead
                        }
                    }
                }
                if (lr == 0) sx[grow_] = ssum;
                if (!SAMP) {
                    const float xw = lr == 0 ? e[0] : (lr == 1 ? e[1] : e[2]);
                    const int sl = grow_ + 16 * lr;
                    if (lr < PSG && sl < PSL && PSL * ap + sl < alen) st4_sc1(P.WU + ((long)t * PROWS + an) * WU_LD + PSL * ap + sl, xw);
                    if (has2) {          // lanes 3..5 of the DPP row store the second set's unnormalised weights
                        const float xw2 = lr == 3 ? e2[0] : (lr == 4 ? e2[1] : e2[2]);
                        const int sl2 = grow_ + 16 * (lr - 3);
                        if (lr >= 3 && lr < 3 + PSG && sl2 < PSL && PSET2 + PSL * ap + sl2 < alen)
                            st4_sc1(P.WU + ((long)t * PROWS + an) * WU_LD + PSET2 + PSL * ap + sl2, xw2);
                    }
                }
                // cross-row sum of the context partials through 16 KB of LDS: DPP rows 0-7, then rows 8-15 (LDS float atomics into one
                // vector were tried: 4-way same-address ds_add_f32 made this phase 3x slower)
                // the wave's 4 rows are summed in registers (permlane swaps), the 4 waves through 8 KB of LDS
                float csum[2] = {0.f, 0.f};
#pragma unroll
                for (int h = 0; h < 8; ++h) cx[h] = make_float4(xrow_sum4(cx[h].x), xrow_sum4(cx[h].y), xrow_sum4(cx[h].z), xrow_sum4(cx[h].w));
                if ((lane >> 4) == 0) {
#pragma unroll
                    for (int h = 0; h < 8; ++h) *reinterpret_cast<float4*>(red + w * PH + 32 * lr + 4 * h) = cx[h];
                }
                __syncthreads();
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int d = tid + 256 * h;
                    csum[h] = (red[d] + red[PH + d]) + (red[2 * PH + d] + red[3 * PH + d]);
                }
                __syncthreads();
                if (SAMP) {
                    // decoding: one slot per workgroup of the event, plain stores; the readers add the three slots in slot order (bitwise reproducible)
                    float* xc = Q->XC3 + (((long)t * 2 + m) * 3 + ap) * XHALF;
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const int d = tid + 256 * h;
                        st4_sc1(xc + ((d >> 3) * HR + ar) * 8 + (d & 7), d < D ? csum[h] : 0.f);
                    }
                    if (tid == 0) {
                        float sum = 0.f;
#pragma unroll
                        for (int g = 0; g < 16; ++g) sum += sx[g];
                        st4_sc1(Q->XS3 + ((long)t * PROWS + an) * 3 + ap, sum);
                    }
                } else {
                float* xc = P.XC + ((long)t * 2 + m) * XHALF;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int d = tid + 256 * h;
                    if (d < D) atomicAdd(xc + ((d >> 3) * HR + ar) * 8 + (d & 7), csum[h]);
                }
                if (tid == 0) {
                    float sum = 0.f;
#pragma unroll
                    for (int g = 0; g < 16; ++g) sum += sx[g];
                    atomicAdd(P.XS + (long)t * PROWS + an, sum);
                }
                }
            }
            if (srole >= 0) STAMP(srole, 6);
            if (!BIG && H2 && is_g1 && t > 0) {
                // the gate workgroups' h1(t-1) fetch (16 loads per lane) goes out behind the context atomics and stays in flight across
                // the publish (q(t) complete => the q workgroups have consumed all of h1(t-1), so it is complete and visible here too)
                h1_fetch();
                publish_keep<16>(cnt(C_C, t));
            } else {
                publish(cnt(C_C, t));
            }
            if (srole >= 0) STAMP(srole, 7);
        }
        if (H2 && is_g1 && t > 0) {
            if (BIG) {                    // BIG: fetched here, not across the context publish -- the 64 fragment registers hold the second e^{2p} set
                h1_fetch();
                __builtin_amdgcn_sched_barrier(0);
            }
            h1_product();
        }
        // ---- phase C: attended-context columns + gate math; the new h1 goes to the next step ----
        if (is_g1) {
            if (!wait_total(P, cnt(C_C, t), HWG, flag, 300000u + t)) return;
            if (srole >= 0) STAMP(srole, 8);
            f32x16 accc;
#pragma unroll
            for (int g = 0; g < 16; ++g) accc[g] = 0.f;
            float4 av_pre = make_float4(0.f, 0.f, 0.f, 0.f);          // (teacher-forced h2 form) this lane's piece of the context, stored behind the hand-off
            float ssum_l = 1.f;
            {
                const __amdgpu_buffer_rsrc_t rc = mk_rsrc(P.XC + ((long)t * 2 + m) * XHALF, XBH);
                float4 a[H2 ? 1 : 16];
                // teacher-forced h2 form: the 16 context fragments of this wave are requested FIRST, so that their round trip overlaps the
                // table phase (a dependent load + a barrier) and the saved-context store below instead of following them (context product
                // 3.6 -> 2.x us per step; not in the BIG instantiation, whose registers hold the second e^{2p} set)
                constexpr bool CPRE = H2 && !SAMP && !BIG;
                float4 cv[CPRE ? 8 : 1][2];
                if constexpr (CPRE) {
#pragma unroll
                    for (int s_ = 0; s_ < 8; ++s_) {
                        const u32 off = (u32)((((16 * w + 2 * s_ + (lane >> 5)) * HR + (lane & 31)) * 8) * 4);
                        cv[s_][0] = ld16_bulk(rc, off); cv[s_][1] = ld16_bulk(rc, off + 16);
                    }
                    // ... and with them the sum of exponentials of this lane's row: the conversion factor is formed per lane, so only the first
                    // step (where the bounds on |context| become known) goes through the table phase and its barrier
                    ssum_l = ld4_sc1(P.XS + (long)t * PROWS + HR * m + (lane & 31));
                    // the piece of the context this wave saves for the backward pass rides along; its store waits until h1 has been handed on
                    if (8 * lb < D && w == (lb >> 4)) av_pre = ld16_sc1(rc, (u32)(((lb * HR + (lane & 31)) * 8 + 4 * (lane >> 5)) * 4));
                }
                if (H2 && (!CPRE || t == 0)) {
                    // per-row tables: conversion factor 2^(12 - e_r) / s_r (e_r from the bound on |context|) and its inverse scale
                    if (tid < HR) {
                        if (t == 0) cmx[tid] = ld4_sc1(P.XCMAX + HR * m + tid);
                        const int ex = (int)((__float_as_uint(fmaxf(cmx[tid], 1e-30f)) >> 23) & 0xFFu) - 127 + 1;      // |context| < 2^ex
                        float ssum;
                        if (SAMP) {
                            const float* xs3 = Q->XS3 + ((long)t * PROWS + HR * m + tid) * 3;
                            ssum = (ld4_sc1(xs3) + ld4_sc1(xs3 + 1)) + ld4_sc1(xs3 + 2);
                        } else ssum = ld4_sc1(P.XS + (long)t * PROWS + HR * m + tid);
                        sCt[tid] = ldexpf(1.f, 12 - ex) / ssum;
                        invAt[tid] = ldexpf(1.f, ex - 12);
                    }
                    __syncthreads();
                } else if (!H2) {
                    load_afrag32(reinterpret_cast<float4(&)[16]>(a), rc, w, lane);
                }
                // normalised context, saved for backward: this workgroup stores features [8 lb, 8 lb + 8) (wave lb / 16, chunk lb % 16)
                if (!SAMP && !CPRE && 8 * lb < D && w == (lb >> 4)) {
                    float4 av = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (H2) av = ld16_sc1(rc, (u32)(((lb * HR + (lane & 31)) * 8 + 4 * (lane >> 5)) * 4));
                    else {
#pragma unroll
                        for (int c = 0; c < (H2 ? 1 : 16); ++c)
                            if (c == (lb & 15)) av = a[c];
                    }
                    const int n = HR * m + (lane & 31), d0 = 8 * lb + 4 * (lane >> 5);
                    if (n < N && d0 < D) {
                        const float is = 1.0f / (CPRE ? ssum_l : ld4_sc1(P.XS + (long)t * PROWS + n));
                        *reinterpret_cast<float4*>(P.ATT + ((long)t * N + n) * D + d0) = make_float4(av.x * is, av.y * is, av.z * is, av.w * is);
                    }
                }
                if (H2) {
                    // fp32 context -> fp16 pair fragments on the fly: chunk c = 2 s + kh' holds k = 128 w + 8 c + 4 kh .. ; the 32x32x16 A
                    // fragment of k step s wants lane (r, kh): k = 128 w + 16 s + 8 kh + j, j < 8 = the two float4 of chunk 2 s + kh held by
                    // lanes (r, 0) and (r, 1) -> re-read them in that shape from the exchange buffer instead: 8 consecutive floats per lane
                    float f;
                    if constexpr (CPRE) {
                        const int ex = (int)((__float_as_uint(fmaxf(cmx[lane & 31], 1e-30f)) >> 23) & 0xFFu) - 127 + 1;
                        f = ldexpf(1.f, 12 - ex) / ssum_l;
                    } else f = sCt[lane & 31];
#pragma unroll
                    for (int s_ = 0; s_ < 8; ++s_) {
                        const u32 off = (u32)((((16 * w + 2 * s_ + (lane >> 5)) * HR + (lane & 31)) * 8) * 4);
                        float4 v0, v1;
                        if (SAMP) {
                            // the event's three partial contexts, added in slot order
                            const __amdgpu_buffer_rsrc_t r3 = mk_rsrc(Q->XC3 + ((long)t * 2 + m) * 3 * XHALF, 3 * XBH);
                            const float4 a0 = ld16_bulk(r3, off), a1 = ld16_bulk(r3, off + 16), b0 = ld16_bulk(r3, off + XBH), b1 = ld16_bulk(r3, off + XBH + 16),
                                         c0 = ld16_bulk(r3, off + 2 * XBH), c1 = ld16_bulk(r3, off + 2 * XBH + 16);
                            v0 = make_float4((a0.x + b0.x) + c0.x, (a0.y + b0.y) + c0.y, (a0.z + b0.z) + c0.z, (a0.w + b0.w) + c0.w);
                            v1 = make_float4((a1.x + b1.x) + c1.x, (a1.y + b1.y) + c1.y, (a1.z + b1.z) + c1.z, (a1.w + b1.w) + c1.w);
                        } else if constexpr (CPRE) { v0 = cv[s_][0]; v1 = cv[s_][1]; }
                        else { v0 = ld16_bulk(rc, off); v1 = ld16_bulk(rc, off + 16); }
                        const float x[8] = {v0.x * f, v0.y * f, v0.z * f, v0.w * f, v1.x * f, v1.y * f, v1.z * f, v1.w * f};
                        unsigned hw[8], lw[8];
#pragma unroll
                        for (int j = 0; j < 8; ++j) { unsigned short hi, lo; split_h2(x[j], hi, lo); hw[j] = hi; lw[j] = lo; }
                        const uint4 uh = make_uint4(hw[0] | (hw[1] << 16), hw[2] | (hw[3] << 16), hw[4] | (hw[5] << 16), hw[6] | (hw[7] << 16));
                        const uint4 ul = make_uint4(lw[0] | (lw[1] << 16), lw[2] | (lw[3] << 16), lw[4] | (lw[5] << 16), lw[6] | (lw[7] << 16));
                        const f16x8p ah = __builtin_bit_cast(f16x8p, uh), al = __builtin_bit_cast(f16x8p, ul);
                        const float4* bp = wimg + 4096 + (w * 8 + s_) * 128 + lane;
                        const f16x8p bh = as_f16x8(bp[0]), bl = as_f16x8(bp[64]);
                        accc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, accc, 0, 0, 0);
                        accc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, accc, 0, 0, 0);
                        accc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, accc, 0, 0, 0);
                    }
                    const float cb = invbC[lane & 31];
#pragma unroll
                    for (int g = 0; g < 16; ++g) acc[g] += accc[g] * (cb * invAt[(g & 3) + 8 * (g >> 2) + 4 * (lane >> 5)]);
                } else {
                    // sums of exponentials of the rows this lane's accumulator registers belong to (row = (reg & 3) + 8 (reg >> 2) + 4 (l >> 5))
                    float4 s4[4];
                    const __amdgpu_buffer_rsrc_t rsum = mk_rsrc(P.XS + (long)t * PROWS + HR * m, HR * 4);
#pragma unroll
                    for (int g = 0; g < 4; ++g) s4[g] = ld16_sc1(rsum, (u32)((8 * g + 4 * (lane >> 5)) * 4));
                    mfma_tile32(accc, reinterpret_cast<float4(&)[16]>(a), wimg + 4096 + w * 1024, lane);
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        acc[4 * g + 0] += accc[4 * g + 0] / s4[g].x; acc[4 * g + 1] += accc[4 * g + 1] / s4[g].y;
                        acc[4 * g + 2] += accc[4 * g + 2] / s4[g].z; acc[4 * g + 3] += accc[4 * g + 3] / s4[g].w;
                    }
                }
            }
            if (srole >= 0) STAMP(srole, 9);
            if (SAMP) {
                // token-side gate pre-activations: row `token` of stream 1's table.  The token (arg-max of step t - 1's logits) is the last
                // thing this step waits for: everything above depends on h1(t - 1) only
                u32 tok = 0;
                if (t > 0) {
                    if (!wait_total(P, Q->cnt_tok + (long)(t - 1) * CNT_LINE, LWG, flag, 400000u + t)) return;
                    tok = 0xFFFFFFFFu - (u32)__hip_atomic_load(Q->KEY + (long)(t - 1) * PROWS + min(gn, N - 1), __ATOMIC_RELAXED, ECHR_AGENT);
                    tok = min(tok, (u32)(Q->V1 - 1));
                }
                const float* grow = Q->TG[1] + (long)tok * 4 * PH + 8 * lb + gu;
#pragma unroll
                for (int g = 0; g < 4; ++g) pre[g] = grow[g * PH];
                if (srole >= 0) STAMP(srole, 12);
            }
            acc_to_lds32(acc, red, w, lane);
            __syncthreads();
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int o = gr * 32 + 8 * g + gu;
                pre[g] += red[o] + red[HR * 32 + o] + red[2 * HR * 32 + o] + red[3 * HR * 32 + o];
            }
            const CellOut co = lstm_cell(pre[0], pre[1], pre[2], pre[3], c1, mh1, mo1);
            c1 = co.c;
            if (H2) {
                // fp16 planes [plane][k / 8][32 rows][8 halves]: this workgroup's 8 units are group k / 8 = lb; staged through LDS so that
                // 64 threads store one 16-byte piece each
                __syncthreads();                        // every thread has read its tile sums
                unsigned short hi, lo;
                split_h2(co.h * H2_SA, hi, lo);
                unsigned short* sh = reinterpret_cast<unsigned short*>(red);
                sh[(0 * HR + gr) * 8 + gu] = hi;
                sh[(1 * HR + gr) * 8 + gu] = lo;
                __syncthreads();
                {
                    const int pl = lane >> 5, row = lane & 31;
                    if (w == 0) st16_sc1(mk_rsrc(P.XH1 + ((long)t * 2 + m) * XHALF, XBH), (u32)(((pl * 64 + lb) * HR + row) * 16), reinterpret_cast<const float4*>(red)[pl * HR + row]);
                }
            } else {
                st4_sc1(P.XH1 + ((long)t * 2 + m) * XHALF + (lb * HR + gr) * 8 + gu, co.h);
            }
            if (srole >= 0) STAMP(srole, 10);
            publish(cnt(C_H1, t));          // (its barrier also protects `red` for the next step)
            if (srole >= 0) STAMP(srole, 11);
            if (SAMP) continue;             // decoding keeps no activations
            if constexpr (H2 && !SAMP && !BIG) {
                // normalised context, saved for backward: this workgroup stores features [8 lb, 8 lb + 8) (wave lb / 16, chunk lb % 16)
                const int n = HR * m + (lane & 31), d0 = 8 * lb + 4 * (lane >> 5);
                if (8 * lb < D && w == (lb >> 4) && n < N && d0 < D) {
                    const float is = 1.0f / ssum_l;
                    *reinterpret_cast<float4*>(P.ATT + ((long)t * N + n) * D + d0) = make_float4(av_pre.x * is, av_pre.y * is, av_pre.z * is, av_pre.w * is);
                }
            }
            if (gn < N) {
                const int j = 8 * lb + gu;
                float* go = P.GATES1 + ((long)t * N + gn) * 4 * PH + j;
                // (saved for the backward pass, read ~0.4 ms from now: non-temporal, they do not displace the hand-off operands in L2)
                if (P.nt_saved) {
                    __builtin_nontemporal_store(co.gi, go); __builtin_nontemporal_store(co.gf, go + PH);
                    __builtin_nontemporal_store(co.gg, go + 2 * PH); __builtin_nontemporal_store(co.go, go + 3 * PH);
                    __builtin_nontemporal_store(co.c, P.CS1 + ((long)(t + 1) * N + gn) * PH + j);
                } else {
                go[0] = co.gi; go[PH] = co.gf; go[2 * PH] = co.gg; go[3 * PH] = co.go;
                P.CS1[((long)(t + 1) * N + gn) * PH + j] = co.c;
                }
                const long o = ((long)gn * 3 + 1) * PH + j;
                P.HS[(long)(t + 1) * N * 3 * PH + o] = co.h;
                P.OUTD[(long)t * N * 3 * PH + o] = co.hd;
            }
            {   // normalised attention weights: rows of event lb / 2 of this half, slots [65 (lb & 1), +65) -- BIG: [130 (lb & 1), +130)
                constexpr int WSPAN = BIG ? 130 : 65;
                const int n = HR * m + (lb >> 1), a0 = WSPAN * (lb & 1) + tid;
                if (tid < WSPAN && n < N && a0 < P.A) {
                    const int len = P.ev_len[n];
                    float wv = 0.f;
                    if (a0 < len) wv = ld4_sc1(P.WU + ((long)t * PROWS + n) * WU_LD + a0) / ld4_sc1(P.XS + (long)t * PROWS + n);
                    P.WT[((long)t * N + n) * P.A + a0] = wv;
                }
            }
        }
    }
}
template <bool H2>
__global__ __launch_bounds__(256, 1) void dec_persist_att2_kernel(PersistK2 P) { dec_persist_att2_body<H2, false>(P, blockIdx.x); }

// ==========================================================================================================================
// fp16-PAIR ("h2") products inside the persistent forward kernels.  The recurrent activations are bounded (|h| <= dropout scale), so
// h is exchanged as two fp16 planes hi = fp16(h * 2^12), lo = fp16(h * 2^12 - hi) (22+ significant bits, no value ever held unscaled),
// the weight columns live in LDS as two fp16 planes with one power-of-two scale per output column, and a product is three
// v_mfma_f32_32x32x16_f16 (hi.hi' + hi.lo' + lo.hi', each exact in the fp32 accumulator; the dropped lo.lo' is < 2^-22 of a term) --
// the same operand format as gemm.hip's h2 GEMM with the scale block stretched over the whole contraction.  16x the MFMA rate of the
// f32 forms at 3 products: the plain LSTM streams' 8 us of MFMA time per step becomes 1.5 us.
// ==========================================================================================================================
// ---- the two plain LSTM streams, h2 products: 64 rows x 64 gate columns (16 units) per workgroup ----
// h exchange layout (per timestep, 128 KB): [plane 2][k / 8 (64)][row 64][8 halves]
__device__ __forceinline__ void dec_persist_lstm_h2_body(const PersistK& P, const int bid) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    float4* wimg = reinterpret_cast<float4*>(lds);                    // 128 KB
    float* red = reinterpret_cast<float*>(lds + LDS_W);               // 16 KB: [4 waves][32][32] tile partials / staging of the exchanged h
    float* invb = red + 4096;                                         // [64] inverse column scales
    float* scr = invb + 64;                                           // [64] scratch
    int* flag = reinterpret_cast<int*>(lds + LDS_W + LDS_RED + 1024);
    const int b = bid, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const bool is_s0 = b < NS;
    const int N = P.N, S = P.S;
    const int k = is_s0 ? 0 : 2, bs = is_s0 ? b : b - NS, ck = is_s0 ? C_H0 : C_H2;
    float* XH = is_s0 ? P.XH0 : P.XH2;
    auto cnt = [&](int kind, int t) { return P.cnt + ((long)kind * (S + 1) + t) * CNT_LINE; };
    {   // column 32 cb + g * 8 + u8  <->  gate g of unit 16 bs + 8 cb + u8
        auto row = [&](int col) { return ((col >> 3) & 3) * PH + 16 * bs + 8 * (col >> 5) + (col & 7); };
        fill_bimg_h2(wimg, invb, scr, P.w_hh[k], PH, PH, 2, row, tid);
    }
    // gate-math ownership in round (rb, cb): thread -> (row 32 rb + (tid >> 3), unit 16 bs + 8 cb + (tid & 7))
    const int gr = tid >> 3, g8 = tid & 7;
    float cs[4] = {0.f, 0.f, 0.f, 0.f};
    // stream 0: the event part of the gate pre-activations (time-invariant) is added here instead of by a pass over GATES[0] in front of the launch
    float evb[4][4];
#pragma unroll
    for (int rd = 0; rd < 4; ++rd) {
        const int n = 32 * (rd >> 1) + gr, j = 16 * bs + 8 * (rd & 1) + g8;
#pragma unroll
        for (int g = 0; g < 4; ++g) evb[rd][g] = (k == 0 && P.evb0) ? P.evb0[(long)min(n, N - 1) * 4 * PH + g * PH + j] : 0.f;
    }
    const u32 XB = PROWS * PH * 4;
    const bool st_on = b == 0;
    for (int t = 0; t < S; ++t) {
        if (st_on) STAMP(3, 0);
        float pre[4][4], mh[4], mo[4];
#pragma unroll
        for (int rd = 0; rd < 4; ++rd) {
            const int n = 32 * (rd >> 1) + gr, j = 16 * bs + 8 * (rd & 1) + g8;
            const float* grow = P.GATES[k] + ((long)t * N + min(n, N - 1)) * 4 * PH + j;
#pragma unroll
            for (int g = 0; g < 4; ++g) pre[rd][g] = grow[g * PH] + evb[rd][g];
            mh[rd] = mask_h(P.dh, n, j, k, t);
            mo[rd] = mask_o(P.dout, n, j, k, t);
        }
        f32x16 acc[2][2];
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int g = 0; g < 16; ++g) acc[rb][cb][g] = 0.f;
        if (t > 0) {
            if (!wait_total(P, cnt(ck, t - 1), NS, flag, 1000u * (ck + 1) + t)) return;
            if (st_on) STAMP(3, 1);
            // A fragments: lane (r = l & 31, kh = l >> 5): 8 halves of row 32 rb + r at k = 128 w + 16 s + 8 kh
            const __amdgpu_buffer_rsrc_t ra = mk_rsrc(XH + (long)(t - 1) * PROWS * PH, XB);
            float4 a[8][2][2];          // [k step][row block][plane]
#pragma unroll
            for (int s_ = 0; s_ < 8; ++s_)
#pragma unroll
                for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                    for (int pl = 0; pl < 2; ++pl)
                        a[s_][rb][pl] = ld16_bulk(ra, (u32)((((pl * 64 + 16 * w + 2 * s_ + (lane >> 5)) * PROWS) + 32 * rb + (lane & 31)) * 16));
#pragma unroll
            for (int s_ = 0; s_ < 8; ++s_) {
                const float4* bp = wimg + ((long)(w * 8 + s_) * 2) * 2 * 64 + lane;
                f16x8p bh[2], bl[2];
#pragma unroll
                for (int cb = 0; cb < 2; ++cb) { bh[cb] = as_f16x8(bp[(cb * 2) * 64]); bl[cb] = as_f16x8(bp[(cb * 2 + 1) * 64]); }
#pragma unroll
                for (int rb = 0; rb < 2; ++rb) {
                    const f16x8p ah = as_f16x8(a[s_][rb][0]), al = as_f16x8(a[s_][rb][1]);
#pragma unroll
                    for (int cb = 0; cb < 2; ++cb) {
                        acc[rb][cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[cb], acc[rb][cb], 0, 0, 0);
                        acc[rb][cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[cb], acc[rb][cb], 0, 0, 0);
                        acc[rb][cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[cb], acc[rb][cb], 0, 0, 0);
                    }
                }
            }
            if (st_on) STAMP(3, 2);
        }
        CellOut co[4];
#pragma unroll
        for (int rd = 0; rd < 4; ++rd) {
            const int rb = rd >> 1, cb = rd & 1;
            if (t > 0) {
                acc_to_lds32(acc[rb][cb], red, w, lane);
                __syncthreads();
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int o = gr * 32 + 8 * g + g8;
                    pre[rd][g] += (red[o] + red[HR * 32 + o] + red[2 * HR * 32 + o] + red[3 * HR * 32 + o]) * (H2_INV_SA * invb[32 * cb + 8 * g + g8]);
                }
                __syncthreads();
            }
            co[rd] = lstm_cell(pre[rd][0], pre[rd][1], pre[rd][2], pre[rd][3], cs[rd], mh[rd], mo[rd]);
            cs[rd] = co[rd].c;
        }
        // new h as fp16 planes: staged in LDS as [plane][cb][row 64][8 halves], then one 16-byte write-through store per thread
        {
            unsigned short* sh = reinterpret_cast<unsigned short*>(red);
#pragma unroll
            for (int rd = 0; rd < 4; ++rd) {
                unsigned short hi, lo;
                split_h2(co[rd].h * H2_SA, hi, lo);
                const int row = 32 * (rd >> 1) + gr, cb = rd & 1;
                sh[((0 * 2 + cb) * 64 + row) * 8 + g8] = hi;
                sh[((1 * 2 + cb) * 64 + row) * 8 + g8] = lo;
            }
            __syncthreads();
            const int pl = tid >> 7, cb = (tid >> 6) & 1, row = tid & 63;
            const float4 v = reinterpret_cast<const float4*>(red)[(pl * 2 + cb) * 64 + row];
            st16_sc1(mk_rsrc(XH + (long)t * PROWS * PH, XB), (u32)((((pl * 64 + 2 * bs + cb) * PROWS) + row) * 16), v);
        }
        if (st_on) STAMP(3, 3);
        publish(cnt(ck, t));
        if (st_on) STAMP(3, 4);
#pragma unroll
        for (int rd = 0; rd < 4; ++rd) {          // saved activations, off the critical path
            const int n = 32 * (rd >> 1) + gr, j = 16 * bs + 8 * (rd & 1) + g8;
            if (n < N) {
                float* go = P.GATES[k] + ((long)t * N + n) * 4 * PH + j;
                if (P.nt_saved) {
                    __builtin_nontemporal_store(co[rd].gi, go); __builtin_nontemporal_store(co[rd].gf, go + PH);
                    __builtin_nontemporal_store(co[rd].gg, go + 2 * PH); __builtin_nontemporal_store(co[rd].go, go + 3 * PH);
                    __builtin_nontemporal_store(co[rd].c, P.CS[k] + ((long)(t + 1) * N + n) * PH + j);
                } else {
                go[0] = co[rd].gi; go[PH] = co[rd].gf; go[2 * PH] = co[rd].gg; go[3 * PH] = co[rd].go;
                P.CS[k][((long)(t + 1) * N + n) * PH + j] = co[rd].c;
                }
                const long o = ((long)n * 3 + k) * PH + j;
                P.HS[(long)(t + 1) * N * 3 * PH + o] = co[rd].h;
                P.OUTD[(long)t * N * 3 * PH + o] = co[rd].hd;
            }
        }
    }
}
__global__ __launch_bounds__(256, 1) void dec_persist_lstm_h2_kernel(PersistK P) { dec_persist_lstm_h2_body(P, blockIdx.x); }

// ---- decoding: the two plain LSTM streams (as dec_persist_lstm_h2_body) + the logits role (see PersistS) ----
// Per step t a workgroup: waits for token t (arg-max of step t - 1), gathers its gate pre-activations from the stream's table, runs its
// cells and publishes h(t); multiplies the stream's complete h(t) by its W_hh tile (the recurrent part of step t + 1); accumulates its 80
// logit columns over the h0(t) and h2(t) thirds of the contraction; waits for h1(t), adds that third, folds its columns into the events'
// keys and publishes.  v_mfma_f32_16x16x32_f16 on fp16-pair operands: 4 row tiles x 5 column tiles, the wave's k range is a quarter of
// every 512-wide stream block; the four waves' partial tiles are added through LDS, one column tile per round.
typedef float f32x4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void dec_persist_lstm_samp_body(const PersistK& P, const PersistS& Q, const int bid) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    float4* wimg = reinterpret_cast<float4*>(lds);                    // 128 KB
    float* red = reinterpret_cast<float*>(lds + LDS_W);               // 16 KB
    float* invb = red + 4096;
    float* scr = invb + 64;
    int* flag = reinterpret_cast<int*>(lds + LDS_W + LDS_RED + 1024);
    const int b = bid, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const bool is_s0 = b < NS;
    const int N = P.N, S = P.S, V1 = Q.V1;
    const int k = is_s0 ? 0 : 2, bs = is_s0 ? b : b - NS, ck = is_s0 ? C_H0 : C_H2, ck_other = is_s0 ? C_H2 : C_H0;
    float* XH = is_s0 ? P.XH0 : P.XH2;
    auto cnt = [&](int kind, int t) { return P.cnt + ((long)kind * (S + 1) + t) * CNT_LINE; };
    auto cnt2 = [&](int kind, int t, int m) { return Q.cnt2 + (((long)kind * (S + 1) + t) * 2 + m) * CNT_LINE; };
    {
        auto row = [&](int col) { return ((col >> 3) & 3) * PH + 16 * bs + 8 * (col >> 5) + (col & 7); };
        fill_bimg_h2(wimg, invb, scr, P.w_hh[k], PH, PH, 2, row, tid);
    }
    const int gr = tid >> 3, g8 = tid & 7;
    float cs[4] = {0.f, 0.f, 0.f, 0.f};
    float rec[4][4], base[4][4];
#pragma unroll
    for (int rd = 0; rd < 4; ++rd) {
        const int n = 32 * (rd >> 1) + gr, j = 16 * bs + 8 * (rd & 1) + g8;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            rec[rd][g] = 0.f;
            base[rd][g] = k == 0 ? Q.base0[(long)min(n, N - 1) * 4 * PH + g * PH + j] : Q.base2[g * PH + j];
        }
    }
    const u32 XB = PROWS * PH * 4, XBH = HR * PH * 4;
    const int ev = tid >> 2, q4 = tid & 3;                            // epilogue ownership: event, columns 16 c + 4 q4 .. +3 of every tile
    float* lsc = reinterpret_cast<float*>(lds + LDS_W + LDS_RED + 1280);      // [nch][80] column scale back to true units, [nch][80] bias
    float* lbi = lsc + LCOLS * LCHMAX;
    if (tid < LCOLS * Q.nch) {
        const int gv = LCOLS * (b + LWG * (tid / LCOLS)) + tid % LCOLS;
        lsc[tid] = H2_INV_SA * Q.linv[gv];
        lbi[tid] = gv < V1 ? Q.lbias[gv] : 0.f;
    }
    __syncthreads();
    const bool st_on = b == 0;
    const int nhalf = N > HR ? 2 : 1;
    u32 peek_tok = 0;
    int* stopf = flag + 1;
    bool unf[2] = {gr < N, 32 + gr < N};          // rows past the batch count as finished
    for (int t = 0; t < S; ++t) {
        if (st_on) STAMP(3, 0);
        // ---- token t -> gate pre-activations, cells, h(t) ----
        u32 tok[2] = {0u, 0u};
        if (t > 0) {
            if (!wait_peeked2(P, peek_tok, Q.cnt_tok + (long)(t - 1) * CNT_LINE, LWG, Q.cnt_tok + (long)(t - 1) * CNT_LINE, LWG, flag, 400000u + t)) return;
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const u32 x = 0xFFFFFFFFu - (u32)__hip_atomic_load(Q.KEY + (long)(t - 1) * PROWS + min(32 * r + gr, N - 1), __ATOMIC_RELAXED, ECHR_AGENT);
                tok[r] = min(x, (u32)(V1 - 1));
            }
        }
        if (st_on) STAMP(3, 1);
        if (t > 0 && Q.stop_early) {
            // OldModel.sample's stop (:171-180): unfinished &= token > 0; when no event of the group is unfinished any more, nothing of step t
            // or later is ever emitted -- every workgroup of this role sees all 64 tokens and reaches the same verdict; the others leave
            // through their wait loops
            unf[0] = unf[0] && tok[0] > 0u;
            unf[1] = unf[1] && tok[1] > 0u;
            if (tid == 0) *stopf = 0;
            __syncthreads();
            if (unf[0] || unf[1]) *stopf = 1;
            __syncthreads();
            if (*stopf == 0) {
                if (tid == 0) __hip_atomic_store(P.abort_word, STOP_ALL_FINISHED, __ATOMIC_RELAXED, ECHR_AGENT);
                return;
            }
        }
        CellOut co[4];
        {
            float pre[4][4];
#pragma unroll
            for (int rd = 0; rd < 4; ++rd) {
                const int j = 16 * bs + 8 * (rd & 1) + g8;
                const float* grow = Q.TG[k] + (long)tok[rd >> 1] * 4 * PH + j;
#pragma unroll
                for (int g = 0; g < 4; ++g) pre[rd][g] = grow[g * PH];
            }
#pragma unroll
            for (int rd = 0; rd < 4; ++rd) {
                const int n = 32 * (rd >> 1) + gr, j = 16 * bs + 8 * (rd & 1) + g8;
#pragma unroll
                for (int g = 0; g < 4; ++g) pre[rd][g] += base[rd][g] + rec[rd][g];
                co[rd] = lstm_cell(pre[rd][0], pre[rd][1], pre[rd][2], pre[rd][3], cs[rd], mask_h(P.dh, n, j, k, t), mask_o(P.dout, n, j, k, t));
                cs[rd] = co[rd].c;
            }
        }
        {
            unsigned short* sh = reinterpret_cast<unsigned short*>(red);
#pragma unroll
            for (int rd = 0; rd < 4; ++rd) {
                unsigned short hi, lo;
                split_h2(co[rd].h * H2_SA, hi, lo);           // decoding runs without dropout: the logits' input (hd) and the recurrent input (h) coincide
                const int row = 32 * (rd >> 1) + gr, cb = rd & 1;
                sh[((0 * 2 + cb) * 64 + row) * 8 + g8] = hi;
                sh[((1 * 2 + cb) * 64 + row) * 8 + g8] = lo;
            }
            __syncthreads();
            const int pl = tid >> 7, cb = (tid >> 6) & 1, row = tid & 63;
            const float4 v = reinterpret_cast<const float4*>(red)[(pl * 2 + cb) * 64 + row];
            st16_sc1(mk_rsrc(XH + (long)t * PROWS * PH, XB), (u32)((((pl * 64 + 2 * bs + cb) * PROWS) + row) * 16), v);
        }
        publish(cnt(ck, t));
        if (st_on) STAMP(3, 2);
        // ---- logits of step t: columns [80 vb, 80 vb + 80) of every column chunk (vb = b + 64 ch: vocabularies above 5120 words take a second /
        //      third pass over the same h operands; only the first chunk's h0 / h2 part runs ahead of h1's arrival) ----
        float4 ar[8][2][2];
        const int ws = __builtin_amdgcn_readfirstlane(w);
        // the stream's own h(t) for the recurrent product at the end of the step: its 32 fragment loads ride behind the fold
        auto fetch_r = [&](int s_) {
            const __amdgpu_buffer_rsrc_t ra = mk_rsrc(XH + (long)t * PROWS * PH, XB);
#pragma unroll
            for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) {
                    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(ra, (u32)(((lane >> 5) * PROWS + (lane & 31)) * 16),
                                                                          (u32)((((pl * 64 + 16 * ws + 2 * s_) * PROWS) + 32 * rb) * 16), 16);
                    ar[s_][rb][pl] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
                }
        };
        for (int ch = 0; ch < Q.nch; ++ch) {
        const int vb = b + LWG * ch;
        const bool last_ch = ch + 1 == Q.nch;
        f32x4v la[4][LCT];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int c = 0; c < LCT; ++c) la[i][c] = f32x4v{0.f, 0.f, 0.f, 0.f};
        // one k step (32 wide) of stream block kb: 10 B fragments (5 column tiles x 2 planes: static weights, fetched ahead of the hand-off
        // waits) + 8 A fragments (4 row tiles x 2 planes)
        // every fragment address = one lane-dependent VGPR offset + a wave-uniform scalar offset (kept in SGPRs / immediates: per-fragment
        // vector offsets would be hoisted out of the step loop as invariants, a few hundred registers' worth)
        const u32 vo_b = (u32)lane * 16u;
        const u32 vo_a64 = (u32)(((lane >> 4) * PROWS + (lane & 15)) * 16), vo_a32 = (u32)(((lane >> 4) * HR + (lane & 15)) * 16);
        const __amdgpu_buffer_rsrc_t rl = mk_rsrc(Q.LIMG + (long)vb * (3 * 16 * LCT * 2 * 64), 3 * 16 * LCT * 2 * 64 * 16);
        auto fetch_b = [&](int kb, int si, float4 (&fb)[LCT][2]) {
            const u32 so = (u32)(((kb * 16 + 4 * ws + si) * LCT) * 2 * 64 * 16);
#pragma unroll
            for (int c = 0; c < LCT; ++c)
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) {
                    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rl, vo_b, so + (u32)((c * 2 + pl) * 64 * 16), 0);
                    fb[c][pl] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
                }
        };
        auto fetch_a = [&](int kb, int si, const float* abase, float4 (&fa)[4][2]) {
            const int s_ = 4 * ws + si;
            const __amdgpu_buffer_rsrc_t ra = mk_rsrc(abase, XB);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) {
                    const u32 so = kb == 1 ? (u32)((i >> 1) * XBH + (((pl * 64 + 4 * s_) * HR) + 16 * (i & 1)) * 16)
                                           : (u32)((((pl * 64 + 4 * s_) * PROWS) + 16 * i) * 16);
                    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(ra, kb == 1 ? vo_a32 : vo_a64, so, 16);
                    fa[i][pl] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
                }
        };
        auto mma = [&](const float4 (&fa)[4][2], const float4 (&fb)[LCT][2]) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const f16x8p ah = as_f16x8(fa[i][0]), al = as_f16x8(fa[i][1]);
#pragma unroll
                for (int c = 0; c < LCT; ++c) {
                    const f16x8p bh = as_f16x8(fb[c][0]), bl = as_f16x8(fb[c][1]);
                    la[i][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, la[i][c], 0, 0, 0);
                    la[i][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, la[i][c], 0, 0, 0);
                    la[i][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, la[i][c], 0, 0, 0);
                }
            }
        };
        u32 peek_h1 = 0;
        {
            // h0's four k steps, then h2's: four rotating buffers, three steps' loads in flight; the first three steps' weights go out ahead
            // of the wait for the two streams' h(t)
            const float* a0 = P.XH0 + (long)t * PROWS * PH;
            const float* a2 = P.XH2 + (long)t * PROWS * PH;
            float4 fa0[4][2], fb0[LCT][2], fa1[4][2], fb1[LCT][2], fa2[4][2], fb2[LCT][2];
            fetch_b(0, 0, fb0); fetch_b(0, 1, fb1);
            if (ch == 0 && !wait_total2(P, cnt(ck, t), NS, cnt(ck_other, t), NS, flag, 1000u * (ck + 1) + t)) return;
            if (st_on && ch == 0) STAMP(3, 3);
            fetch_a(0, 0, a0, fa0); fetch_a(0, 1, a0, fa1);
            // steps 0..7 = (kb, si): (0,0) (0,1) (0,2) (0,3) (2,0) (2,1) (2,2) (2,3); three rotating buffers, two steps' loads in flight
            fetch_b(0, 2, fb2); fetch_a(0, 2, a0, fa2);
            mma(fa0, fb0);
            __builtin_amdgcn_sched_barrier(0);
            fetch_b(0, 3, fb0); fetch_a(0, 3, a0, fa0);
            mma(fa1, fb1);
            __builtin_amdgcn_sched_barrier(0);
            fetch_b(2, 0, fb1); fetch_a(2, 0, a2, fa1);
            mma(fa2, fb2);
            __builtin_amdgcn_sched_barrier(0);
            fetch_b(2, 1, fb2); fetch_a(2, 1, a2, fa2);
            mma(fa0, fb0);
            __builtin_amdgcn_sched_barrier(0);
            fetch_b(2, 2, fb0); fetch_a(2, 2, a2, fa0);
            mma(fa1, fb1);
            __builtin_amdgcn_sched_barrier(0);
            fetch_b(2, 3, fb1); fetch_a(2, 3, a2, fa1);
            mma(fa2, fb2);
            __builtin_amdgcn_sched_barrier(0);
            if (ch == 0 && tid < 64) peek_h1 = peek_issue(cnt2(C_H1, t, 0), cnt2(C_H1, t, nhalf - 1));      // h1(t) is usually complete by now: sample its counters behind the last products
            mma(fa0, fb0);
            mma(fa1, fb1);
        }
        if (st_on && ch == 0) STAMP(3, 4);
        {
            // h1's four k steps: all weights ahead of the wait, all of h1's fragments behind it
            const float* a1 = Q.XH1 + (long)t * 2 * HR * PH;          // both halves: [m][plane][k / 8][32 rows][8 halves] (2 x 64 KB = XB bytes)
            float4 fa0[4][2], fb0[LCT][2], fa1[4][2], fb1[LCT][2], fa2[4][2], fb2[LCT][2], fb3[LCT][2];
            fetch_b(1, 0, fb0); fetch_b(1, 1, fb1); fetch_b(1, 2, fb2); fetch_b(1, 3, fb3);
            if (ch == 0 && !wait_peeked2(P, peek_h1, cnt2(C_H1, t, 0), HG1, cnt2(C_H1, t, nhalf - 1), HG1, flag, 150000u + t)) return;
            if (st_on && ch == 0) STAMP(3, 5);
            fetch_a(1, 0, a1, fa0); fetch_a(1, 1, a1, fa1); fetch_a(1, 2, a1, fa2);
            __builtin_amdgcn_sched_barrier(0);
            mma(fa0, fb0);
            __builtin_amdgcn_sched_barrier(0);
            fetch_a(1, 3, a1, fa0);
            mma(fa1, fb1); mma(fa2, fb2); mma(fa0, fb3);
        }
        if (st_on && last_ch) STAMP(3, 6);
        __builtin_amdgcn_sched_barrier(0);
        if (last_ch && t + 1 < S) {
#pragma unroll
            for (int s_ = 0; s_ < 4; ++s_) fetch_r(s_);          // the first half now (register budget), the second behind the publish
        }
        __builtin_amdgcn_sched_barrier(0);
        // the four waves' partial tiles -> finished logits of (event ev, 20 columns) per thread
        float lv[LCT][4];
#pragma unroll
        for (int c = 0; c < LCT; ++c) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r)          // [row tile i][r][lane group][16 columns, 4-float groups XOR-ed with r]: a wave's store covers 64 distinct banks,
                    red[w * 1024 + ((i * 4 + r) * 4 + (lane >> 4)) * 16 + ((lane & 15) ^ (r << 2))] = la[i][c][r];      // and so do the float4 reads below
            __syncthreads();
            const int ro = (((ev >> 4) * 4 + (ev & 3)) * 4 + ((ev >> 2) & 3)) * 16 + 4 * (q4 ^ (ev & 3));          // event ev = 16 i + 4 group + r
            const float4 p0 = *reinterpret_cast<const float4*>(red + ro), p1 = *reinterpret_cast<const float4*>(red + 1024 + ro),
                         p2 = *reinterpret_cast<const float4*>(red + 2048 + ro), p3 = *reinterpret_cast<const float4*>(red + 3072 + ro);
            const float4 sc4 = *reinterpret_cast<const float4*>(lsc + LCOLS * ch + 16 * c + 4 * q4), bi4 = *reinterpret_cast<const float4*>(lbi + LCOLS * ch + 16 * c + 4 * q4);
            lv[c][0] = ((p0.x + p1.x) + (p2.x + p3.x)) * sc4.x + bi4.x;
            lv[c][1] = ((p0.y + p1.y) + (p2.y + p3.y)) * sc4.y + bi4.y;
            lv[c][2] = ((p0.z + p1.z) + (p2.z + p3.z)) * sc4.z + bi4.z;
            lv[c][3] = ((p0.w + p1.w) + (p2.w + p3.w)) * sc4.w + bi4.w;
            __syncthreads();
        }
        if (Q.force_eos > 0 && t + 1 >= Q.force_eos && vb == 0 && q4 == 0) lv[0][0] = 1e30f;          // diagnostic: column 0 wins from this step on
        {
            float mx = -INFINITY;
            int mi = 0x7fffffff;
#pragma unroll
            for (int c = 0; c < LCT; ++c)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int gv = LCOLS * vb + 16 * c + 4 * q4 + e;
                    if (gv < V1 && lv[c][e] > mx) { mx = lv[c][e]; mi = gv; }          // ascending index per thread: first maximum kept
                }
#pragma unroll
            for (int off = 1; off <= 2; off <<= 1) {
                const float om = __shfl_xor(mx, off, 64);
                const int oi = __shfl_xor(mi, off, 64);
                if (om > mx || (om == mx && oi < mi)) { mx = om; mi = oi; }
            }
            float se = 0.f;
#pragma unroll
            for (int c = 0; c < LCT; ++c)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int gv = LCOLS * vb + 16 * c + 4 * q4 + e;
                    if (gv < V1) se += __expf(lv[c][e] - mx);
                }
            se += __shfl_xor(se, 1, 64);
            se += __shfl_xor(se, 2, 64);
            if (q4 == 0 && ev < N) {
                if (mi != 0x7fffffff) {          // (a workgroup past the end of the vocabulary has nothing to fold)
                    u32 u = __float_as_uint(mx);
                    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);           // order-preserving map of the float onto unsigned
                    const unsigned long long key = ((unsigned long long)u << 32) | (unsigned long long)(0xFFFFFFFFu - (u32)mi);
                    __hip_atomic_fetch_max(Q.KEY + (long)t * PROWS + ev, key, __ATOMIC_RELAXED, ECHR_AGENT);
                }
                float* lp = Q.LSE + (((long)t * PROWS + ev) * (LWG * Q.nch) + vb) * 2;
                lp[0] = mx; lp[1] = se;
            }
        }
        }          // column chunks
        publish(Q.cnt_tok + (long)t * CNT_LINE);
        if (st_on) STAMP(3, 7);
        // ---- the stream's complete h(t) (waited for above) x W_hh: the recurrent part of step t + 1, behind the token's hand-off ----
        if (t + 1 < S) {
            f32x16 acc[2][2];
#pragma unroll
            for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                    for (int g = 0; g < 16; ++g) acc[rb][cb][g] = 0.f;
#pragma unroll
            for (int s_ = 4; s_ < 8; ++s_) fetch_r(s_);
#pragma unroll
            for (int s_ = 0; s_ < 8; ++s_) {
                const float4* bp = wimg + ((long)(w * 8 + s_) * 2) * 2 * 64 + lane;
                f16x8p bh[2], bl[2];
#pragma unroll
                for (int cb = 0; cb < 2; ++cb) { bh[cb] = as_f16x8(bp[(cb * 2) * 64]); bl[cb] = as_f16x8(bp[(cb * 2 + 1) * 64]); }
#pragma unroll
                for (int rb = 0; rb < 2; ++rb) {
                    const f16x8p ah = as_f16x8(ar[s_][rb][0]), al = as_f16x8(ar[s_][rb][1]);
#pragma unroll
                    for (int cb = 0; cb < 2; ++cb) {
                        acc[rb][cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[cb], acc[rb][cb], 0, 0, 0);
                        acc[rb][cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[cb], acc[rb][cb], 0, 0, 0);
                        acc[rb][cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[cb], acc[rb][cb], 0, 0, 0);
                    }
                }
            }
            // token t is on its way (this workgroup published its fold above): sample its counter behind the tile sums
            if (tid < 64) peek_tok = peek_issue(Q.cnt_tok + (long)t * CNT_LINE, Q.cnt_tok + (long)t * CNT_LINE);
#pragma unroll
            for (int rd = 0; rd < 4; ++rd) {
                const int rb = rd >> 1, cb = rd & 1;
                acc_to_lds32(acc[rb][cb], red, w, lane);
                __syncthreads();
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int o = gr * 32 + 8 * g + g8;
                    rec[rd][g] = (red[o] + red[HR * 32 + o] + red[2 * HR * 32 + o] + red[3 * HR * 32 + o]) * (H2_INV_SA * invb[32 * cb + 8 * g + g8]);
                }
                __syncthreads();
            }
        }
        if (st_on) STAMP(3, 8);
    }
}

template <bool BIG>
__global__ __launch_bounds__(256, 1) void dec_persist_sample_kernel(PersistK2 P2, PersistK P1, PersistS Q) {
    const int rb = persist_role_index(blockIdx.x, P2.xcd_map);
    if (rb < 2 * HWG) dec_persist_att2_body<true, BIG, true>(P2, rb, &Q);
    else dec_persist_lstm_samp_body(P1, Q, rb - 2 * HWG);
}

// logit weights -> the image the logits role streams: one power-of-two scale per vocabulary column over the whole contraction (3 x 512)
__global__ __launch_bounds__(256) void logit_scale_kernel(const float* __restrict__ W, int V1, int nvb, float* __restrict__ sc, float* __restrict__ inv) {
    const int v = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;          // one wave per column
    if (v >= nvb * LCOLS) return;
    float mx = 0.f;
    if (v < V1) {
        const float4* wp = reinterpret_cast<const float4*>(W + (long)v * 3 * PH);
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const float4 x = wp[lane + 64 * i];
            mx = fmaxf(fmaxf(mx, fmaxf(fabsf(x.x), fabsf(x.y))), fmaxf(fabsf(x.z), fabsf(x.w)));
        }
    }
    mx = wave_max(mx);
    if (lane == 0) {
        const int ex = (int)((__float_as_uint(mx) >> 23) & 0xFFu);
        int e = (ex == 0 || ex == 255) ? 14 : ex - 127;
        e = max(e, 14 - 126);
        sc[v] = __uint_as_float((unsigned)(127 + 14 - e) << 23);
        inv[v] = ldexpf(1.f, e - 14);
    }
}
__global__ __launch_bounds__(256) void logit_image_kernel(const float* __restrict__ W, int V1, int nvb, const float* __restrict__ sc, float4* __restrict__ img) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;           // (workgroup, stream block, k step, tile, lane)
    if (idx >= (long)nvb * 3 * 16 * LCT * 64) return;
    const int lane = (int)(idx & 63);
    long r = idx >> 6;
    const int c = (int)(r % LCT); r /= LCT;
    const int s_ = (int)(r & 15); r >>= 4;
    const int kb = (int)(r % 3), b = (int)(r / 3);
    const int v = LCOLS * b + 16 * c + (lane & 15), kk = PH * kb + 32 * s_ + 8 * (lane >> 4);
    float xv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (v < V1) {
        const float4 x0 = *reinterpret_cast<const float4*>(W + (long)v * 3 * PH + kk), x1 = *reinterpret_cast<const float4*>(W + (long)v * 3 * PH + kk + 4);
        xv[0] = x0.x; xv[1] = x0.y; xv[2] = x0.z; xv[3] = x0.w; xv[4] = x1.x; xv[5] = x1.y; xv[6] = x1.z; xv[7] = x1.w;
    }
    const float f = sc[v];
    unsigned hw[8], lw[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { unsigned short hi, lo; split_h2(xv[j] * f, hi, lo); hw[j] = hi; lw[j] = lo; }
    const long base = ((((long)(b * 3 + kb) * 16 + s_) * LCT + c) * 2) * 64 + lane;
    reinterpret_cast<uint4*>(img)[base] = make_uint4(hw[0] | (hw[1] << 16), hw[2] | (hw[3] << 16), hw[4] | (hw[5] << 16), hw[6] | (hw[7] << 16));
    reinterpret_cast<uint4*>(img)[base + 64] = make_uint4(lw[0] | (lw[1] << 16), lw[2] | (lw[3] << 16), lw[4] | (lw[5] << 16), lw[6] | (lw[7] << 16));
}

// keys + per-workgroup (maximum, sum of exponentials) -> seq / seq_logp / the unfinished bookkeeping of OldModel.sample (:171-183): one
// workgroup per event; wave 0 turns the event's L tokens into the unfinished flags with one ballot (unfinished after step t <=> every
// arg-max of steps 0..t is a word: the network kept consuming the raw arg-max, only the emitted token is masked), then the four waves share
// the steps: the sum over the 64 x nch logits workgroups' partials runs in a fixed butterfly order.  L <= 64 (checked by the caller).
__global__ __launch_bounds__(256) void sample_finish_kernel(const unsigned long long* __restrict__ KEY, const float* __restrict__ LSE, int N, int L, int nch, long stride_s,
                                                          long long* __restrict__ seq, float* __restrict__ seq_logp, int* __restrict__ n_unfinished, const u32* __restrict__ stop_words) {
    __shared__ unsigned long long skey[64];
    __shared__ unsigned long long sun;
    const int ng = blockIdx.x, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int n = ng & (PROWS - 1);
    KEY += (long)(ng / PROWS) * (stride_s / 2);          // the event's group of 64 has its own keys / partials
    LSE += (long)(ng / PROWS) * stride_s;
    seq += (long)(ng - n) * L; seq_logp += (long)(ng - n) * L;
    if (w == 0) {
        const unsigned long long key = lane < L ? KEY[(long)lane * PROWS + n] : 0ull;
        skey[lane] = key;
        const unsigned long long word = __ballot(lane < L && (0xFFFFFFFFu - (u32)key) > 0u);      // bit t: the arg-max of step t is a word
        if (lane == 0) sun = word;
    }
    __syncthreads();
    const unsigned long long word = sun;
    for (int t = w; t < L; t += 4) {
        const unsigned long long key = skey[t];
        u32 u = (u32)(key >> 32);
        u = (u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u;
        const float M = __uint_as_float(u);
        const int bi = (int)(0xFFFFFFFFu - (u32)key);
        float s = 0.f;
        for (int c = 0; c < nch; ++c) {          // chunk by chunk: a fixed order
            const float* lp = LSE + (((long)t * PROWS + n) * (LWG * nch) + LWG * c + lane) * 2;
            const float m = lp[0];
            s += m > -INFINITY ? lp[1] * expf(m - M) : 0.f;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
        const unsigned long long need = t == 63 ? ~0ull : ((2ull << t) - 1ull);
        const int un = (word & need) == need;
        if (lane == 0) {
            seq[(long)n * L + t] = un ? bi : 0;
            seq_logp[(long)n * L + t] = key ? -logf(s) : 0.f;          // (steps behind an early stop were never computed: their columns are trimmed by the caller)
            if (un) atomicAdd(&n_unfinished[t + 1], 1);
            if (ng == 0 && t == 0) n_unfinished[0] = stop_words[0] == STOP_ALL_FINISHED ? 1 : 0;          // [0] is otherwise unused: 1 = group 0's launch stopped early
        }
    }
}

// ---- the forward pair as ONE launch: workgroups [0, 2 HWG) run the attention chain, [2 HWG, 2 HWG + 2 NS) the two plain LSTM streams.
// Two concurrent launches on two HIP streams need two hardware queues; a process that owns more streams than the runtime has queues
// (collective streams of a data-parallel run, user streams) can find both streams on one queue, and the pair then runs back to back.
// One grid of 256 workgroups has no such dependence.
template <bool H2, bool BIG>
__global__ __launch_bounds__(256, 1) void dec_persist_fwd_kernel(PersistK2 P2, PersistK P1) {
    const int rb = persist_role_index(blockIdx.x, P2.xcd_map);
    if (rb < 2 * HWG) dec_persist_att2_body<H2, BIG>(P2, rb);
    else if (H2) dec_persist_lstm_h2_body(P1, rb - 2 * HWG);
    else dec_persist_lstm_body(P1, rb - 2 * HWG);
}

// ==========================================================================================================================
// PERSISTENT REVERSE RECURRENCE (backward): all S timesteps of the three streams' BPTT in two concurrent launches.
// Reference semantics: autograd of models/OldModel_NEW.py:801-823 (ThreeStream_Core.forward) and :376-401 (Attention.forward); the
// launch-per-phase form is decoder.hip's bwd_step (four dependent launches per timestep).  Per reverse timestep the attention chain
// makes three hand-offs:
//   gate-gradient workgroups ("GD", 32 x 16 hidden units): d h1(t) = d OUTD part + [d G1(t+1) . W_hh1] (slabs) + d q(t+1) . W_h2a
//       (MFMA on the ingested d q) -> LSTM-cell gradient -> d G1(t)                                  --- hand-off 1 (32 -> 128) --->
//   product workgroups ("P", 32 column tiles x 4 k-slices of 512 gate columns): d ATT(t) partial = d G1(t)[:, slice] . W_att[slice, tile]
//       --- hand-off 2 (128 -> 192) --->, then (off the critical path) the d h1 slab of the same slice for step t-1
//   attention workgroups (all 192, operands register-resident as in the forward kernel): d score, d q(t) partial (fp32 atomics per event)
//       --- hand-off 3 (192 -> 32) ---> back to the gate-gradient workgroups.
// The plain LSTM streams 0 / 2 run in their own 64-workgroup launch (one hand-off per step: a workgroup owns 16 units, ingests all of
// d G_k(t+1) and multiplies by its 16 columns of W_hh_k).  Hand-off protocol, abort handling and MFMA fragment scheme: as forward.
// ==========================================================================================================================
enum { CB_DG = 0, CB_DA = 1, CB_HH = 2, CB_DQ = 3, CB_G0 = 4, CB_G2 = 5, CB_P0 = 6, CB_P2 = 7, CB_KINDS = 8 };
constexpr int XPSTEP = 32 * 4 * PROWS * 16;     // floats per timestep of an LSTM stream's partial-product exchange [dst workgroup][source k group][row][16]
constexpr int NGD = 32, NP = 128;
constexpr int XSTEP4 = 4 * PROWS * PH;          // floats per timestep of a [4][512-wide] exchange buffer

struct PersistLayoutB { long cnt, xdq, zero_end, xdg, xda, xdh, xg0, xg2, xp0, xp2, total; };
static PersistLayoutB persist_layout_b(int S) {
    PersistLayoutB L;
    long off = 0;
    auto take = [&](long n) { long o = off; off += (n + 63) / 64 * 64; return o; };
    L.cnt = take((long)CB_KINDS * (S + 1) * CNT_LINE);
    L.xdq = take((long)S * PROWS * PH);
    L.xda = take((long)S * PROWS * PH);          // d ATT and the d h1 product: the four k-slice partials are added atomically
    L.xdh = take((long)S * PROWS * PH);
    L.zero_end = off;
    L.xdg = take((long)S * XSTEP4);
    L.xg0 = take((long)S * XSTEP4);
    L.xg2 = take((long)S * XSTEP4);
    L.xp0 = take((long)S * XPSTEP);
    L.xp2 = take((long)S * XPSTEP);
    L.total = off;
    return L;
}
long persist_bwd_ws_floats(int S);

struct PersistB {
    unsigned c3d_bytes;            // bytes of the whole [Tv, D] feature tensor (BIG kernels' C3D row stream)
    int N, A, D, S, ld_att;
    const float* w_hh[3]; const float* w_h2a; const float* w_att; const float* w_alpha;
    const float* PALL; const float* c3d; const int* ev_start; const int* ev_len;
    const float* GATES[3]; const float* CS[3]; const float* QS; const float* WT; const float* ATT;
    const float* DOUT;
    float* DG[3]; float* DQ; float* DSC;
    float *XDG, *XDA, *XDH, *XDQ, *XG0, *XG2, *XP0, *XP2;
    int lstm_kgroups;
    u32* cnt; u32* abort_word; u32* host_flag;
    u32 spin_limit, inject;          // bound of every hand-off spin; diagnostic: the wait whose code equals `inject` never completes (0 = none)
    unsigned long long* stamps;
    DropCfg dh, dout;
    int xcd_map;                   // 1: role index from (XCD, slot) instead of blockIdx.x, see persist_role_index
};
#define BSTAMP(role, i) do { if (P.stamps && tid == 0 && t >= 0) P.stamps[((role) * S + t) * 16 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)

// B image of one 16-column tile whose source is k-strided: element (k, cc) = W[k * ld_k + cc] for k < K, cc < ncols; zero elsewhere
__device__ __forceinline__ void fill_bimg_t(float4* img, const float* W, long ld_k, int K, int ncols, int tid) {
    // all 32 loads of a thread are requested before the first LDS store (the rolled loop waited for each group of four: eight dependent
    // memory round trips per image in front of the first timestep)
    float v[8][4];
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int idx = tid + 256 * it;
        const int lane = idx & 63, c = (idx >> 6) & 7, w = idx >> 9;
        const int cc = lane & 15, kq = lane >> 4;
        const int k = 128 * w + 16 * c + 4 * kq;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[it][j] = (k + j < K && cc < ncols) ? W[(long)(k + j) * ld_k + cc] : 0.f;
    }
#pragma unroll
    for (int it = 0; it < 8; ++it) img[tid + 256 * it] = make_float4(v[it][0], v[it][1], v[it][2], v[it][3]);
}

struct CellGrad { float4 dg[4]; float4 dc; };
// LSTM-cell backward for 4 consecutive units of one event (lstm_pointwise_bwd_kernel of decoder.hip): dhv = upstream d h (before the
// recurrent dropout mask), dcin = carried d c
__device__ __forceinline__ float cg1(float dhv, float gi, float gf, float gg, float go, float cn, float cp, float dcin, float& dgi, float& dgf,
                                     float& dgg, float& dgo) {
    const float tc = fast_tanh(cn);
    const float dcv = dhv * go * (1.f - tc * tc) + dcin;
    dgi = dcv * gg * gi * (1.f - gi);
    dgf = dcv * cp * gf * (1.f - gf);
    dgg = dcv * gi * (1.f - gg * gg);
    dgo = dhv * tc * go * (1.f - go);
    return dcv * gf;
}
__device__ __forceinline__ float4 f4add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }

// shared by the LSTM kernel and the GD role: upstream d h of (event gn, units u0..u0+3 of stream k) from d OUTD, then gate gradients
struct GradIn { float4 dout, g[4], cn, cp; };
__device__ __forceinline__ GradIn load_grad_in(const PersistB& P, int k, int t, int gn, int u0) {
    GradIn r;
    const int n = min(gn, P.N - 1);
    r.dout = *reinterpret_cast<const float4*>(P.DOUT + ((long)t * P.N + n) * 3 * PH + k * PH + u0);
    const float* gp = P.GATES[k] + ((long)t * P.N + n) * 4 * PH + u0;
#pragma unroll
    for (int g = 0; g < 4; ++g) r.g[g] = *reinterpret_cast<const float4*>(gp + g * PH);
    r.cn = *reinterpret_cast<const float4*>(P.CS[k] + ((long)(t + 1) * P.N + n) * PH + u0);
    r.cp = *reinterpret_cast<const float4*>(P.CS[k] + ((long)t * P.N + n) * PH + u0);
    return r;
}
struct DropM { float mo[4], mh[4]; };
// the two dropout multipliers of (event gn, units u0..u0+3, stream k, step t): index-only work, done ahead of the hand-off waits
__device__ __forceinline__ DropM drop_masks4(const PersistB& P, int k, int t, int gn, int u0) {
    DropM m;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        m.mo[i] = drop_mult(P.dout, (unsigned)(gn * 3 * PH + k * PH + u0 + i), (unsigned)t, 4u);
        m.mh[i] = drop_mult(P.dh, (unsigned)(gn * PH + u0 + i), (unsigned)t, (unsigned)(1 + k));
    }
    return m;
}
__device__ __forceinline__ CellGrad cell_grad4(const PersistB& P, const GradIn& in, const DropM& m, float4 dh_rec, float4 dc) {
    CellGrad o;
    float dhv[4] = {in.dout.x, in.dout.y, in.dout.z, in.dout.w};
    const float rec[4] = {dh_rec.x, dh_rec.y, dh_rec.z, dh_rec.w};
    const float dcin[4] = {dc.x, dc.y, dc.z, dc.w};
    const float gi[4] = {in.g[0].x, in.g[0].y, in.g[0].z, in.g[0].w}, gf[4] = {in.g[1].x, in.g[1].y, in.g[1].z, in.g[1].w};
    const float gg[4] = {in.g[2].x, in.g[2].y, in.g[2].z, in.g[2].w}, go[4] = {in.g[3].x, in.g[3].y, in.g[3].z, in.g[3].w};
    const float cn[4] = {in.cn.x, in.cn.y, in.cn.z, in.cn.w}, cp[4] = {in.cp.x, in.cp.y, in.cp.z, in.cp.w};
    float dg[4][4], dcn[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float v = (dhv[i] * m.mo[i] + rec[i]) * m.mh[i];
        dcn[i] = cg1(v, gi[i], gf[i], gg[i], go[i], cn[i], cp[i], dcin[i], dg[0][i], dg[1][i], dg[2][i], dg[3][i]);
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) o.dg[g] = make_float4(dg[g][0], dg[g][1], dg[g][2], dg[g][3]);
    o.dc = make_float4(dcn[0], dcn[1], dcn[2], dcn[3]);
    return o;
}

// ---- backward kernel 1: the two plain LSTM streams ------------------------------------------------------------------------------
__device__ __forceinline__ void dec_persist_lstm_bwd_body(const PersistB& P, const int bid) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    float4* wimg = reinterpret_cast<float4*>(lds);
    float* red = reinterpret_cast<float*>(lds + LDS_W);
    int* flag = reinterpret_cast<int*>(lds + LDS_W + LDS_RED);
    const int b = bid, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const bool is_s0 = b < NS;
    const int N = P.N, S = P.S;
    const int k = is_s0 ? 0 : 2, bs = is_s0 ? b : b - NS, ck = is_s0 ? CB_G0 : CB_G2;
    float* XG = is_s0 ? P.XG0 : P.XG2;
    auto cnt = [&](int kind, int t) { return P.cnt + ((long)kind * (S + 1) + t) * CNT_LINE; };
    // d h_k(t)[:, u] = sum_c d G_k(t+1)[:, c] W_hh_k[c, u]: B[k = c][col = u], four k-slices (= gates) of 512
    for (int ks = 0; ks < 4; ++ks) fill_bimg_t(wimg + ks * 2048, P.w_hh[k] + (long)ks * PH * PH + 16 * bs, PH, PH, 16, tid);
    __syncthreads();
    const int gn = tid >> 2, gq = tid & 3, u0 = 16 * bs + 4 * gq;
    float4 dc = make_float4(0.f, 0.f, 0.f, 0.f);
    const u32 XB = PROWS * PH * 4;
    for (int t = S - 1; t >= 0; --t) {
        if (b == 0) BSTAMP(3, 0);
        const GradIn in = load_grad_in(P, k, t, gn, u0);
        const DropM dm = drop_masks4(P, k, t, gn, u0);
        float4 rec = make_float4(0.f, 0.f, 0.f, 0.f);
        if (t < S - 1) {
            if (!wait_total(P, cnt(ck, t + 1), NS, flag, 5000u * (ck + 1) + t)) return;
            if (b == 0) BSTAMP(3, 1);
            f32x4 acc[4];
#pragma unroll
            for (int rb = 0; rb < 4; ++rb) acc[rb] = f32x4{0.f, 0.f, 0.f, 0.f};
            for (int ks = 0; ks < 4; ++ks) {
                float4 a[4][8];
                load_afrag<1>(a, mk_rsrc(XG + (long)(t + 1) * XSTEP4 + (long)ks * PROWS * PH, XB), w, lane);
                mfma_tile(acc, a, wimg + ks * 2048 + w * 512, lane);
            }
            if (b == 0) BSTAMP(3, 2);
            acc_to_lds(acc, red, w, lane);
            __syncthreads();
            const float* rp = red + gn * 16 + 4 * gq;
#pragma unroll
            for (int ww = 0; ww < 4; ++ww) rec = f4add(rec, *reinterpret_cast<const float4*>(rp + ww * PROWS * 16));
        }
        const CellGrad cgd = cell_grad4(P, in, dm, rec, dc);
        dc = cgd.dc;
        // exchange layout [gate][unit / 16][n][16]
        const __amdgpu_buffer_rsrc_t rx = mk_rsrc(XG + (long)t * XSTEP4, 4 * XB);
#pragma unroll
        for (int g = 0; g < 4; ++g) st16_sc1(rx, (u32)((((g * 32 + bs) * PROWS + gn) * 16 + 4 * gq) * 4), cgd.dg[g]);
        if (b == 0) BSTAMP(3, 3);
        publish(cnt(ck, t));            // (its barrier also protects `red`)
        if (b == 0) BSTAMP(3, 4);
        if (gn < N) {
            float* dgp = P.DG[k] + ((long)t * N + gn) * 4 * PH + u0;
#pragma unroll
            for (int g = 0; g < 4; ++g) *reinterpret_cast<float4*>(dgp + g * PH) = cgd.dg[g];
        }
    }
}
// The same role with the contraction split over workgroup groups: workgroup bs = 4 cg + kg multiplies the k group kg (= gate kg, 512 gate
// columns) of d G(t+1) by W_hh[kg-th 512 rows][64 units of column group cg] and hands the three [64 x 16] partial tiles it does not own to
// their owners (the workgroups of the same cg).  It ingests 128 KB per step instead of 512 KB (the per-CU ingest rate, not the fp32 MFMA, had
// set the step: 15 us), at the price of a second, small hand-off per step.
__device__ __forceinline__ void dec_persist_lstm_bwd_kg_body(const PersistB& P, const int bid) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    float4* wimg = reinterpret_cast<float4*>(lds);
    float* red = reinterpret_cast<float*>(lds + LDS_W);
    int* flag = reinterpret_cast<int*>(lds + LDS_W + LDS_RED);
    const int b = bid, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const bool is_s0 = b < NS;
    const int N = P.N, S = P.S;
    const int k = is_s0 ? 0 : 2, bs = is_s0 ? b : b - NS, ck = is_s0 ? CB_G0 : CB_G2, cp = is_s0 ? CB_P0 : CB_P2;
    const int kg = bs & 3, cg = bs >> 2;
    float* XG = is_s0 ? P.XG0 : P.XG2;
    float* XP = is_s0 ? P.XP0 : P.XP2;
    auto cnt = [&](int kind, int t) { return P.cnt + ((long)kind * (S + 1) + t) * CNT_LINE; };
    // B[k = c][col = u]: rows kg * 512 + [0, 512) of W_hh_k, columns 64 cg + 16 ct + [0, 16), ct = 0..3 (tile ct belongs to workgroup 4 cg + ct)
    for (int ct = 0; ct < 4; ++ct) fill_bimg_t(wimg + ct * 2048, P.w_hh[k] + (long)kg * PH * PH + 64 * cg + 16 * ct, PH, PH, 16, tid);
    __syncthreads();
    const int gn = tid >> 2, gq = tid & 3, u0 = 16 * bs + 4 * gq;
    float4 dc = make_float4(0.f, 0.f, 0.f, 0.f);
    const u32 XB = PROWS * PH * 4;
    for (int t = S - 1; t >= 0; --t) {
        if (b == 0) BSTAMP(3, 0);
        const GradIn in = load_grad_in(P, k, t, gn, u0);
        const DropM dm = drop_masks4(P, k, t, gn, u0);
        float4 rec = make_float4(0.f, 0.f, 0.f, 0.f);
        if (t < S - 1) {
            if (!wait_total(P, cnt(ck, t + 1), NS, flag, 5000u * (ck + 1) + t)) return;
            if (b == 0) BSTAMP(3, 1);
            f32x4 acc[4][4];
#pragma unroll
            for (int ct = 0; ct < 4; ++ct)
#pragma unroll
                for (int rb = 0; rb < 4; ++rb) acc[ct][rb] = f32x4{0.f, 0.f, 0.f, 0.f};
            {
                float4 a[4][8];
                load_afrag<1>(a, mk_rsrc(XG + (long)(t + 1) * XSTEP4 + (long)kg * PROWS * PH, XB), w, lane);
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) mfma_tile(acc[ct], a, wimg + ct * 2048 + w * 512, lane);
            }
            if (b == 0) BSTAMP(3, 2);
            // cross-wave sums, one column tile at a time through the 16 KB buffer; foreign tiles leave for their owners
            const __amdgpu_buffer_rsrc_t rp = mk_rsrc(XP + (long)(t + 1) * XPSTEP, (u32)(XPSTEP * 4));
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) {
                acc_to_lds(acc[ct], red, w, lane);
                __syncthreads();
                const float* rq = red + gn * 16 + 4 * gq;
                float4 v = *reinterpret_cast<const float4*>(rq);
#pragma unroll
                for (int ww = 1; ww < 4; ++ww) v = f4add(v, *reinterpret_cast<const float4*>(rq + ww * PROWS * 16));
                if (ct == kg) rec = v;
                else st16_sc1(rp, (u32)(((((4 * cg + ct) * 4 + kg) * PROWS + gn) * 16 + 4 * gq) * 4), v);
                __syncthreads();
            }
            publish(cnt(cp, t + 1));
            if (!wait_total(P, cnt(cp, t + 1), NS, flag, 7000u * (cp + 1) + t)) return;
#pragma unroll
            for (int src = 0; src < 4; ++src)
                if (src != kg) rec = f4add(rec, ld16_sc1(rp, (u32)((((bs * 4 + src) * PROWS + gn) * 16 + 4 * gq) * 4)));
        }
        const CellGrad cgd = cell_grad4(P, in, dm, rec, dc);
        dc = cgd.dc;
        // exchange layout [gate][unit / 16][n][16]
        const __amdgpu_buffer_rsrc_t rx = mk_rsrc(XG + (long)t * XSTEP4, 4 * XB);
#pragma unroll
        for (int g = 0; g < 4; ++g) st16_sc1(rx, (u32)((((g * 32 + bs) * PROWS + gn) * 16 + 4 * gq) * 4), cgd.dg[g]);
        if (b == 0) BSTAMP(3, 3);
        publish(cnt(ck, t));
        if (b == 0) BSTAMP(3, 4);
        if (gn < N) {
            float* dgp = P.DG[k] + ((long)t * N + gn) * 4 * PH + u0;
#pragma unroll
            for (int g = 0; g < 4; ++g) *reinterpret_cast<float4*>(dgp + g * PH) = cgd.dg[g];
        }
    }
}
__global__ __launch_bounds__(256, 1) void dec_persist_lstm_bwd_kernel(PersistB P) {
    if (P.lstm_kgroups) dec_persist_lstm_bwd_kg_body(P, blockIdx.x);
    else dec_persist_lstm_bwd_body(P, blockIdx.x);
}

// ---- backward kernel 2: the attention chain ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 1) void dec_persist_att_bwd_kernel(PersistB P) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    float4* wimg = reinterpret_cast<float4*>(lds);
    float* red = reinterpret_cast<float*>(lds + LDS_WA);
    int* flag = reinterpret_cast<int*>(lds + LDS_WA + LDS_RED_ATT);
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const bool is_gd = b < NGD, is_p = b >= NGD && b < NGD + NP;
    const int N = P.N, D = P.D, S = P.S;
    auto cnt = [&](int kind, int t) { return P.cnt + ((long)kind * (S + 1) + t) * CNT_LINE; };
    const int pct = (b - NGD) & 31, pks = (b - NGD) >> 5;         // product role: column tile, k-slice
    if (is_gd) {
        // d h1[:, u] += sum_j d q[:, j] W_h2a[j, u]
        fill_bimg_t(wimg, P.w_h2a + 16 * b, PH, PH, 16, tid);
    } else if (is_p) {
        // d ATT[:, d] = sum_c d G1[:, c] W_ih1[c, E + d];  d h1[:, u] = sum_c d G1[:, c] W_hh1[c, u];  c in this workgroup's slice
        fill_bimg_t(wimg, P.w_att + (long)pks * PH * P.ld_att + 16 * pct, P.ld_att, PH, max(0, min(16, D - 16 * pct)), tid);
        fill_bimg_t(wimg + 2048, P.w_hh[1] + (long)pks * PH * PH + 16 * pct, PH, PH, 16, tid);
    }
    // ---- attention operands -> registers (as the forward kernel: e^{2p} and the C3D rows of this workgroup's slots) ----
    float* red2 = red;
    float* sal = red + 16 * PH;
    float* sat = sal + PH + 64;           // [512] saved context row ATT[t][n][:] of the current step (zero beyond D)
    const int an = b / 3, ap = b - 3 * an;
    const bool att_live = an < N;
    const int grow_ = 4 * w + (lane >> 4), lr = lane & 15;
    int alen = 0;
    float4 Pr[PSG][8], Cr[PSG][8];
    for (int j = tid; j < PH; j += 256) sal[j] = P.w_alpha[j];
    if (att_live) {
        alen = P.ev_len[an];
        const long row0 = P.ev_start[an];
#pragma unroll
        for (int i = 0; i < PSG; ++i) {
            const int sl = grow_ + 16 * i;
            const int a = min(PSL * ap + min(sl, PSL - 1), alen - 1);
            const float* pr = P.PALL + (row0 + a) * PH + 32 * lr;
            const float* cr = P.c3d + (row0 + a) * D;
#pragma unroll
            for (int h = 0; h < 8; ++h) {
                const float4 pv = *reinterpret_cast<const float4*>(pr + 4 * h);
                Pr[i][h] = make_float4(__expf(2.f * fminf(fmaxf(pv.x, -43.f), 43.f)), __expf(2.f * fminf(fmaxf(pv.y, -43.f), 43.f)),
                                       __expf(2.f * fminf(fmaxf(pv.z, -43.f), 43.f)), __expf(2.f * fminf(fmaxf(pv.w, -43.f), 43.f)));
                const int d = 32 * lr + 4 * h;
                float4 v = *reinterpret_cast<const float4*>(cr + min(d, D - 4));
                if (d >= D) v = make_float4(0.f, 0.f, 0.f, 0.f);
                Cr[i][h] = v;
            }
        }
    } else {
#pragma unroll
        for (int i = 0; i < PSG; ++i)
#pragma unroll
            for (int h = 0; h < 8; ++h) Pr[i][h] = Cr[i][h] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();

    const int gn = tid >> 2, gq = tid & 3, u0 = 16 * b + 4 * gq;        // gate-gradient ownership (GD role)
    float4 dc = make_float4(0.f, 0.f, 0.f, 0.f);
    const u32 XB = PROWS * PH * 4;

    const int srole = b == 0 ? 0 : (b == NGD ? 1 : (b == NGD + NP ? 2 : -1));
    for (int t = S - 1; t >= -1; --t) {
        if (srole >= 0) BSTAMP(srole, 0);
        // ============ GD: d h1(t) -> d G1(t); at t = -1 only d q(0) is copied out ============
        if (is_gd) {
            GradIn in;
            DropM dm;
            if (t >= 0) { in = load_grad_in(P, 1, t, gn, u0); dm = drop_masks4(P, 1, t, gn, u0); }
            float4 rec = make_float4(0.f, 0.f, 0.f, 0.f);
            if (t < S - 1 && t >= 0) {
                // + d G1(t+1) . W_hh1 (the product workgroups' atomically summed tile): complete well before d q(t+1), fetched first
                if (!wait_total(P, cnt(CB_HH, t + 1), NP, flag, 500000u + t + 1)) return;
                rec = ld16_sc1(mk_rsrc(P.XDH + (long)(t + 1) * PROWS * PH, XB), (u32)(((b * PROWS + gn) * 16 + 4 * gq) * 4));
            }
            if (t < S - 1) {
                if (!wait_total(P, cnt(CB_DQ, t + 1), NATT, flag, 400000u + t + 1)) return;
                if (srole >= 0) BSTAMP(srole, 1);
                f32x4 acc[4];
#pragma unroll
                for (int rb = 0; rb < 4; ++rb) acc[rb] = f32x4{0.f, 0.f, 0.f, 0.f};
                {
                    float4 a[4][8];
                    load_afrag<1>(a, mk_rsrc(P.XDQ + (long)(t + 1) * PROWS * PH, XB), w, lane);
                    if (t >= 0) mfma_tile(acc, a, wimg + w * 512, lane);
                    // d q(t+1) row-major for the weight-gradient products: this workgroup stores columns [16 b, 16 b + 16)
                    if (w == (b >> 3)) {
#pragma unroll
                        for (int c = 0; c < 8; ++c)
                            if (c == (b & 7)) {
#pragma unroll
                                for (int rb = 0; rb < 4; ++rb) {
                                    const int n = 16 * rb + (lane & 15);
                                    if (n < N) *reinterpret_cast<float4*>(P.DQ + ((long)(t + 1) * N + n) * PH + 16 * b + 4 * (lane >> 4)) = a[rb][c];
                                }
                            }
                    }
                }
                if (t >= 0) {
                    acc_to_lds(acc, red, w, lane);
                    __syncthreads();
                    const float* rp = red + gn * 16 + 4 * gq;
#pragma unroll
                    for (int ww = 0; ww < 4; ++ww) rec = f4add(rec, *reinterpret_cast<const float4*>(rp + ww * PROWS * 16));
                    if (srole >= 0) BSTAMP(srole, 2);
                    if (srole >= 0) BSTAMP(srole, 3);
                }
            }
            if (t >= 0) {
                const CellGrad cgd = cell_grad4(P, in, dm, rec, dc);
                dc = cgd.dc;
                const __amdgpu_buffer_rsrc_t rx = mk_rsrc(P.XDG + (long)t * XSTEP4, 4 * XB);
#pragma unroll
                for (int g = 0; g < 4; ++g) st16_sc1(rx, (u32)((((g * 32 + b) * PROWS + gn) * 16 + 4 * gq) * 4), cgd.dg[g]);
                if (srole >= 0) BSTAMP(srole, 4);
                publish(cnt(CB_DG, t));
                if (srole >= 0) BSTAMP(srole, 5);
                if (gn < N) {
                    float* dgp = P.DG[1] + ((long)t * N + gn) * 4 * PH + u0;
#pragma unroll
                    for (int g = 0; g < 4; ++g) *reinterpret_cast<float4*>(dgp + g * PH) = cgd.dg[g];
                }
            }
        }
        if (t < 0) break;
        // ============ P: d ATT(t) slab (critical), then the d h1 slab for step t-1 ============
        if (is_p) {
            if (!wait_total(P, cnt(CB_DG, t), NGD, flag, 600000u + t)) return;
            if (srole >= 0) BSTAMP(srole, 6);
            float4 a[4][8];
            load_afrag<1>(a, mk_rsrc(P.XDG + (long)t * XSTEP4 + (long)pks * PROWS * PH, XB), w, lane);
            {
                f32x4 acc[4];
#pragma unroll
                for (int rb = 0; rb < 4; ++rb) acc[rb] = f32x4{0.f, 0.f, 0.f, 0.f};
                mfma_tile(acc, a, wimg + w * 512, lane);
                acc_to_lds(acc, red, w, lane);
                __syncthreads();
                // the four k-slices of a tile add atomically; element e = row * 16 + col of the exchange tile is contiguous over the
                // lanes, so every atomic wave-instruction covers 256 contiguous bytes (the full-rate shape)
                float* xa = P.XDA + (long)t * PROWS * PH + pct * PROWS * 16;
#pragma unroll
                for (int e4 = 0; e4 < 4; ++e4) {
                    const int e = tid + 256 * e4;
                    atomicAdd(xa + e, red[e] + red[PROWS * 16 + e] + red[2 * PROWS * 16 + e] + red[3 * PROWS * 16 + e]);
                }
                if (srole >= 0) BSTAMP(srole, 7);
                publish(cnt(CB_DA, t));
                if (srole >= 0) BSTAMP(srole, 8);
            }
            if (t > 0) {
                f32x4 acc[4];
#pragma unroll
                for (int rb = 0; rb < 4; ++rb) acc[rb] = f32x4{0.f, 0.f, 0.f, 0.f};
                mfma_tile(acc, a, wimg + 2048 + w * 512, lane);
                acc_to_lds(acc, red, w, lane);
                __syncthreads();
                float* xh = P.XDH + (long)t * PROWS * PH + pct * PROWS * 16;
#pragma unroll
                for (int e4 = 0; e4 < 4; ++e4) {
                    const int e = tid + 256 * e4;
                    atomicAdd(xh + e, red[e] + red[PROWS * 16 + e] + red[2 * PROWS * 16 + e] + red[3 * PROWS * 16 + e]);
                }
                publish(cnt(CB_HH, t));
                if (srole >= 0) BSTAMP(srole, 9);
            }
        }
        // ============ attention backward of step t (all workgroups) ============
        {
            float4 q[8];
            float wt[PSG];
            if (att_live) {          // saved forward activations: plain loads, in flight while the hand-off is awaited
                const float* qp = P.QS + ((long)t * N + an) * PH + 32 * lr;
#pragma unroll
                for (int h = 0; h < 8; ++h) q[h] = *reinterpret_cast<const float4*>(qp + 4 * h);
#pragma unroll
                for (int i = 0; i < PSG; ++i) {
                    const int sl = grow_ + 16 * i;
                    const bool valid = sl < PSL && PSL * ap + sl < alen;
                    wt[i] = valid ? P.WT[((long)t * N + an) * P.A + PSL * ap + sl] : 0.f;
                }
                // the saved context row goes to LDS before the wait (no register cost across it; the wait's barriers publish it)
                for (int d = tid; d < PH; d += 256) sat[d] = d < D ? P.ATT[((long)t * N + an) * D + d] : 0.f;
            }
            if (!wait_total(P, cnt(CB_DA, t), NP, flag, 700000u + t)) return;
            if (srole >= 0) BSTAMP(srole, 10);
            if (att_live) {
                // d ATT[n, 32 lr .. +32), exchange layout [d / 16][n][16]; the saved context row rides along (plain loads)
                const __amdgpu_buffer_rsrc_t ra = mk_rsrc(P.XDA + (long)t * PROWS * PH, XB);
                float4 da[8];
                float s0 = 0.f;
#pragma unroll
                for (int h = 0; h < 8; ++h) da[h] = ld16_sc1(ra, (u32)((((2 * lr + (h >> 2)) * PROWS + an) * 16 + 4 * (h & 3)) * 4));
#pragma unroll
                for (int h = 0; h < 8; ++h) {
                    const float4 at = *reinterpret_cast<const float4*>(sat + 32 * lr + 4 * h);
                    s0 += at.x * da[h].x + at.y * da[h].y + at.z * da[h].z + at.w * da[h].w;
                }
                s0 = row16_sum(s0);
                float dsc[PSG];
#pragma unroll
                for (int i = 0; i < PSG; ++i) {
                    float dw = 0.f;
#pragma unroll
                    for (int h = 0; h < 8; ++h) dw += Cr[i][h].x * da[h].x + Cr[i][h].y * da[h].y + Cr[i][h].z * da[h].z + Cr[i][h].w * da[h].w;
                    dw = row16_sum(dw);
                    dsc[i] = wt[i] * (dw - s0);
                }
                // d q[j] = sum_a dsc_a alpha_j (1 - tanh^2(p_aj + q_j)),  1 - tanh^2 = 4 r (1 - r) with r = 1 / (e^{2p} e^{2q} + 1)
#pragma unroll
                for (int h = 0; h < 8; ++h) {
                    const float4 eq = make_float4(__expf(2.f * fminf(fmaxf(q[h].x, -43.f), 43.f)), __expf(2.f * fminf(fmaxf(q[h].y, -43.f), 43.f)),
                                                  __expf(2.f * fminf(fmaxf(q[h].z, -43.f), 43.f)), __expf(2.f * fminf(fmaxf(q[h].w, -43.f), 43.f)));
                    float4 sacc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                    for (int i = 0; i < PSG; ++i) {
                        float r;
                        r = __builtin_amdgcn_rcpf(fmaf(Pr[i][h].x, eq.x, 1.f)); sacc.x += dsc[i] * (r - r * r);
                        r = __builtin_amdgcn_rcpf(fmaf(Pr[i][h].y, eq.y, 1.f)); sacc.y += dsc[i] * (r - r * r);
                        r = __builtin_amdgcn_rcpf(fmaf(Pr[i][h].z, eq.z, 1.f)); sacc.z += dsc[i] * (r - r * r);
                        r = __builtin_amdgcn_rcpf(fmaf(Pr[i][h].w, eq.w, 1.f)); sacc.w += dsc[i] * (r - r * r);
                    }
                    const float4 a4 = *reinterpret_cast<const float4*>(sal + 32 * lr + 4 * h);
                    // straight to the cross-row reduction buffer (keeps the register footprint under the 256 architectural VGPRs)
                    *reinterpret_cast<float4*>(red2 + grow_ * PH + 32 * lr + 4 * h) =
                        make_float4(4.f * a4.x * sacc.x, 4.f * a4.y * sacc.y, 4.f * a4.z * sacc.z, 4.f * a4.w * sacc.w);
                }
                {   // d score of this row's slots, kept for the post-recurrence pass (d P_all, d alpha)
                    const float xw = lr == 0 ? dsc[0] : (lr == 1 ? dsc[1] : dsc[2]);
                    const int sl = grow_ + 16 * lr;
                    if (lr < PSG && sl < PSL && PSL * ap + sl < alen) P.DSC[((long)t * N + an) * P.A + PSL * ap + sl] = xw;
                }
                __syncthreads();
                float* xq = P.XDQ + (long)t * PROWS * PH;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int j = tid + 256 * h;
                    float sum = 0.f;
#pragma unroll
                    for (int g = 0; g < 16; ++g) sum += red2[g * PH + j];
                    atomicAdd(xq + ((j >> 4) * PROWS + an) * 16 + (j & 15), sum);
                }
            }
            if (srole >= 0) BSTAMP(srole, 11);
            publish(cnt(CB_DQ, t));
            if (srole >= 0) BSTAMP(srole, 12);
        }
    }
}


// ==========================================================================================================================
// Reverse attention chain, version 2: two half-chip machines of 32 rows (see the forward version 2).  Per half: 16 gate-gradient
// workgroups of 32 hidden units, 64 product workgroups (16 column tiles of 32 x 4 k-slices of 512 gate columns, two 64-KB images each:
// W_ih1[:, E:]^T and W_hh1^T), 16 attention-only; all 96 hold the attention operands of the half's 32 events.
// ==========================================================================================================================
constexpr int HGD = 16, HP = 64;
constexpr int LDS_REDB2 = 16 * 1024 + 2048 + 2048 + 512;
constexpr int LDS_BYTES_ATTB2 = LDS_W2 + LDS_REDB2 + 256;
constexpr int LDS_BYTES_FWD = LDS_BYTES_ATT2 > LDS_BYTES_LSTM + 1024 ? LDS_BYTES_ATT2 : LDS_BYTES_LSTM + 1024;       // one-launch pairs: the larger of the two roles
constexpr int LDS_BYTES_BWD = LDS_BYTES_ATTB2 > LDS_BYTES_LSTM ? LDS_BYTES_ATTB2 : LDS_BYTES_LSTM;
constexpr long XSTEPH = 4L * HR * PH;          // floats of one half's [4 gates][32 rows x 512] exchange operand

struct PersistLayoutB2 { long cnt, xdq, xda, xdh, zero_begin, xdg, total; };
static PersistLayoutB2 persist_layout_b2(int S);
long persist_bwd_ws_floats(int S) { return persist_layout_b(S).total + persist_layout_b2(S).total; }
static PersistLayoutB2 persist_layout_b2(int S) {
    PersistLayoutB2 L;
    long off = 0;
    auto take = [&](long n) { long o = off; off += (n + 63) / 64 * 64; return o; };
    L.xdg = take((long)S * 2 * XSTEPH);
    L.zero_begin = off;          // zeroed region last, the version-1 layout (counters first) follows: one memset
    L.cnt = take((long)4 * (S + 1) * 2 * CNT_LINE);
    L.xdq = take((long)S * PROWS * PH);
    L.xda = take((long)S * PROWS * PH);
    L.xdh = take((long)S * PROWS * PH);
    L.total = off;
    return L;
}

// B image of one 32-column tile whose source is k-strided: element (k, cc) = W[k * ld_k + cc] for k < K, cc < ncols; zero elsewhere
__device__ __forceinline__ void fill_bimg32_t(float4* img, const float* W, long ld_k, int K, int ncols, int tid) {
    for (int idx = tid; idx < 4 * 16 * 64; idx += 256) {
        const int lane = idx & 63, c = (idx >> 6) & 15, w = idx >> 10;
        const int cc = lane & 31, kh = lane >> 5;
        const int k = 128 * w + 8 * c + 4 * kh;
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = (k + j < K && cc < ncols) ? W[(long)(k + j) * ld_k + cc] : 0.f;
        img[idx] = make_float4(v[0], v[1], v[2], v[3]);
    }
}

template <bool BIG>          // BIG: events of up to 258 segments: e^{2p} of both slot sets in registers, C3D rows streamed from L2 (see dec_persist_att2_body)
__device__ __forceinline__ void dec_persist_att_bwd2_body(const PersistB& P, const int bid) {
    if (P.stamps && bid == 0 && threadIdx.x == 0) P.stamps[15] = __builtin_amdgcn_s_memrealtime();          // kernel entry (diagnostic)
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    float4* wimg = reinterpret_cast<float4*>(lds);
    float* red = reinterpret_cast<float*>(lds + LDS_W2);          // 16 KB
    float* sal = red + 4096;                                      // [512] alpha
    float* sat = sal + PH;                                        // [512] saved context row of the current step
    int* flag = reinterpret_cast<int*>(lds + LDS_W2 + LDS_REDB2);
    const int b = bid, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int m = b / HWG, lb = b - m * HWG;
    const int N = P.N, D = P.D, S = P.S;
    if (HR * m >= N) return;
    const bool is_gd = lb < HGD, is_p = lb >= HGD && lb < HGD + HP;
    auto cnt = [&](int kind, int t) { return P.cnt + (((long)kind * (S + 1) + t) * 2 + m) * CNT_LINE; };
    const int pct = (lb - HGD) & 15, pks = (lb - HGD) >> 4;       // product role: 32-column tile, k-slice (= gate)
    if (is_gd) {
        fill_bimg32_t(wimg, P.w_h2a + 32 * lb, PH, PH, 32, tid);                      // d h1[:, u] += sum_j d q[:, j] W_h2a[j, u]
    } else if (is_p) {
        fill_bimg32_t(wimg, P.w_att + (long)pks * PH * P.ld_att + 32 * pct, P.ld_att, PH, max(0, min(32, D - 32 * pct)), tid);
        fill_bimg32_t(wimg + 4096, P.w_hh[1] + (long)pks * PH * PH + 32 * pct, PH, PH, 32, tid);
    }
    const int ar = lb / 3, ap = lb - 3 * ar, an = HR * m + ar;
    const bool att_live = an < N;
    int grow_ = 4 * w + (lane >> 4), lr = lane & 15;          // (not const: see the step loop of the BIG instantiation)
    int alen = 0;
    long row0 = 0;
    float4 Pr[PSG][8], Cr[BIG ? 1 : PSG][8], Pr2[BIG ? PSG : 1][8];
    for (int j = tid; j < PH; j += 256) sal[j] = P.w_alpha[j];
    auto exp2x = [](const float4 pv) {
        return make_float4(__expf(2.f * fminf(fmaxf(pv.x, -43.f), 43.f)), __expf(2.f * fminf(fmaxf(pv.y, -43.f), 43.f)),
                           __expf(2.f * fminf(fmaxf(pv.z, -43.f), 43.f)), __expf(2.f * fminf(fmaxf(pv.w, -43.f), 43.f)));
    };
    if (att_live) {
        alen = P.ev_len[an];
        row0 = P.ev_start[an];
#pragma unroll
        for (int i = 0; i < PSG; ++i) {
            const int sl = grow_ + 16 * i;
            const int a = min(PSL * ap + min(sl, PSL - 1), alen - 1);
            const float* pr = P.PALL + (row0 + a) * PH + 32 * lr;
            const float* cr = P.c3d + (row0 + a) * D;
#pragma unroll
            for (int h = 0; h < 8; ++h) {
                Pr[i][h] = exp2x(*reinterpret_cast<const float4*>(pr + 4 * h));
                if constexpr (!BIG) {
                    const int d = 32 * lr + 4 * h;
                    float4 v = *reinterpret_cast<const float4*>(cr + min(d, D - 4));
                    if (d >= D) v = make_float4(0.f, 0.f, 0.f, 0.f);
                    Cr[i][h] = v;
                }
            }
            if constexpr (BIG) {
                const int a2 = min(PSET2 + PSL * ap + min(sl, PSL - 1), alen - 1);          // (clamped: only read by events longer than 129 segments)
                const float* pr2 = P.PALL + (row0 + a2) * PH + 32 * lr;
#pragma unroll
                for (int h = 0; h < 8; ++h) Pr2[i][h] = exp2x(*reinterpret_cast<const float4*>(pr2 + 4 * h));
            }
        }
    } else {
#pragma unroll
        for (int i = 0; i < PSG; ++i)
#pragma unroll
            for (int h = 0; h < 8; ++h) {
                Pr[i][h] = make_float4(0.f, 0.f, 0.f, 0.f);
                if constexpr (!BIG) Cr[i][h] = make_float4(0.f, 0.f, 0.f, 0.f);
                else Pr2[i][h] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
    }
    __syncthreads();
    const bool has2 = BIG && att_live && alen > PSET2;           // uniform over the workgroup

    const int gr = tid >> 3, u4 = 4 * (tid & 7), gn = HR * m + gr, u0 = 32 * lb + u4;      // gate-gradient ownership (GD): row, 4 units
    float4 dc = make_float4(0.f, 0.f, 0.f, 0.f);
    const u32 XBH = HR * PH * 4;
    const long XHALF = (long)HR * PH;
    const int srole = b == 0 ? 0 : (b == HGD ? 1 : (b == HGD + HP ? 2 : -1));
    if (P.stamps && b == 0 && tid == 0) P.stamps[14] = __builtin_amdgcn_s_memrealtime();          // set-up done

    for (int t = S - 1; t >= -1; --t) {
        if constexpr (BIG) asm volatile("" : "+v"(grow_), "+v"(lr), "+v"(alen), "+v"(row0));          // see dec_persist_att2_body
        if (srole >= 0) BSTAMP(srole, 0);
        // ============ GD: d h1(t) -> d G1(t); at t = -1 only d q(0) is copied out ============
        if (is_gd) {
            GradIn in;
            DropM dm;
            if (t >= 0) { in = load_grad_in(P, 1, t, gn, u0); dm = drop_masks4(P, 1, t, gn, u0); }
            float4 rec = make_float4(0.f, 0.f, 0.f, 0.f);
            if (t < S - 1) {
                if (!wait_total(P, cnt(CB_DQ, t + 1), HWG, flag, 400000u + t + 1)) return;
                if (srole >= 0) BSTAMP(srole, 1);
                f32x16 acc;
#pragma unroll
                for (int g = 0; g < 16; ++g) acc[g] = 0.f;
                {
                    float4 a[16];
                    load_afrag32(a, mk_rsrc(P.XDQ + ((long)(t + 1) * 2 + m) * XHALF, XBH), w, lane);
                    if (t >= 0) mfma_tile32(acc, a, wimg + w * 1024, lane);
                    // d q(t+1) row-major for the weight-gradient products: this workgroup stores columns [32 lb, 32 lb + 32)
                    if (w == (lb >> 2)) {
#pragma unroll
                        for (int c = 0; c < 16; ++c)
                            if ((c >> 2) == (lb & 3)) {
                                const int n = HR * m + (lane & 31);
                                if (n < N) *reinterpret_cast<float4*>(P.DQ + ((long)(t + 1) * N + n) * PH + 128 * w + 8 * c + 4 * (lane >> 5)) = a[c];
                            }
                    }
                }
                if (t >= 0) {
                    acc_to_lds32(acc, red, w, lane);
                    __syncthreads();
                    const float* rp = red + gr * 32 + u4;
#pragma unroll
                    for (int ww = 0; ww < 4; ++ww) rec = f4add(rec, *reinterpret_cast<const float4*>(rp + ww * HR * 32));
                    if (srole >= 0) BSTAMP(srole, 2);
                    // + d G1(t+1) . W_hh1: the product workgroups form that tile behind their attention role of step t+1 (it is off their
                    // critical path there), so it lands while the d q product above runs
                    if (!wait_total(P, cnt(CB_HH, t + 1), HP, flag, 500000u + t + 1)) return;
                    rec = f4add(rec, ld16_sc1(mk_rsrc(P.XDH + ((long)(t + 1) * 2 + m) * XHALF, XBH), (u32)((((4 * lb + (u4 >> 3)) * HR + gr) * 8 + (u4 & 7)) * 4)));
                    if (srole >= 0) BSTAMP(srole, 3);
                }
            }
            if (t >= 0) {
                const CellGrad cgd = cell_grad4(P, in, dm, rec, dc);
                dc = cgd.dc;
                // exchange layout of this half: [gate][unit / 8][32 rows][8]
                const __amdgpu_buffer_rsrc_t rx = mk_rsrc(P.XDG + ((long)t * 2 + m) * XSTEPH, (u32)(XSTEPH * 4));
#pragma unroll
                for (int g = 0; g < 4; ++g) st16_sc1(rx, (u32)((((g * 64 + 4 * lb + (u4 >> 3)) * HR + gr) * 8 + (u4 & 7)) * 4), cgd.dg[g]);
                if (srole >= 0) BSTAMP(srole, 4);
                publish(cnt(CB_DG, t));
                if (srole >= 0) BSTAMP(srole, 5);
                if (gn < N) {
                    float* dgp = P.DG[1] + ((long)t * N + gn) * 4 * PH + u0;
#pragma unroll
                    for (int g = 0; g < 4; ++g) *reinterpret_cast<float4*>(dgp + g * PH) = cgd.dg[g];
                }
            }
        }
        if (t < 0) break;
        // ============ P: d ATT(t) tile (critical) now; the d h1 tile of the same fragments behind the attention role ============
        float4 pa[16];
        auto p_tile = [&](int which) {
            f32x16 acc;
#pragma unroll
            for (int g = 0; g < 16; ++g) acc[g] = 0.f;
            mfma_tile32(acc, pa, wimg + which * 4096 + w * 1024, lane);
            acc_to_lds32(acc, red, w, lane);
            __syncthreads();
            // element e of the tile in exchange order ([k8 of the tile][row][8]) is contiguous over the lanes: full-rate atomics
            float* xo = (which == 0 ? P.XDA : P.XDH) + ((long)t * 2 + m) * XHALF + (long)4 * pct * HR * 8;
#pragma unroll
            for (int e4 = 0; e4 < 4; ++e4) {
                const int e = tid + 256 * e4, k8l = e >> 8, r = (e >> 3) & 31, cc = 8 * k8l + (e & 7), o = r * 32 + cc;
                atomicAdd(xo + e, red[o] + red[HR * 32 + o] + red[2 * HR * 32 + o] + red[3 * HR * 32 + o]);
            }
            if (which == 0 && srole >= 0) BSTAMP(srole, 7);
            publish(cnt(which == 0 ? CB_DA : CB_HH, t));
            if (which == 0 && srole >= 0) BSTAMP(srole, 8);
        };
        if (is_p) {
            if (!wait_total(P, cnt(CB_DG, t), HGD, flag, 600000u + t)) return;
            if (srole >= 0) BSTAMP(srole, 6);
            load_afrag32(pa, mk_rsrc(P.XDG + ((long)t * 2 + m) * XSTEPH + (long)pks * XHALF, XBH), w, lane);
            if constexpr (BIG) __builtin_amdgcn_sched_barrier(0);          // all 16 fragment loads ahead of the first MFMA (see dec_persist_att2_body)
            p_tile(0);
        }
        // ============ attention backward of step t (all workgroups) ============
        {
            float4 q[8];
            float wt[PSG], wt2[PSG] = {0.f, 0.f, 0.f};
            if (att_live) {
                const float* qp = P.QS + ((long)t * N + an) * PH + 32 * lr;
#pragma unroll
                for (int h = 0; h < 8; ++h) q[h] = *reinterpret_cast<const float4*>(qp + 4 * h);
#pragma unroll
                for (int i = 0; i < PSG; ++i) {
                    const int sl = grow_ + 16 * i;
                    const bool valid = sl < PSL && PSL * ap + sl < alen;
                    wt[i] = valid ? P.WT[((long)t * N + an) * P.A + PSL * ap + sl] : 0.f;
                    if (has2) {
                        const bool valid2 = sl < PSL && PSET2 + PSL * ap + sl < alen;
                        wt2[i] = valid2 ? P.WT[((long)t * N + an) * P.A + PSET2 + PSL * ap + sl] : 0.f;
                    }
                }
                for (int d = tid; d < PH; d += 256) sat[d] = d < D ? P.ATT[((long)t * N + an) * D + d] : 0.f;
            }
            // BIG: the C3D rows of both slot sets are streamed from L2 through NCBB row buffers
            float4 cbuf[BIG ? NCBB : 1][8];
            const __amdgpu_buffer_rsrc_t rc3 = mk_rsrc(P.c3d, P.c3d_bytes);
            auto fetch_c = [&](int g, float4 (&dst)[8]) {
                const int sl = grow_ + 16 * (g % PSG);
                const int a = min(PSET2 * (g / PSG) + PSL * ap + min(sl, PSL - 1), alen - 1);
                const u32 off = (u32)(((row0 + a) * D + 32 * lr) * 4);
#pragma unroll
                for (int h = 0; h < 8; ++h) dst[h] = ld16_plain(rc3, off + 16 * h);
            };
            const int ng = has2 ? 2 * PSG : PSG;
            if (!wait_total(P, cnt(CB_DA, t), HP, flag, 700000u + t)) return;
            if (srole >= 0) BSTAMP(srole, 10);
            if (att_live) {
                const __amdgpu_buffer_rsrc_t ra = mk_rsrc(P.XDA + ((long)t * 2 + m) * XHALF, XBH);
                float4 da[8];
                float s0 = 0.f;
#pragma unroll
                for (int h = 0; h < 8; ++h) da[h] = ld16_sc1(ra, (u32)((((4 * lr + (h >> 1)) * HR + ar) * 8 + 4 * (h & 1)) * 4));
                if constexpr (BIG) {
                    // the first C3D rows go out behind the d ATT loads (requested ahead of the wait they would share the hand-off path with
                    // the product workgroups' fragment fetch, which is on the critical path: d ATT tile 2.6 -> 5.9 us)
#pragma unroll
                    for (int g = 0; g < NCBB; ++g) fetch_c(g, cbuf[g]);
                }
#pragma unroll
                for (int h = 0; h < 8; ++h) {
                    const float4 at = *reinterpret_cast<const float4*>(sat + 32 * lr + 4 * h);
                    s0 += at.x * da[h].x + at.y * da[h].y + at.z * da[h].z + at.w * da[h].w;
                }
                s0 = row16_sum(s0);
                float dsc[PSG], dsc2[PSG] = {0.f, 0.f, 0.f};
                if constexpr (!BIG) {
#pragma unroll
                    for (int i = 0; i < PSG; ++i) {
                        float dw = 0.f;
#pragma unroll
                        for (int h = 0; h < 8; ++h) dw += Cr[i][h].x * da[h].x + Cr[i][h].y * da[h].y + Cr[i][h].z * da[h].z + Cr[i][h].w * da[h].w;
                        dw = row16_sum(dw);
                        dsc[i] = wt[i] * (dw - s0);
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < PSG; ++i) dsc[i] = 0.f;
#pragma unroll
                    for (int g = 0; g < 2 * PSG; ++g) {
                        if (g < ng) {
                            float dw = 0.f;
#pragma unroll
                            for (int h = 0; h < 8; ++h) {
                                float4 c4 = cbuf[g % NCBB][h];
                                if (32 * lr + 4 * h >= D) c4 = make_float4(0.f, 0.f, 0.f, 0.f);
                                dw += c4.x * da[h].x + c4.y * da[h].y + c4.z * da[h].z + c4.w * da[h].w;
                            }
                            if (g + NCBB < ng) fetch_c(g + NCBB, cbuf[g % NCBB]);
                            dw = row16_sum(dw);
                            if (g < PSG) dsc[g % PSG] = wt[g % PSG] * (dw - s0);
                            else dsc2[g % PSG] = wt2[g % PSG] * (dw - s0);
                        }
                    }
                }
                {
                    const float xw = lr == 0 ? dsc[0] : (lr == 1 ? dsc[1] : dsc[2]);
                    const int sl = grow_ + 16 * lr;
                    if (lr < PSG && sl < PSL && PSL * ap + sl < alen) P.DSC[((long)t * N + an) * P.A + PSL * ap + sl] = xw;
                }
                // second slot set: its share of d q from its own e^{2p} registers
                float4 s2[8];
#pragma unroll
                for (int h = 0; h < 8; ++h) s2[h] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (has2) {
                    {
                        const float xw2 = lr == 3 ? dsc2[0] : (lr == 4 ? dsc2[1] : dsc2[2]);
                        const int sl2 = grow_ + 16 * (lr - 3);
                        if (lr >= 3 && lr < 3 + PSG && sl2 < PSL && PSET2 + PSL * ap + sl2 < alen) P.DSC[((long)t * N + an) * P.A + PSET2 + PSL * ap + sl2] = xw2;
                    }
                    if constexpr (BIG) {
#pragma unroll
                        for (int h = 0; h < 8; ++h) {
                            const float4 eq = exp2x(q[h]);
#pragma unroll
                            for (int i = 0; i < PSG; ++i) {
                                float r;
                                r = __builtin_amdgcn_rcpf(fmaf(Pr2[i][h].x, eq.x, 1.f)); s2[h].x += dsc2[i] * (r - r * r);
                                r = __builtin_amdgcn_rcpf(fmaf(Pr2[i][h].y, eq.y, 1.f)); s2[h].y += dsc2[i] * (r - r * r);
                                r = __builtin_amdgcn_rcpf(fmaf(Pr2[i][h].z, eq.z, 1.f)); s2[h].z += dsc2[i] * (r - r * r);
                                r = __builtin_amdgcn_rcpf(fmaf(Pr2[i][h].w, eq.w, 1.f)); s2[h].w += dsc2[i] * (r - r * r);
                            }
                        }
                    }
                }
                // d q partial of this DPP row (in place of q), then summed over the 16 rows through 16 KB of LDS: rows 0-7, then rows 8-15
#pragma unroll
                for (int h = 0; h < 8; ++h) {
                    const float4 eq = make_float4(__expf(2.f * fminf(fmaxf(q[h].x, -43.f), 43.f)), __expf(2.f * fminf(fmaxf(q[h].y, -43.f), 43.f)),
                                                  __expf(2.f * fminf(fmaxf(q[h].z, -43.f), 43.f)), __expf(2.f * fminf(fmaxf(q[h].w, -43.f), 43.f)));
                    float4 sacc = BIG ? s2[h] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                    for (int i = 0; i < PSG; ++i) {
                        float r;
                        r = __builtin_amdgcn_rcpf(fmaf(Pr[i][h].x, eq.x, 1.f)); sacc.x += dsc[i] * (r - r * r);
                        r = __builtin_amdgcn_rcpf(fmaf(Pr[i][h].y, eq.y, 1.f)); sacc.y += dsc[i] * (r - r * r);
                        r = __builtin_amdgcn_rcpf(fmaf(Pr[i][h].z, eq.z, 1.f)); sacc.z += dsc[i] * (r - r * r);
                        r = __builtin_amdgcn_rcpf(fmaf(Pr[i][h].w, eq.w, 1.f)); sacc.w += dsc[i] * (r - r * r);
                    }
                    const float4 a4 = *reinterpret_cast<const float4*>(sal + 32 * lr + 4 * h);
                    q[h] = make_float4(4.f * a4.x * sacc.x, 4.f * a4.y * sacc.y, 4.f * a4.z * sacc.z, 4.f * a4.w * sacc.w);
                }
                float qsum[2] = {0.f, 0.f};
#pragma unroll
                for (int pass = 0; pass < 2; ++pass) {
                    if ((grow_ >> 3) == pass) {
#pragma unroll
                        for (int h = 0; h < 8; ++h) *reinterpret_cast<float4*>(red + (grow_ & 7) * PH + 32 * lr + 4 * h) = q[h];
                    }
                    __syncthreads();
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const int j = tid + 256 * h;
#pragma unroll
                        for (int g = 0; g < 8; ++g) qsum[h] += red[g * PH + j];
                    }
                    __syncthreads();
                }
                float* xq = P.XDQ + ((long)t * 2 + m) * XHALF;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int j = tid + 256 * h;
                    atomicAdd(xq + ((j >> 3) * HR + ar) * 8 + (j & 7), qsum[h]);
                }
            }
            if (srole >= 0) BSTAMP(srole, 11);
            publish(cnt(CB_DQ, t));
            if (srole >= 0) BSTAMP(srole, 12);
        }
        // the d h1 tile (W_hh1 columns) of the fragments fetched above: its consumers (the gate-gradient workgroups of step t-1) take
        // it after their d q product, so it runs here instead of holding back this workgroup's attention role
        if (is_p && t > 0) {
            // BIG: the fragments are fetched again here (the tile is off the critical path) instead of staying live across the attention role --
            // their 64 registers hold the second e^{2p} set
            if constexpr (BIG) {
                load_afrag32(pa, mk_rsrc(P.XDG + ((long)t * 2 + m) * XSTEPH + (long)pks * XHALF, XBH), w, lane);
                __builtin_amdgcn_sched_barrier(0);
            }
            p_tile(1);
            if (srole >= 0) BSTAMP(srole, 9);
        }
    }
}
__global__ __launch_bounds__(256, 1) void dec_persist_att_bwd2_kernel(PersistB P) { dec_persist_att_bwd2_body<false>(P, blockIdx.x); }

// ---- the reverse pair as one launch (see dec_persist_fwd_kernel) ----
template <bool BIG>
__global__ __launch_bounds__(256, 1) void dec_persist_bwd_kernel(PersistB P2, PersistB P1) {
    const int rb = persist_role_index(blockIdx.x, P2.xcd_map);
    if (rb < 2 * HWG) dec_persist_att_bwd2_body<BIG>(P2, rb);
    else if (P1.lstm_kgroups) dec_persist_lstm_bwd_kg_body(P1, rb - 2 * HWG);
    else dec_persist_lstm_bwd_body(P1, rb - 2 * HWG);
}

// ---- host side -----------------------------------------------------------------------------------------------------------
struct PersistHost { u32* abort_dev = nullptr; u32* flag_host = nullptr; u32* flag_dev = nullptr; int cus = 0; bool ok = false; bool init = false; unsigned long long* stamps = nullptr; int stamps_S = 0;
                     hipStream_t side = nullptr; hipEvent_t fork = nullptr, join = nullptr; };
// one state block per device, created once (std::call_once: autograd's device threads or a user thread may make the first call concurrently)
constexpr int MAX_DEVICES = 16;
static void phost_init(PersistHost& h, int dev) {
    {
        h.init = true;
        hipDeviceProp_t prop;
        bool good = hipGetDeviceProperties(&prop, dev) == hipSuccess;
        if (good) h.cus = prop.multiProcessorCount;
        good = good && hipMalloc(&h.abort_dev, 256) == hipSuccess && hipMemset(h.abort_dev, 0, 256) == hipSuccess;
        good = good && hipHostMalloc(&h.flag_host, 64, hipHostMallocMapped) == hipSuccess;
        if (good) { h.flag_host[0] = 0; good = hipHostGetDevicePointer((void**)&h.flag_dev, h.flag_host, 0) == hipSuccess; }
        good = good && hipFuncSetAttribute(reinterpret_cast<const void*>(dec_persist_att_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES_ATT) == hipSuccess;
        good = good && hipFuncSetAttribute(reinterpret_cast<const void*>(dec_persist_lstm_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES_LSTM) == hipSuccess;
        good = good && hipFuncSetAttribute(reinterpret_cast<const void*>(dec_persist_lstm_h2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES_LSTM + 1024) == hipSuccess;
        good = good && hipFuncSetAttribute(reinterpret_cast<const void*>(dec_persist_att2_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES_ATT2) == hipSuccess;
        good = good && hipFuncSetAttribute(reinterpret_cast<const void*>(dec_persist_att2_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES_ATT2) == hipSuccess;
        good = good && hipFuncSetAttribute(reinterpret_cast<const void*>(dec_persist_att_bwd2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES_ATTB2) == hipSuccess;
        good = good && hipFuncSetAttribute(reinterpret_cast<const void*>(dec_persist_att_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES_ATT) == hipSuccess;
        good = good && hipFuncSetAttribute(reinterpret_cast<const void*>(dec_persist_lstm_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES_LSTM) == hipSuccess;
        good = good && hipFuncSetAttribute(reinterpret_cast<const void*>(dec_persist_fwd_kernel<false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES_FWD) == hipSuccess;
        good = good && hipFuncSetAttribute(reinterpret_cast<const void*>(dec_persist_fwd_kernel<true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES_FWD) == hipSuccess;
        good = good && hipFuncSetAttribute(reinterpret_cast<const void*>(dec_persist_fwd_kernel<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES_FWD) == hipSuccess;
        good = good && hipFuncSetAttribute(reinterpret_cast<const void*>(dec_persist_fwd_kernel<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES_FWD) == hipSuccess;
        good = good && hipFuncSetAttribute(reinterpret_cast<const void*>(dec_persist_sample_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES_FWD) == hipSuccess;
        good = good && hipFuncSetAttribute(reinterpret_cast<const void*>(dec_persist_sample_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES_FWD) == hipSuccess;
        good = good && hipFuncSetAttribute(reinterpret_cast<const void*>(dec_persist_bwd_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES_BWD) == hipSuccess;
        good = good && hipFuncSetAttribute(reinterpret_cast<const void*>(dec_persist_bwd_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES_BWD) == hipSuccess;
        (void)hipGetLastError();
        h.ok = good;
    }
}
static PersistHost& phost() {
    static PersistHost hosts[MAX_DEVICES + 1];          // [MAX_DEVICES]: "no such device" (ok stays false)
    static std::once_flag once[MAX_DEVICES];
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEVICES) { (void)hipGetLastError(); return hosts[MAX_DEVICES]; }
    std::call_once(once[dev], phost_init, std::ref(hosts[dev]), dev);
    return hosts[dev];
}
// the second stream of the two-launch form, created on first use only: every HIP stream of a process competes for the runtime's few
// hardware queues (a fifth active stream made the whole iteration 2x slower on this path), and the default one-launch form needs none
static bool side_stream(PersistHost& h) {
    if (!h.side) {
        int lo = 0, hi = 0;
        bool good = hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess;
        good = good && hipStreamCreateWithPriority(&h.side, hipStreamNonBlocking, hi) == hipSuccess;
        good = good && hipEventCreateWithFlags(&h.fork, echr::sync_event_flags()) == hipSuccess;
        good = good && hipEventCreateWithFlags(&h.join, echr::sync_event_flags()) == hipSuccess;
        if (!good) { (void)hipGetLastError(); h.side = nullptr; }
    }
    return h.side != nullptr;
}

void coop_refused(const char* who, const char* why) {
    static bool said = false;
    if (!said) { said = true; fprintf(stderr, "[libechr_hip] %s: cooperative launch refused (%s); falling back to the plain launch\n", who, why); }
}

// sticky asynchronous error of an earlier persistent launch (a bounded spin timed out): reported once, at the next library call
const unsigned* persist_abort_word() {
    PersistHost& h = phost();
    return h.ok ? h.abort_dev : nullptr;
}

unsigned* persist_host_flag() {
    PersistHost& h = phost();
    return h.ok ? h.flag_dev : nullptr;
}

static std::atomic<long long> g_skipped_updates{0};
long long persist_take_skipped_updates() { return g_skipped_updates.exchange(0); }

int persist_check_async() {
    // The library's helper streams, events and the abort word (phost / tail / prep / side) are process-wide and live on the device that was
    // current at the first call: a call from another device would mix foreign-device streams and memory, so it is refused.
    // (the persistent kernels' own state above is per device; the helper streams of decoder.hip / tsrm.hip / sst.hip are not)
    static std::atomic<int> first_dev{-1};
    int dev = -1;
    if (hipGetDevice(&dev) == hipSuccess) {
        int expect = -1;
        if (!first_dev.compare_exchange_strong(expect, dev) && expect != dev) {
            set_error("libechr_hip.so keeps per-process helper state on device %d (one process per GPU); the current device is %d", expect, dev);
            return -18;   // -EXDEV
        }
    }
    PersistHost& h = phost();
    if (h.ok && h.flag_host[0]) {
        const u32 code = h.flag_host[0];
        h.flag_host[0] = 0;
        // error path: let everything queued behind the aborted launch retire (the optimiser kernels among it skip and count themselves),
        // then read the count and clear the block
        (void)hipDeviceSynchronize();
        u32 skipped = 0;
        if (hipMemcpy(&skipped, h.abort_dev + ABORT_SKIPPED_WORD, sizeof(u32), hipMemcpyDeviceToHost) != hipSuccess) { (void)hipGetLastError(); skipped = 0; }
        g_skipped_updates.fetch_add(skipped);
        (void)hipMemset(h.abort_dev, 0, 256);
        // codes: attention chain 100000 * edge + timestep (edge 1 h1, 2 q, 3 context; reverse: 4 d q, 5 d h, 6 d G, 7 d ATT); plain LSTM streams
        // 1000 / 5000 / 7000 * (counter kind + 1) + timestep; 9000 + timestep: the softmax max exchange
        set_error("persistent decoder kernel aborted: a hand-off wait timed out (code %u, timestep %u); the outputs of that call and of every library "
                  "call enqueued since are invalid, and the optimiser kernels enqueued since were skipped (parameters unchanged)", code, code % 1000);
        return -62;   // -ETIME
    }
    return 0;
}

// diagnostic: copy the last stamped launch's s_memrealtime stamps ([4 roles][S][16] uint64, 100 MHz) to the host; returns S
int persist_read_stamps(unsigned long long* dst, int max_entries) {
    PersistHost& h = phost();
    if (!h.stamps || h.stamps_S <= 0 || max_entries < 4 * h.stamps_S * 16) return 0;
    if (hipDeviceSynchronize() != hipSuccess) return 0;
    if (hipMemcpy(dst, h.stamps, (size_t)4 * h.stamps_S * 16 * 8, hipMemcpyDeviceToHost) != hipSuccess) return 0;
    return h.stamps_S;
}

// diagnostic stamp buffer for other persistent kernels (csrc/sst.hip): zeroed on `st`, read back with persist_read_stamps
unsigned long long* persist_stamp_buffer(int S, hipStream_t st) {
    PersistHost& h = phost();
    if (!h.ok) return nullptr;
    if (!h.stamps && hipMalloc(&h.stamps, 4 * 256 * 16 * 8) != hipSuccess) { h.stamps = nullptr; return nullptr; }
    h.stamps_S = S < 256 ? S : 256;
    (void)hipMemsetAsync(h.stamps, 0, 4 * 256 * 16 * 8, st);
    return h.stamps;
}

static bool persist_shape_ok(const echr_dec_args* a) {
    PersistHost& h = phost();
    // A <= 129: every form; 129 < A <= 258: the one-launch pairs of two half-chip machines only (their BIG instantiations: second slot set)
    const bool a_ok = a->A <= PSET2 || (a->A <= 2 * PSET2 && config().persist_split && config().persist_merge);
    return h.ok && h.cus >= NWG && a->N <= PROWS && a_ok && a->H == PH && a->Ha == PH && a->D <= PH && a->D % 4 == 0 && a->D >= 8 &&
           a->S >= 1 && !a->h0;          // (a non-zero initial state: the persistent kernels assume h(-1) = c(-1) = 0)
}

bool persist_fwd_eligible(const echr_dec_args* a) { return config().persist && !det_mode() && persist_shape_ok(a); }          // (fixed-order mode: the exchange adds of the persistent kernels are atomics)

// the part of the exchange workspace the persistent launch needs zeroed (counters, atomically accumulated buffers): callers that run a
// multi-range fill anyway fold it in (PersistFwdBufs::prezeroed / PersistBwdBufs::prezeroed) instead of paying a memset launch
void persist_fwd_zero_range(const echr_dec_args* a, float* xws, float** ptr, long* count) {
    const PersistLayout L = persist_layout(a->S);
    const PersistLayout2 L2 = persist_layout2(a->S);
    if (config().persist_split) { *ptr = xws + L2.zero_begin; *count = L2.total - L2.zero_begin + L.xc; }
    else { *ptr = xws; *count = L.zero_end; }
}

static const float*& prebuilt_for() { static const float* p = nullptr; return p; }          // workspace whose images persist_fwd_prebuild queued last
int persist_fwd_prebuild(const echr_dec_args* a, float* xws, hipStream_t st) {
    prebuilt_for() = nullptr;
    static const bool on = [] { const char* e = getenv("ECHR_PERSIST_PREBUILD"); return !(e && e[0] == '0'); }();      // A/B switch
    if (!on || !persist_fwd_eligible(a) || !config().persist_h2 || !config().persist_split || !config().persist_merge || !xws) return 0;
    const PersistLayout2 L2 = persist_layout2(a->S);
    PersistK2 K2 = {};
    K2.D = a->D; K2.ld_att = a->E + a->D; K2.w_hh1 = a->w_hh[1]; K2.w_h2a = a->w_h2a; K2.w_att = a->w_ih[1] + a->E;
    hipLaunchKernelGGL(dec_persist_prebuild_kernel, dim3(HG1 + HQ), dim3(256), 0, st, K2, xws + L2.pimg);
    if (int rc = check_launch("dec_persist_prebuild")) return rc;
    prebuilt_for() = xws;
    return 0;
}

int persist_fwd(const echr_dec_args* a, const PersistFwdBufs& B, const DropCfg& dh, const DropCfg& dout, hipStream_t st) {
    PersistHost& h = phost();
    ECHR_REQUIRE(h.ok, "persist_fwd: device state unavailable");
    const PersistLayout L = persist_layout(a->S);
    PersistK K;
    K.N = a->N; K.A = a->A; K.D = a->D; K.S = a->S; K.ld_att = a->E + a->D;
    for (int k = 0; k < 3; ++k) { K.w_hh[k] = a->w_hh[k]; K.GATES[k] = B.GATES[k]; K.CS[k] = B.CS[k]; }
    K.w_h2a = a->w_h2a; K.b_h2a = a->b_h2a; K.w_att = a->w_ih[1] + a->E; K.w_alpha = a->w_alpha;
    K.PALL = B.PALL; K.c3d = a->c3d; K.ev_start = a->ev_start; K.ev_len = a->ev_len;
    K.HS = B.HS; K.OUTD = B.OUTD; K.QS = B.QS; K.WT = B.WT; K.ATT = B.ATT;
    K.evb0 = config().persist_h2 ? B.EVB0 : nullptr;          // (only the fp16-pair LSTM role adds it: see persist_fwd_adds_evb0)
    const bool split = config().persist_split != 0;
    const PersistLayout2 L2 = persist_layout2(a->S);
    float* x2 = B.xws;                                  // version-2 layout first (its zeroed region last), version 1 behind it
    float* x = split ? x2 + L2.total : B.xws;
    K.cnt = reinterpret_cast<u32*>(x + L.cnt); K.XC = x + L.xc; K.XS = x + L.xs; K.GRAN = reinterpret_cast<unsigned long long*>(x + L.gran);
    K.XH1 = x + L.xh1; K.XH0 = x + L.xh0; K.XH2 = x + L.xh2; K.XQ = x + L.xq; K.WU = x + L.wu;
    K.abort_word = h.abort_dev; K.host_flag = h.flag_dev;
    K.spin_limit = config().persist_spin_limit > 0 ? (u32)config().persist_spin_limit : SPIN_LIMIT; K.inject = (u32)config().persist_inject_timeout;
    K.dh = dh; K.dout = dout;
    static const int nt_saved = [] { const char* e = getenv("ECHR_NT_SAVED"); return e ? atoi(e) : 1; }();      // A/B switch
    K.nt_saved = nt_saved;
    K.stamps = nullptr;
    if (config().persist_stamps == 1) {
        if (!h.stamps && hipMalloc(&h.stamps, 4 * 256 * 16 * 8) != hipSuccess) h.stamps = nullptr;
        if (h.stamps && a->S <= 256) { K.stamps = h.stamps; h.stamps_S = a->S; (void)hipMemsetAsync(h.stamps, 0, 4 * 256 * 16 * 8, st); }
    }
    PersistK2 K2;
    K2.pimg = nullptr;
    K2.nt_saved = 0;
    static const int xcd_map = [] { const char* e = getenv("ECHR_PERSIST_XCD"); return e ? atoi(e) : 1; }();      // A/B switch (default on, round 6)
    K2.xcd_map = (xcd_map && 2 * HWG + 2 * NS == 256 && HWG == 96) ? 1 : 0;
    if (split) {
        K2.N = K.N; K2.A = K.A; K2.D = K.D; K2.S = K.S; K2.ld_att = K.ld_att;
        K2.w_hh1 = a->w_hh[1]; K2.w_h2a = a->w_h2a; K2.b_h2a = a->b_h2a; K2.w_att = K.w_att; K2.w_alpha = a->w_alpha;
        K2.PALL = B.PALL; K2.c3d = a->c3d; K2.ev_start = a->ev_start; K2.ev_len = a->ev_len; K2.c3d_bytes = (unsigned)((size_t)a->Tv * a->D * 4);
        K2.GATES1 = B.GATES[1]; K2.CS1 = B.CS[1]; K2.HS = B.HS; K2.OUTD = B.OUTD; K2.QS = B.QS; K2.WT = B.WT; K2.ATT = B.ATT;
        K2.cnt = reinterpret_cast<u32*>(x2 + L2.cnt); K2.XC = x2 + L2.xc; K2.XS = x2 + L2.xs; K2.GRAN = reinterpret_cast<unsigned long long*>(x2 + L2.gran);
        K2.XH1 = x2 + L2.xh1; K2.XQ = x2 + L2.xq; K2.WU = x2 + L2.wu; K2.XCMAX = x2 + L2.xcmax;
        K2.abort_word = h.abort_dev; K2.host_flag = h.flag_dev; K2.stamps = K.stamps; K2.dh = dh; K2.dout = dout;
        K2.spin_limit = K.spin_limit; K2.inject = K.inject;
        K2.nt_saved = K.nt_saved;
        K2.pimg = (prebuilt_for() == x2) ? x2 + L2.pimg : nullptr;          // (queued on an earlier point of this launch's stream chain by the decoder's prepare)
        prebuilt_for() = nullptr;
        // one memset: version 2's zeroed region and, right behind it, version 1's counters (all the LSTM kernel needs of that layout)
        if (!B.prezeroed && hipMemsetAsync(x2 + L2.zero_begin, 0, (size_t)(L2.total - L2.zero_begin + L.xc) * sizeof(float), st) != hipSuccess) {
            set_error("persist_fwd: memset failed");
            return -5;
        }
    } else if (!B.prezeroed && hipMemsetAsync(x, 0, (size_t)L.zero_end * sizeof(float), st) != hipSuccess) { set_error("persist_fwd: memset failed"); return -5; }
    // algorithmic bytes of the forward pair: every recurrent weight and every attention operand row once, plus the per-step activations in
    // (input-side gate pre-activations) and out (c, h of three streams, dropped output, q, attention weights, context)
    const double wbytes = 4.0 * (3.0 * 4 * PH * PH + (double)PH * PH + 4.0 * PH * a->D);
    const double obytes = 4.0 * (double)a->N * a->A * (PH + a->D);
    const double sbytes = 4.0 * (double)a->S * a->N * (3.0 * 4 * PH + 3.0 * 2 * PH + PH + PH + a->A + a->D);
    ProfScope prof(PROF_PERSIST, 2.0 * a->S * PROWS * PH * (double)(3 * 4 * PH + PH + 4 * PH), wbytes + obytes + sbytes, st);
    const bool big = a->A > PSET2;          // events longer than 129 segments: BIG instantiation (persist_shape_ok admitted it for this form only)
    if (split && config().persist_merge) {
        // one launch of 192 + 64 workgroups (no dependence on two hardware queues being free)
        if (config().persist_coop) {
            // cooperative launch: the dispatch starts only when all 256 workgroups can be resident together, whatever else holds CUs
            // (collective kernels, another process' grid) -- the hand-off spins then never wait for a workgroup that has no CU
            void* kargs[2] = {&K2, &K};
            const void* fn = big ? (config().persist_h2 ? reinterpret_cast<const void*>(dec_persist_fwd_kernel<true, true>) : reinterpret_cast<const void*>(dec_persist_fwd_kernel<false, true>))
                                 : (config().persist_h2 ? reinterpret_cast<const void*>(dec_persist_fwd_kernel<true, false>) : reinterpret_cast<const void*>(dec_persist_fwd_kernel<false, false>));
            if (hipLaunchCooperativeKernel(fn, dim3(2 * HWG + 2 * NS), dim3(256), kargs, LDS_BYTES_FWD, st) == hipSuccess) return 0;
            // the runtime refused the cooperative form (configuration / driver): say so once and use the plain launch, whose bounded waits
            // still turn a co-residency problem into -ETIME rather than a hang
            coop_refused("persist_fwd", hipGetErrorString(hipGetLastError()));
        }
        if (big) {
            if (config().persist_h2) hipLaunchKernelGGL((dec_persist_fwd_kernel<true, true>), dim3(2 * HWG + 2 * NS), dim3(256), LDS_BYTES_FWD, st, K2, K);
            else hipLaunchKernelGGL((dec_persist_fwd_kernel<false, true>), dim3(2 * HWG + 2 * NS), dim3(256), LDS_BYTES_FWD, st, K2, K);
        } else if (config().persist_h2) hipLaunchKernelGGL((dec_persist_fwd_kernel<true, false>), dim3(2 * HWG + 2 * NS), dim3(256), LDS_BYTES_FWD, st, K2, K);
        else hipLaunchKernelGGL((dec_persist_fwd_kernel<false, false>), dim3(2 * HWG + 2 * NS), dim3(256), LDS_BYTES_FWD, st, K2, K);
        return check_launch("dec_persist_fwd");
    }
    // two launches: the two plain LSTM streams recur on a second HIP stream, concurrently with the attention chain (192 + 64 workgroups = 256 CUs;
    // neither kernel waits on the other, so any residency order makes progress)
    ECHR_REQUIRE(side_stream(h), "persist_fwd: second stream unavailable");
    if (hipEventRecord(h.fork, st) != hipSuccess || hipStreamWaitEvent(h.side, h.fork, 0) != hipSuccess) { set_error("persist_fwd: fork failed"); return -5; }
    if (config().persist_h2) hipLaunchKernelGGL(dec_persist_lstm_h2_kernel, dim3(2 * NS), dim3(256), LDS_BYTES_LSTM + 1024, h.side, K);
    else hipLaunchKernelGGL(dec_persist_lstm_kernel, dim3(2 * NS), dim3(256), LDS_BYTES_LSTM, h.side, K);
    if (int rc = check_launch("dec_persist_lstm")) return rc;
    if (split && config().persist_h2) hipLaunchKernelGGL(dec_persist_att2_kernel<true>, dim3(2 * HWG), dim3(256), LDS_BYTES_ATT2, st, K2);
    else if (split) hipLaunchKernelGGL(dec_persist_att2_kernel<false>, dim3(2 * HWG), dim3(256), LDS_BYTES_ATT2, st, K2);
    else hipLaunchKernelGGL(dec_persist_att_kernel, dim3(NATT), dim3(256), LDS_BYTES_ATT, st, K);
    if (int rc = check_launch("dec_persist_att")) return rc;
    if (hipEventRecord(h.join, h.side) != hipSuccess || hipStreamWaitEvent(st, h.join, 0) != hipSuccess) { set_error("persist_fwd: join failed"); return -5; }
    return 0;
}


// ---- greedy decoding on the persistent kernels (SAMP instantiations) ----
struct PersistLayoutS { long key, cnt, stop, zero_end, lse, xc3, xs3, total; };
static inline int logit_chunks(int V1) { return (V1 + LWG * LCOLS - 1) / (LWG * LCOLS); }
static PersistLayoutS persist_layout_s(int S, int V1) {
    const int nch = logit_chunks(V1);
    PersistLayoutS L;
    long off = 0;
    auto take = [&](long n) { long o = off; off += (n + 63) / 64 * 64; return o; };
    L.key = take((long)S * PROWS * 2);
    L.cnt = take((long)S * CNT_LINE);
    L.stop = take(64);
    L.zero_end = off;
    L.lse = take((long)S * LWG * nch * PROWS * 2);
    L.xc3 = take((long)S * 2 * 3 * HR * PH);
    L.xs3 = take((long)S * PROWS * 3);
    L.total = off;
    return L;
}
long persist_sample_ws_floats(int S, int V1) { return persist_layout_s(S, V1).total; }
long persist_sample_x_floats(int S) { return persist_layout(S).total + persist_layout2(S).total; }          // one group's exchange workspace (both layouts)

// two zero ranges per event group, `stride` floats apart from group to group: blocks [0, per) of a group cover range A then range B
__global__ __launch_bounds__(256) void fill_zero_groups_kernel(float* __restrict__ a, long na, long stride_a, float* __restrict__ b, long nb, long stride_b, int per) {
    const int g = blockIdx.x / per, i = blockIdx.x % per;
    const int ba = (int)((na + 4095) / 4096);
    float* p = i < ba ? a + (long)g * stride_a : b + (long)g * stride_b;
    const long n = i < ba ? na : nb, base = (long)(i < ba ? i : i - ba) * 4096;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const long j = base + k * 256 + threadIdx.x;
        if (j < n) p[j] = 0.f;
    }
}
long persist_logit_image_floats(int V1) { const long nvb = (long)LWG * logit_chunks(V1); return nvb * 3 * 16 * LCT * 2 * 64 * 4 + 2L * nvb * LCOLS; }

// vocabulary within the logits role's 64 x 80 columns, fp16-pair forms on, shapes as the teacher-forced kernel (events are processed 64 per launch)
// ... by shape alone (what the workspace carving goes by: a switch flipped between the size query and the call must not move the carving) ...
bool persist_sample_shape_ok(const echr_dec_args* a) {
    PersistHost& h = phost();
    return h.ok && h.cus >= NWG && a->N >= 1 && a->A <= 2 * PSET2 && a->H == PH && a->Ha == PH && a->D <= PH && a->D % 4 == 0 && a->D >= 8 && a->S >= 1 && a->S <= 64 &&
           a->V1 <= LWG * LCOLS * LCHMAX && a->V1 >= 2 && !a->h0;          // (the persistent decoder starts from the zero state)
}
// ... and with the switches that select it
bool persist_sample_eligible(const echr_dec_args* a) {
    return config().persist && config().persist_sample && config().persist_h2 && config().persist_split && config().persist_merge && config().gemm_h2 &&
           persist_sample_shape_ok(a);
}

int persist_logit_image(const float* w_logit, int V1, float* img, hipStream_t st) {
    const int nvb = LWG * logit_chunks(V1);
    float* sc = img + (long)nvb * 3 * 16 * LCT * 2 * 64 * 4;
    float* inv = sc + nvb * LCOLS;
    hipLaunchKernelGGL(logit_scale_kernel, dim3(nvb * LCOLS / 4), dim3(256), 0, st, w_logit, V1, nvb, sc, inv);
    if (int rc = check_launch("logit_scale")) return rc;
    const long items = (long)nvb * 3 * 16 * LCT * 64;
    hipLaunchKernelGGL(logit_image_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, st, w_logit, V1, nvb, sc, reinterpret_cast<float4*>(img));
    return check_launch("logit_image");
}

// One launch per group of 64 events (back to back on `st`; every group has its own copy of the exchange workspaces, so ONE fill launch ahead
// of the first and ONE finishing launch behind the last serve all of them).  B.xws / B.sws: groups x persist_sample_x_floats / _ws_floats.
static int persist_sample_group(const echr_dec_args* a, const PersistSampleBufs& B, bool stop_early, hipStream_t st);
int persist_sample(const echr_dec_args* a, const PersistSampleBufs& B, hipStream_t st) {
    PersistHost& h = phost();
    ECHR_REQUIRE(h.ok && a->N >= 1, "persist_sample: device state unavailable");
    const PersistLayout L = persist_layout(a->S);
    const PersistLayout2 L2 = persist_layout2(a->S);
    const PersistLayoutS LS = persist_layout_s(a->S, a->V1);
    const int groups = (a->N + PROWS - 1) / PROWS;
    const long sx = persist_sample_x_floats(a->S), ss = LS.total;
    {
        // counters, atomically folded buffers and the arg-max keys of every group
        const long zx = L2.total - L2.zero_begin + L.xc, zs = LS.zero_end;
        const long per = (zx + 4095) / 4096 + (zs + 4095) / 4096;
        hipLaunchKernelGGL(fill_zero_groups_kernel, dim3((unsigned)(per * groups)), dim3(256), 0, st, B.xws + L2.zero_begin, zx, sx, B.sws, zs, ss, (int)per);
        if (int rc = check_launch("fill_zero_groups")) return rc;
    }
    for (int g = 0; g < groups; ++g) {
        echr_dec_args p = *a;
        const int n0 = PROWS * g;
        p.N = a->N - n0 < PROWS ? a->N - n0 : PROWS;
        p.ev_start = a->ev_start + n0; p.ev_len = a->ev_len + n0;
        PersistSampleBufs Bg = B;
        Bg.EVB0 = B.EVB0 + (long)n0 * 4 * PH; Bg.xws = B.xws + (long)g * sx; Bg.sws = B.sws + (long)g * ss;
        if (int rc = persist_sample_group(&p, Bg, groups == 1, st)) return rc;
    }
    hipLaunchKernelGGL(sample_finish_kernel, dim3(a->N), dim3(256), 0, st, reinterpret_cast<const unsigned long long*>(B.sws + LS.key), B.sws + LS.lse, a->N, a->S,
                       logit_chunks(a->V1), ss, B.seq, B.seq_logp, B.n_unfinished, reinterpret_cast<const u32*>(B.sws + LS.stop));
    return check_launch("sample_finish");
}
static int persist_sample_group(const echr_dec_args* a, const PersistSampleBufs& B, bool stop_early, hipStream_t st) {
    PersistHost& h = phost();
    ECHR_REQUIRE(h.ok && a->N <= PROWS, "persist_sample_group: device state unavailable");
    const PersistLayout L = persist_layout(a->S);
    const PersistLayout2 L2 = persist_layout2(a->S);
    const PersistLayoutS LS = persist_layout_s(a->S, a->V1);
    const DropCfg off{0u, 0u, 0u, 0u, 1.f, 0};          // decoding runs in eval mode: every dropout multiplier is 1
    PersistK K;
    K.nt_saved = 0;
    K.N = a->N; K.A = a->A; K.D = a->D; K.S = a->S; K.ld_att = a->E + a->D;
    for (int k = 0; k < 3; ++k) { K.w_hh[k] = a->w_hh[k]; K.GATES[k] = nullptr; K.CS[k] = nullptr; }
    K.w_h2a = a->w_h2a; K.b_h2a = a->b_h2a; K.w_att = a->w_ih[1] + a->E; K.w_alpha = a->w_alpha;
    K.PALL = B.PALL; K.c3d = a->c3d; K.ev_start = a->ev_start; K.ev_len = a->ev_len;
    K.HS = nullptr; K.OUTD = nullptr; K.QS = nullptr; K.WT = nullptr; K.ATT = nullptr; K.evb0 = nullptr;
    float* x2 = B.xws;
    float* x = x2 + L2.total;
    K.cnt = reinterpret_cast<u32*>(x + L.cnt); K.XC = x + L.xc; K.XS = x + L.xs; K.GRAN = reinterpret_cast<unsigned long long*>(x + L.gran);
    K.XH1 = x + L.xh1; K.XH0 = x + L.xh0; K.XH2 = x + L.xh2; K.XQ = x + L.xq; K.WU = x + L.wu;
    // the group's own word serves as the launch's abort word: a hand-off time-out writes its code there (and to the host flag, as everywhere),
    // and the decoder itself writes STOP_ALL_FINISHED there once every event of the group has emitted <eos> -- every wait loop then lets its
    // workgroup leave, which is exactly OldModel.sample's `break` (models/OldModel_NEW.py:179-180)
    u32* stopw = reinterpret_cast<u32*>(B.sws + LS.stop);
    K.abort_word = stopw; K.host_flag = h.flag_dev;
    K.spin_limit = config().persist_spin_limit > 0 ? (u32)config().persist_spin_limit : SPIN_LIMIT; K.inject = (u32)config().persist_inject_timeout;
    K.dh = off; K.dout = off;
    K.stamps = nullptr;
    if (config().persist_stamps == 1) {
        if (!h.stamps && hipMalloc(&h.stamps, 4 * 256 * 16 * 8) != hipSuccess) h.stamps = nullptr;
        if (h.stamps && a->S <= 256) { K.stamps = h.stamps; h.stamps_S = a->S; (void)hipMemsetAsync(h.stamps, 0, 4 * 256 * 16 * 8, st); }
    }
    PersistK2 K2;
    // the decoder keeps the plain placement: its 64 LSTM + logits workgroups stream 492 KB of logit weights each per step -- spread over eight
    // XCDs that is 3.9 MB per 4 MB L2, packed onto two it is 15.7 MB (measured: 256 events 2.93 vs 2.61 ms per decode; 64 and 1000 events equal)
    static const int xcd_map_s = [] { const char* e = getenv("ECHR_PERSIST_XCD_SAMPLE"); return e ? atoi(e) : 0; }();      // A/B switch
    K2.xcd_map = (xcd_map_s && 2 * HWG + 2 * NS == 256 && HWG == 96) ? 1 : 0;
    K2.N = K.N; K2.A = K.A; K2.D = K.D; K2.S = K.S; K2.ld_att = K.ld_att;
    K2.w_hh1 = a->w_hh[1]; K2.w_h2a = a->w_h2a; K2.b_h2a = a->b_h2a; K2.w_att = K.w_att; K2.w_alpha = a->w_alpha;
    K2.PALL = B.PALL; K2.c3d = a->c3d; K2.ev_start = a->ev_start; K2.ev_len = a->ev_len; K2.c3d_bytes = (unsigned)((size_t)a->Tv * a->D * 4);
    K2.GATES1 = nullptr; K2.CS1 = nullptr; K2.HS = nullptr; K2.OUTD = nullptr; K2.QS = nullptr; K2.WT = nullptr; K2.ATT = nullptr;
    K2.cnt = reinterpret_cast<u32*>(x2 + L2.cnt); K2.XC = x2 + L2.xc; K2.XS = x2 + L2.xs; K2.GRAN = reinterpret_cast<unsigned long long*>(x2 + L2.gran);
    K2.XH1 = x2 + L2.xh1; K2.XQ = x2 + L2.xq; K2.WU = x2 + L2.wu; K2.XCMAX = x2 + L2.xcmax;
    K2.pimg = nullptr;
    K2.nt_saved = 0;
    K2.abort_word = stopw; K2.host_flag = h.flag_dev; K2.stamps = K.stamps; K2.dh = off; K2.dout = off;
    K2.spin_limit = K.spin_limit; K2.inject = K.inject;
    PersistS Q;
    for (int k = 0; k < 3; ++k) Q.TG[k] = B.TG[k];
    Q.base0 = B.EVB0; Q.base2 = B.VIDB;
    Q.KEY = reinterpret_cast<unsigned long long*>(B.sws + LS.key);
    Q.cnt_tok = reinterpret_cast<u32*>(B.sws + LS.cnt);
    Q.LIMG = reinterpret_cast<const float4*>(B.limg);
    Q.nch = logit_chunks(a->V1);
    Q.force_eos = config().persist_sample_force_eos;
    Q.stop_early = stop_early ? 1 : 0;
    Q.linv = B.limg + (long)LWG * Q.nch * 3 * 16 * LCT * 2 * 64 * 4 + LWG * Q.nch * LCOLS;
    Q.lbias = a->b_logit;
    Q.LSE = B.sws + LS.lse; Q.XC3 = B.sws + LS.xc3; Q.XS3 = B.sws + LS.xs3;
    Q.cnt2 = K2.cnt; Q.XH1 = K2.XH1; Q.V1 = a->V1;
    {
        const double wbytes = 4.0 * (3.0 * 4 * PH * PH + (double)PH * PH + 4.0 * PH * a->D) + 4.0 * a->S * 3.0 * PH * a->V1;      // the logit weights are streamed once per step
        const double obytes = 4.0 * (double)a->N * a->A * (PH + a->D);
        ProfScope prof(PROF_PERSIST, 2.0 * a->S * PROWS * PH * (double)(3 * 4 * PH + PH + 4 * PH) + 2.0 * a->S * PROWS * 3.0 * PH * a->V1, wbytes + obytes, st);
        const bool big = a->A > PSET2;
        bool launched = false;
        if (config().persist_coop) {
            void* kargs[3] = {&K2, &K, &Q};
            const void* fn = big ? reinterpret_cast<const void*>(dec_persist_sample_kernel<true>) : reinterpret_cast<const void*>(dec_persist_sample_kernel<false>);
            if (hipLaunchCooperativeKernel(fn, dim3(2 * HWG + 2 * NS), dim3(256), kargs, LDS_BYTES_FWD, st) == hipSuccess) launched = true;
            else coop_refused("persist_sample", hipGetErrorString(hipGetLastError()));
        }
        if (!launched) {
            if (big) hipLaunchKernelGGL((dec_persist_sample_kernel<true>), dim3(2 * HWG + 2 * NS), dim3(256), LDS_BYTES_FWD, st, K2, K, Q);
            else hipLaunchKernelGGL((dec_persist_sample_kernel<false>), dim3(2 * HWG + 2 * NS), dim3(256), LDS_BYTES_FWD, st, K2, K, Q);
            if (int rc = check_launch("dec_persist_sample")) return rc;
        }
    }
    return 0;
}

bool persist_fwd_adds_evb0() {
    static const bool off = [] { const char* e = getenv("ECHR_EVB0_FOLD"); return e && e[0] == '0'; }();      // A/B switch (tools/ab_env.sh)
    return config().persist_h2 != 0 && !off;
}
bool persist_bwd_eligible(const echr_dec_args* a) { return config().persist_bwd && !det_mode() && persist_shape_ok(a); }

void persist_bwd_zero_range(const echr_dec_args* a, float* xws, float** ptr, long* count) {
    const PersistLayoutB L = persist_layout_b(a->S);
    const PersistLayoutB2 L2 = persist_layout_b2(a->S);
    if (config().persist_split) { *ptr = xws + L2.zero_begin; *count = L2.total - L2.zero_begin + L.xdq; }
    else { *ptr = xws; *count = L.zero_end; }
}

int persist_bwd(const echr_dec_args* a, const PersistBwdBufs& B, const DropCfg& dh, const DropCfg& dout, hipStream_t st) {
    PersistHost& h = phost();
    ECHR_REQUIRE(h.ok, "persist_bwd: device state unavailable");
    const PersistLayoutB L = persist_layout_b(a->S);
    PersistB K;
    K.N = a->N; K.A = a->A; K.D = a->D; K.S = a->S; K.ld_att = a->E + a->D;
    for (int k = 0; k < 3; ++k) { K.w_hh[k] = a->w_hh[k]; K.GATES[k] = B.GATES[k]; K.CS[k] = B.CS[k]; K.DG[k] = B.DG[k]; }
    K.w_h2a = a->w_h2a; K.w_att = a->w_ih[1] + a->E; K.w_alpha = a->w_alpha;
    K.PALL = B.PALL; K.c3d = a->c3d; K.ev_start = a->ev_start; K.ev_len = a->ev_len; K.c3d_bytes = (unsigned)((size_t)a->Tv * a->D * 4);
    K.QS = B.QS; K.WT = B.WT; K.ATT = B.ATT; K.DOUT = B.DOUT; K.DQ = B.DQ; K.DSC = B.DSC;
    const bool split = config().persist_split != 0;
    const PersistLayoutB2 L2 = persist_layout_b2(a->S);
    float* x2 = B.xws;                                  // version-2 layout first (its zeroed region last), version 1 behind it
    float* x = split ? x2 + L2.total : B.xws;
    K.cnt = reinterpret_cast<u32*>(x + L.cnt); K.XDQ = x + L.xdq; K.XDG = x + L.xdg; K.XDA = x + L.xda; K.XDH = x + L.xdh;
    K.XG0 = x + L.xg0; K.XG2 = x + L.xg2; K.XP0 = x + L.xp0; K.XP2 = x + L.xp2;
    K.lstm_kgroups = config().persist_kgroups;
    K.abort_word = h.abort_dev; K.host_flag = h.flag_dev;
    K.spin_limit = config().persist_spin_limit > 0 ? (u32)config().persist_spin_limit : SPIN_LIMIT; K.inject = (u32)config().persist_inject_timeout;
    K.dh = dh; K.dout = dout;
    K.stamps = nullptr;
    if (config().persist_stamps == 2) {
        if (!h.stamps && hipMalloc(&h.stamps, 4 * 256 * 16 * 8) != hipSuccess) h.stamps = nullptr;
        if (h.stamps && a->S <= 256) { K.stamps = h.stamps; h.stamps_S = a->S; (void)hipMemsetAsync(h.stamps, 0, 4 * 256 * 16 * 8, st); }
    }
    static const int xcd_map = [] { const char* e = getenv("ECHR_PERSIST_XCD"); return e ? atoi(e) : 1; }();      // A/B switch (default on, round 6)
    static const int xcd_bwd = [] { const char* e = getenv("ECHR_PERSIST_XCD_BWD"); return e ? atoi(e) : 2; }();      // A/B switch: 2 (default) = product roles on two XCDs, 1 = roles in order
    K.xcd_map = (xcd_map && 2 * HWG + 2 * NS == 256 && HWG == 96) ? (xcd_bwd == 2 ? 2 : 1) : 0;
    PersistB K2 = K;
    if (split) {
        K2.cnt = reinterpret_cast<u32*>(x2 + L2.cnt); K2.XDQ = x2 + L2.xdq; K2.XDA = x2 + L2.xda; K2.XDH = x2 + L2.xdh; K2.XDG = x2 + L2.xdg;
        // one memset: version 2's zeroed region and, right behind it, version 1's counters (all the LSTM kernel needs of that layout)
        if (!B.prezeroed && hipMemsetAsync(x2 + L2.zero_begin, 0, (size_t)(L2.total - L2.zero_begin + L.xdq) * sizeof(float), st) != hipSuccess) {
            set_error("persist_bwd: memset failed");
            return -5;
        }
    } else if (!B.prezeroed && hipMemsetAsync(x, 0, (size_t)L.zero_end * sizeof(float), st) != hipSuccess) { set_error("persist_bwd: memset failed"); return -5; }
    // algorithmic bytes of the reverse pair: weights and attention operands once, per step the saved activations in (gates, cells, q,
    // weights, context, upstream gradient) and the gradients out (gate gradients of three streams, d q, d score)
    const double wbytes = 4.0 * (3.0 * 4 * PH * PH + (double)PH * PH + 4.0 * PH * a->D);
    const double obytes = 4.0 * (double)a->N * a->A * (PH + a->D);
    const double sbytes = 4.0 * (double)a->S * a->N * (3.0 * 4 * PH + 3.0 * 2 * PH + PH + a->A + a->D + PH + 3.0 * 4 * PH + PH + a->A);
    ProfScope prof(PROF_PERSIST, 2.0 * a->S * PROWS * PH * (double)(3 * 4 * PH + PH + 4 * PH), wbytes + obytes + sbytes, st);
    if (split && config().persist_merge) {
        if (config().persist_coop) {
            void* kargs[2] = {&K2, &K};
            const void* fn = a->A > PSET2 ? reinterpret_cast<const void*>(dec_persist_bwd_kernel<true>) : reinterpret_cast<const void*>(dec_persist_bwd_kernel<false>);
            if (hipLaunchCooperativeKernel(fn, dim3(2 * HWG + 2 * NS), dim3(256), kargs, LDS_BYTES_BWD, st) == hipSuccess) return 0;
            coop_refused("persist_bwd", hipGetErrorString(hipGetLastError()));
        }
        if (a->A > PSET2) hipLaunchKernelGGL(dec_persist_bwd_kernel<true>, dim3(2 * HWG + 2 * NS), dim3(256), LDS_BYTES_BWD, st, K2, K);
        else hipLaunchKernelGGL(dec_persist_bwd_kernel<false>, dim3(2 * HWG + 2 * NS), dim3(256), LDS_BYTES_BWD, st, K2, K);
        return check_launch("dec_persist_bwd");
    }
    ECHR_REQUIRE(side_stream(h), "persist_bwd: second stream unavailable");
    if (hipEventRecord(h.fork, st) != hipSuccess || hipStreamWaitEvent(h.side, h.fork, 0) != hipSuccess) { set_error("persist_bwd: fork failed"); return -5; }
    hipLaunchKernelGGL(dec_persist_lstm_bwd_kernel, dim3(2 * NS), dim3(256), LDS_BYTES_LSTM, h.side, K);
    if (int rc = check_launch("dec_persist_lstm_bwd")) return rc;
    if (split) hipLaunchKernelGGL(dec_persist_att_bwd2_kernel, dim3(2 * HWG), dim3(256), LDS_BYTES_ATTB2, st, K2);
    else hipLaunchKernelGGL(dec_persist_att_bwd_kernel, dim3(NATT), dim3(256), LDS_BYTES_ATT, st, K);
    if (int rc = check_launch("dec_persist_att_bwd")) return rc;
    if (hipEventRecord(h.join, h.side) != hipSuccess || hipStreamWaitEvent(st, h.join, 0) != hipSuccess) { set_error("persist_bwd: join failed"); return -5; }
    return 0;
}

}  // namespace echr

// diagnostic / test hook (tests/test_host_contract.py): the role a workgroup of the merged 256-workgroup recurrence launches plays under placement
// `mode` (0 = blockIdx order, 1 = a half machine per three XCDs, 2 = the reverse pair's form with the product roles on two XCDs).  A placement
// that is not a bijection of 0..255 would leave roles unfilled and every wait for them timing out.
extern "C" int32_t echr_persist_role_index(int32_t block, int32_t mode) { return echr::persist_role_index(block, mode); }

