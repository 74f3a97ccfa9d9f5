// Price of the three all-to-all hand-offs per decoder timestep inside ONE persistent launch (256 workgroups, one per CU),
// with the payload sizes of the c3 workload (N=64 events, H=Ha=512, D=500):
//   hop 1: 128 gate workgroups publish h1[:, 4 units] (1 KB each)  -> 32 q workgroups ingest all of h1 (128 KB each)
//   hop 2: 32 q workgroups publish q[:, 16 cols] (4 KB each)       -> all 256 workgroups ingest q[n,:] (2 KB each)
//   hop 3: 256 workgroups add a context partial (2 KB, fp32 atomics) -> 128 gate workgroups ingest ctx (128 KB each)
// Protocol (MI355X_MICROARCH.md, visibility table row 1): sc1 stores -> every storing wave s_waitcnt vmcnt(0) -> workgroup barrier
// -> ONE lane's agent-scope atomic add on a per-(hop,step) counter (sharded 8 ways); the consumer polls the shards with sc1 loads,
// joins a workgroup barrier, then reads the payload with 16-byte sc1 buffer loads (no fences).  Counters are per step (never reused
// inside a launch) and zeroed by a memset before every launch.  Every spin is bounded; a timeout raises a global abort word.
// Payloads carry (step, producer) signatures so that every consumer checks EVERY word it ingests (stale data = counted error).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef unsigned u32;
typedef u32 u32x4 __attribute__((ext_vector_type(4)));
#define AGENT __HIP_MEMORY_SCOPE_AGENT

constexpr int NWG = 256, NG = 128, NQ = 32, NEV = 64, H = 512;
constexpr int SHARDS = 8, SHSTRIDE = 32;      // one 128-byte line per shard
constexpr u32 SPIN_LIMIT = 2000000;

struct Bufs {
    float* h;        // [steps+1][NG][NEV][4]
    float* q;        // [steps][NQ][NEV][16]
    float* ctx;      // [steps][NEV][H]   (atomic accumulation, zeroed before launch)
    u32* cnt;        // [3][steps][SHARDS][SHSTRIDE]
    u32* abort_word; // [0] = timeout code, [1] = data errors
    unsigned long long* stamps;   // [steps] s_memrealtime of workgroup 0 at the end of each step
};

__device__ __forceinline__ void st_sc1(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, AGENT); }

__device__ __forceinline__ float4 ld16_sc1(const float* base, u32 byte_off, u32 bytes) {
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, bytes, 0x00020000);
    u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 16);     // aux 16 = sc1
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}

// all threads of the workgroup call this after their sc1 stores / atomics
__device__ __forceinline__ void publish(u32* cnt_line) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(cnt_line + (blockIdx.x % SHARDS) * SHSTRIDE, 1u, __ATOMIC_RELAXED, AGENT);
}

// all threads call; returns false when the launch is being aborted
__device__ __forceinline__ bool wait_total(u32* cnt_line, u32 target, u32* abort_word, int* lds_flag, u32 code, int sleepy = 1) {
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        u32 spins = 0;
        bool ok = false;
        for (;;) {
            u32 v = lane < SHARDS ? __hip_atomic_load(cnt_line + lane * SHSTRIDE, __ATOMIC_RELAXED, AGENT) : 0u;
            for (int o = 4; o > 0; o >>= 1) v += __shfl_xor(v, o);
            v = __shfl(v, 0);
            if (v >= target) { ok = true; break; }
            if ((++spins & 63) == 0) {
                const u32 ab = __hip_atomic_load(abort_word, __ATOMIC_RELAXED, AGENT);
                if (ab) break;
                if (spins > SPIN_LIMIT) { if (lane == 0) __hip_atomic_store(abort_word, code, __ATOMIC_RELAXED, AGENT); break; }
            }
            if (sleepy) __builtin_amdgcn_s_sleep(2);
        }
        if (lane == 0) *lds_flag = ok ? 1 : 0;
    }
    __syncthreads();
    const bool r = *lds_flag != 0;
    __syncthreads();
    return r;
}

__device__ __forceinline__ float sig(int step, int producer, int k) { return (float)(step * 1000 + producer) + 0.001f * (float)(k & 7); }

template <bool COMPUTE>
__global__ __launch_bounds__(256) void hop_kernel(Bufs B, int steps, int ING, int SLEEPY) {
    __shared__ int flag;
    __shared__ float red[256];
    const int b = blockIdx.x, tid = threadIdx.x;
    const bool gate = b < NG, qrole = b >= NG && b < NG + NQ;
    const int n_ev = b / 4;
    u32 errors = 0;
    auto cnt = [&](int hop, int t) { return B.cnt + ((long)(hop * steps + t) * SHARDS) * SHSTRIDE; };
    // h(-1): every gate workgroup publishes its slice as "step 0" payload
    if (gate) {
        float* hp = B.h + ((long)0 * NG + b) * NEV * 4;
        st_sc1(hp + tid, sig(0, b, tid));
    }
    if (gate) publish(cnt(0, 0));
    for (int t = 0; t < steps; ++t) {
        float carry = 0.f;
        // ---- hop 1 consumer: q workgroups ingest all of h(t-1) = slot t, 128 KB ----
        if (qrole) {
            if (!wait_total(cnt(0, t), NG, B.abort_word, &flag, 100 + t, SLEEPY)) return;
            const float* hp = B.h + (long)t * NG * NEV * 4;
            float acc = 0.f;
#pragma unroll 8
            for (int i = 0; i < ING; ++i) {          // 32 x 16 B per thread = 128 KB per workgroup
                const u32 f4 = i * 256 + tid;     // float4 index; producer = f4 / 64
                const float4 v = ld16_sc1(hp, f4 * 16, NG * NEV * 16);
                const int prod = f4 / 64, k = (f4 % 64) * 4;
                if (v.x != sig(t, prod, k) || v.y != sig(t, prod, k + 1) || v.z != sig(t, prod, k + 2) || v.w != sig(t, prod, k + 3)) ++errors;
                acc += v.x + v.y + v.z + v.w;
            }
            carry = acc * 1e-30f;
            if (COMPUTE) {          // stand-in for the 1.7 us of MFMA work
                float x = carry;
                for (int i = 0; i < 1000; ++i) x = __builtin_fmaf(x, 1.0001f, 1e-9f);
                carry = x * 1e-30f;
            }
            float* qp = B.q + ((long)t * NQ + (b - NG)) * NEV * 16;
#pragma unroll
            for (int i = 0; i < 4; ++i) st_sc1(qp + i * 256 + tid, sig(t, b, i * 256 + tid) + carry);
            publish(cnt(1, t));
        }
        // ---- hop 2 consumer: everybody ingests q[n,:] (32 pieces of 64 B) ----
        if (!wait_total(cnt(1, t), NQ, B.abort_word, &flag, 200 + t, SLEEPY)) return;
        {
            const float* qb = B.q + (long)t * NQ * NEV * 16;
            float acc = 0.f;
            if (tid < 128) {      // 128 float4 = 2 KB
                const int c = tid / 4, j4 = tid % 4;
                const u32 f = (c * NEV + n_ev) * 16 + j4 * 4;
                const float4 v = ld16_sc1(qb, f * 4, NQ * NEV * 64);
                const int k = n_ev * 16 + j4 * 4;
                if (v.x != sig(t, NG + c, k) || v.y != sig(t, NG + c, k + 1) || v.z != sig(t, NG + c, k + 2) || v.w != sig(t, NG + c, k + 3)) ++errors;
                acc = v.x + v.y + v.z + v.w;
            }
            if (COMPUTE) {          // stand-in for the score / context arithmetic (about 1 us)
                float x = acc * 1e-30f;
                for (int i = 0; i < 600; ++i) x = __builtin_fmaf(x, 1.0001f, 1e-9f);
                acc += x * 1e-30f;
            }
            // context partial: 2 KB of fp32 atomics into ctx[t][n][:]
            float* cp = B.ctx + ((long)t * NEV + n_ev) * H;
            atomicAdd(cp + tid, 1.0f + acc * 1e-30f);
            atomicAdd(cp + 256 + tid, 1.0f + acc * 1e-30f);
            publish(cnt(2, t));
        }
        // ---- hop 3 consumer: gate workgroups ingest ctx (128 KB), then publish h(t) ----
        if (gate) {
            if (!wait_total(cnt(2, t), NWG, B.abort_word, &flag, 300 + t, SLEEPY)) return;
            const float* cp = B.ctx + (long)t * NEV * H;
            float acc = 0.f;
#pragma unroll 8
            for (int i = 0; i < ING; ++i) {
                const u32 f4 = i * 256 + tid;
                const float4 v = ld16_sc1(cp, f4 * 16, NEV * H * 4);
                if (v.x != 4.0f || v.y != 4.0f || v.z != 4.0f || v.w != 4.0f) ++errors;      // 4 partials of 1.0 each
                acc += v.x;
            }
            if (COMPUTE) {
                float x = acc * 1e-30f;
                for (int i = 0; i < 1000; ++i) x = __builtin_fmaf(x, 1.0001f, 1e-9f);
                acc = x;
            }
            float* hp = B.h + ((long)(t + 1) * NG + b) * NEV * 4;
            st_sc1(hp + tid, sig(t + 1, b, tid) + acc * 1e-30f);
            publish(cnt(0, t + 1));
        }
        if (b == 0 && tid == 0) B.stamps[t] = __builtin_amdgcn_s_memrealtime();
    }
    red[tid] = (float)errors;
    __syncthreads();
    if (tid == 0) {
        u32 e = 0;
        for (int i = 0; i < 256; ++i) e += (u32)red[i];
        if (e) atomicAdd(B.abort_word + 1, e);
    }
}

int main(int argc, char** argv) {
    const int steps = argc > 1 ? atoi(argv[1]) : 20;
    const int ING = argc > 2 ? atoi(argv[2]) : 32, SLEEPY = argc > 3 ? atoi(argv[3]) : 1;
    Bufs B;
    const size_t hb = (size_t)(steps + 1) * NG * NEV * 4 * 4, qb = (size_t)steps * NQ * NEV * 16 * 4, cb = (size_t)steps * NEV * H * 4;
    const size_t nb = (size_t)3 * (steps + 1) * SHARDS * SHSTRIDE * 4;
    CK(hipMalloc(&B.h, hb)); CK(hipMalloc(&B.q, qb)); CK(hipMalloc(&B.ctx, cb)); CK(hipMalloc(&B.cnt, nb));
    CK(hipMalloc(&B.abort_word, 64)); CK(hipMalloc(&B.stamps, steps * 8));
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int variant = 0; variant < 2; ++variant) {
        double best = 1e30; unsigned ab[2] = {0, 0};
        unsigned long long st[64];
        for (int rep = 0; rep < 5; ++rep) {
            CK(hipMemsetAsync(B.cnt, 0, nb, s)); CK(hipMemsetAsync(B.abort_word, 0, 64, s)); CK(hipMemsetAsync(B.ctx, 0, cb, s));
            CK(hipMemsetAsync(B.h, 0xff, hb, s)); CK(hipMemsetAsync(B.q, 0xff, qb, s));
            CK(hipEventRecord(e0, s));
            if (variant == 0) hipLaunchKernelGGL(hop_kernel<false>, dim3(NWG), dim3(256), 0, s, B, steps, ING, SLEEPY);
            else hipLaunchKernelGGL(hop_kernel<true>, dim3(NWG), dim3(256), 0, s, B, steps, ING, SLEEPY);
            CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
            float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
            CK(hipMemcpy(ab, B.abort_word, 8, hipMemcpyDeviceToHost));
            if (ms * 1e3 < best) { best = ms * 1e3; CK(hipMemcpy(st, B.stamps, (steps < 64 ? steps : 64) * 8, hipMemcpyDeviceToHost)); }
            if (ab[0] || ab[1]) break;
        }
        printf("ING=%d sleep=%d variant %d (%s): %d steps, %.1f us per launch = %.2f us per step (3 hops)   abort=%u data_errors=%u\n", ING, SLEEPY, variant,
               variant ? "with stand-in compute" : "hand-offs only", steps, best, best / steps, ab[0], ab[1]);
        if (steps >= 4) printf("   in-kernel step period (s_memrealtime, 100 MHz): %.2f us\n", (double)(st[steps - 1] - st[1]) / (steps - 2) / 100.0);
    }
    return 0;
}
