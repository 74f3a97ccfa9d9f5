// Error plumbing + the small bandwidth-bound kernels of the ECHR hot path (reductions, embedding
// gather/scatter, row log-softmax and its backward, masked NLL, event pooling, greedy arg-max,
// fused clamp+Adam).  All wave64; row kernels use one 256-thread workgroup per row with
// cross-lane shuffle reductions.
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "echr_common.h"
#include "echr_internal.h"

namespace echr {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: launch failed: %s", what, hipGetErrorString(e));
        return -5;  // -EIO
    }
    return 0;
}

// ---- runtime configuration -------------------------------------------------------------------------------
static int env_int(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }
Config& config() {
    static Config c = {env_int("ECHR_GEMM_BF16X3", 1), env_int("ECHR_OVERLAP", 0), env_int("ECHR_ATT_SLOTS", 2), env_int("ECHR_CHAINS2", 0),
                       env_int("ECHR_GEMM_H2", 1), env_int("ECHR_PERSIST", 1), env_int("ECHR_PERSIST_STAMPS", 0),
                       getenv("ECHR_GEMM_TILE") ? (int)getenv("ECHR_GEMM_TILE")[0] : 0, env_int("ECHR_GEMM_SPLIT", 0), env_int("ECHR_PERSIST_BWD", 1), env_int("ECHR_PERSIST_SPLIT", 1), env_int("ECHR_PERSIST_H2", 1), env_int("ECHR_PERSIST_MERGE", 1), env_int("ECHR_PERSIST_KGROUPS", 1), env_int("ECHR_TSRM_FORK", 1),
                       env_int("ECHR_PERSIST_COOP", 0), 0, env_int("ECHR_PERSIST_SPIN_LIMIT", 0), env_int("ECHR_SST_PERSIST", 1), env_int("ECHR_TAIL_EARLY", 0), 0, env_int("ECHR_EMBED_FUSED", 0), env_int("ECHR_PERSIST_SAMPLE", 1), env_int("ECHR_POSEMB_ROWS", 1), env_int("ECHR_GEMM_SKINNY", 1), env_int("ECHR_POSEMB_PACKED", 1), env_int("ECHR_PAIR_TABLES", 1), 0, env_int("ECHR_PERSIST_SAMPLE_MAX", 512), env_int("ECHR_DETERMINISTIC", 0)};
    return c;
}

// 32x32 LDS-tiled transpose with zero padding of the output rows' tail
__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ in, long ld_in, float* __restrict__ out, long ld_out,
                                                        int rows, int cols, int rows_pad) {
    __shared__ float t[32][33];
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int i = ty; i < 32; i += 8)
        t[i][tx] = (r0 + i < rows && c0 + tx < cols) ? in[(long)(r0 + i) * ld_in + c0 + tx] : 0.f;
    __syncthreads();
    for (int i = ty; i < 32; i += 8)
        if (c0 + i < cols && r0 + tx < rows_pad) out[(long)(c0 + i) * ld_out + r0 + tx] = t[tx][i];
}
struct TransposeArgs { TransposeJob job[TRANSPOSE_MAX_JOBS]; int start[TRANSPOSE_MAX_JOBS + 1]; int njobs; };
__global__ __launch_bounds__(256) void transpose_multi_kernel(TransposeArgs a) {
    __shared__ float t[32][33];
    int ji = 0;
    while (ji + 1 < a.njobs && (int)blockIdx.x >= a.start[ji + 1]) ++ji;
    const TransposeJob J = a.job[ji];
    const int ci = (int)blockIdx.x - a.start[ji], ct = (J.cols + 31) / 32;
    const int r0 = (ci / ct) * 32, c0 = (ci % ct) * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int i = ty; i < 32; i += 8)
        t[i][tx] = (r0 + i < J.rows && c0 + tx < J.cols) ? J.in[(long)(r0 + i) * J.ld_in + c0 + tx] : 0.f;
    __syncthreads();
    for (int i = ty; i < 32; i += 8)
        if (c0 + i < J.cols && r0 + tx < J.rows) J.out[(long)(c0 + i) * J.ld_out + r0 + tx] = t[tx][i];
}
// several out[c, r] = in[r, c] problems in one launch
int transpose_multi(const TransposeJob* jobs, int n, hipStream_t st) {
    ECHR_REQUIRE(jobs && n >= 1 && n <= TRANSPOSE_MAX_JOBS, "transpose_multi: 1..%d jobs", TRANSPOSE_MAX_JOBS);
    TransposeArgs a;
    int total = 0;
    for (int i = 0; i < n; ++i) {
        a.job[i] = jobs[i];
        a.start[i] = total;
        total += ((jobs[i].cols + 31) / 32) * ((jobs[i].rows + 31) / 32);
    }
    for (int i = n; i < TRANSPOSE_MAX_JOBS; ++i) a.job[i] = jobs[0];
    for (int i = n; i <= TRANSPOSE_MAX_JOBS; ++i) a.start[i] = total;
    a.njobs = n;
    hipLaunchKernelGGL(transpose_multi_kernel, dim3(total), dim3(256), 0, st, a);
    return check_launch("transpose_multi");
}
int transpose(const float* in, long ld_in, float* out, long ld_out, int rows, int cols, int rows_pad, hipStream_t st) {
    hipLaunchKernelGGL(transpose_kernel, dim3((cols + 31) / 32, (rows_pad + 31) / 32), dim3(256), 0, st, in, ld_in, out, ld_out, rows, cols,
                       rows_pad);
    return check_launch("transpose");
}

// ---- optional HIP-event profiling --------------------------------------------------------------------
struct ProfRec { hipEvent_t e0, e1; int kind; double flops, bytes; };
static bool g_prof_on = false;
static std::vector<ProfRec> g_prof_recs;
static std::vector<hipEvent_t> g_prof_pool;
static double g_prof_ms[PROF_KINDS], g_prof_flops[PROF_KINDS], g_prof_bytes[PROF_KINDS];
static long long g_prof_n[PROF_KINDS];

static hipEvent_t prof_event() {
    if (!g_prof_pool.empty()) { hipEvent_t e = g_prof_pool.back(); g_prof_pool.pop_back(); return e; }
    hipEvent_t e;
    (void)hipEventCreate(&e);
    return e;
}
ProfScope::ProfScope(int k, double f, double b, hipStream_t s) : st(s), kind(k), flops(f), bytes(b) {
    if (!g_prof_on) return;
    e0 = prof_event(); e1 = prof_event();
    (void)hipEventRecord(e0, st);
}
ProfScope::~ProfScope() {
    if (!e0) return;
    (void)hipEventRecord(e1, st);
    g_prof_recs.push_back({e0, e1, kind, flops, bytes});
}
static void prof_resolve() {
    for (auto& r : g_prof_recs) {
        float ms = 0.f;
        (void)hipEventSynchronize(r.e1);
        (void)hipEventElapsedTime(&ms, r.e0, r.e1);
        g_prof_ms[r.kind] += ms; g_prof_flops[r.kind] += r.flops; g_prof_bytes[r.kind] += r.bytes; g_prof_n[r.kind] += 1;
        g_prof_pool.push_back(r.e0); g_prof_pool.push_back(r.e1);
    }
    g_prof_recs.clear();
}

// ---- block reductions ----------------------------------------------------------------------------
__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    float r = 0.f;
    for (int w = 0; w < nw; ++w) r += red[w];
    return r;
}
__device__ __forceinline__ float block_max(float v, float* red) {
    v = wave_max(v);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    float r = red[0];
    for (int w = 1; w < nw; ++w) r = fmaxf(r, red[w]);
    return r;
}

// ---- fill / colsum / sum over time ----------------------------------------------------------------
int fill_zero(float* p, long n, hipStream_t st);
__global__ void fill_zero_kernel(float* p, long n) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long stride = (long)gridDim.x * blockDim.x;
    for (; i < n; i += stride) p[i] = 0.f;
}
__global__ void fill_zero_2d_kernel(float* p, int rows, int cols, long ld) {
    const long tot = (long)rows * cols;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < tot; i += stride) p[(i / cols) * ld + (i % cols)] = 0.f;
}
int fill_zero_2d(float* p, int rows, int cols, long ld, hipStream_t st) {
    if (ld == cols) return fill_zero(p, (long)rows * cols, st);
    const long tot = (long)rows * cols;
    int grid = (int)min((tot + 255) / 256, 2048L);
    hipLaunchKernelGGL(fill_zero_2d_kernel, dim3(grid), dim3(256), 0, st, p, rows, cols, ld);
    return check_launch("fill_zero_2d");
}
int fill_zero(float* p, long n, hipStream_t st) {
    if (n <= 0) return 0;
    int grid = (int)min((n + 255) / 256, 2048L);
    hipLaunchKernelGGL(fill_zero_kernel, dim3(grid), dim3(256), 0, st, p, n);
    return check_launch("fill_zero");
}

// several ranges zeroed by one launch (each workgroup owns 4096 floats of one range)
struct FillArgs { float* p[FILL_MAX_JOBS]; long n[FILL_MAX_JOBS]; int start[FILL_MAX_JOBS + 1]; int njobs; };
__global__ __launch_bounds__(256) void fill_zero_multi_kernel(FillArgs a) {
    int ji = 0;
    while (ji + 1 < a.njobs && (int)blockIdx.x >= a.start[ji + 1]) ++ji;
    float* p = a.p[ji];
    const long n = a.n[ji];
    const long base = (long)((int)blockIdx.x - a.start[ji]) * 4096;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const long k = base + i * 256 + threadIdx.x;
        if (k < n) p[k] = 0.f;
    }
}
int fill_zero_multi(float* const* ptrs, const long* counts, int n, hipStream_t st) {
    ECHR_REQUIRE(ptrs && counts && n >= 1 && n <= FILL_MAX_JOBS, "fill_zero_multi: 1..%d ranges", FILL_MAX_JOBS);
    FillArgs a;
    int total = 0;
    for (int i = 0; i < n; ++i) {
        a.p[i] = ptrs[i]; a.n[i] = counts[i]; a.start[i] = total;
        total += (int)((counts[i] + 4095) / 4096);
    }
    for (int i = n; i < FILL_MAX_JOBS; ++i) { a.p[i] = ptrs[0]; a.n[i] = 0; }
    for (int i = n; i <= FILL_MAX_JOBS; ++i) a.start[i] = total;
    a.njobs = n;
    if (total == 0) return 0;
    hipLaunchKernelGGL(fill_zero_multi_kernel, dim3(total), dim3(256), 0, st, a);
    return check_launch("fill_zero_multi");
}

// one lane per column (coalesced 256 B per wave-row), 4 waves stride the rows of one 256-row chunk, LDS combine;
// several row chunks (grid.y) add their partial sums atomically into the zero-initialised output.
constexpr int CS_ROWS = 128;
// column sum of rows [r0, r1) with stride 4 starting at r0 + wave: four independent loads in flight per lane
__device__ __forceinline__ float colsum_rows(const float* __restrict__ X, long ld, int col, int r0, int r1, int wave) {
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int r = r0 + wave;
    for (; r + 12 < r1; r += 16) {
        s0 += X[(long)r * ld + col]; s1 += X[(long)(r + 4) * ld + col]; s2 += X[(long)(r + 8) * ld + col]; s3 += X[(long)(r + 12) * ld + col];
    }
    for (; r < r1; r += 4) s0 += X[(long)r * ld + col];
    return (s0 + s1) + (s2 + s3);
}
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ X, long ld, int rows, int cols,
                                                     float* __restrict__ out, float* __restrict__ out2, int mode) {
    __shared__ float red[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = blockIdx.x * 64 + lane;
    const int r0 = blockIdx.y * CS_ROWS, r1 = min(rows, r0 + CS_ROWS);
    const float s = col < cols ? colsum_rows(X, ld, col, r0, r1, wave) : 0.f;
    red[wave][lane] = s;
    __syncthreads();
    if (wave == 0 && col < cols) {
        const float t = red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane];
        if (mode == 0) { out[col] = t; if (out2) out2[col] = t; }            // single chunk, overwrite
        else { atomicAdd(&out[col], t); if (out2) atomicAdd(&out2[col], t); }   // accumulate / multi-chunk
    }
}
// scratch slabs of the fixed-order ("deterministic") variants
float* det_scratch(int kind, size_t floats) {
    static float* buf[DET_KINDS] = {};
    static size_t cap[DET_KINDS] = {};
    if (kind < 0 || kind >= DET_KINDS) return nullptr;
    if (floats <= cap[kind]) return buf[kind];
    if (buf[kind]) { (void)hipDeviceSynchronize(); (void)hipFree(buf[kind]); buf[kind] = nullptr; cap[kind] = 0; }
    const size_t want = floats + floats / 4 + 1024;
    if (hipMalloc(&buf[kind], want * sizeof(float)) != hipSuccess) { (void)hipGetLastError(); set_error("deterministic: no memory for %zu scratch floats", want); return nullptr; }
    cap[kind] = want;
    return buf[kind];
}

// fixed-order column sum ("deterministic" = 1): ONE workgroup owns 64 columns over all rows -- 16 waves stride the rows with four loads in flight
// each, combined through LDS in wave order; the output is overwritten or updated by its single writer (no atomics, no chunk order)
__global__ __launch_bounds__(1024) void colsum_det_kernel(const float* __restrict__ X, long ld, int rows, int cols, float* __restrict__ out,
                                                          float* __restrict__ out2, float* __restrict__ out3, int accumulate) {
    __shared__ float red[16][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = blockIdx.x * 64 + lane;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (col < cols) {
        int r = wave;
        for (; r + 48 < rows; r += 64) {
            s0 += X[(long)r * ld + col]; s1 += X[(long)(r + 16) * ld + col]; s2 += X[(long)(r + 32) * ld + col]; s3 += X[(long)(r + 48) * ld + col];
        }
        for (; r < rows; r += 16) s0 += X[(long)r * ld + col];
    }
    red[wave][lane] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (wave == 0 && col < cols) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < 16; ++w) t += red[w][lane];
        if (accumulate) { out[col] += t; if (out2) out2[col] += t; if (out3) out3[col] += t; }
        else { out[col] = t; if (out2) out2[col] = t; if (out3) out3[col] = t; }
    }
}
static int colsum_det(const float* X, long ld, int rows, int cols, float* out, float* out2, float* out3, bool accumulate, hipStream_t st) {
    hipLaunchKernelGGL(colsum_det_kernel, dim3((cols + 63) / 64), dim3(1024), 0, st, X, ld, rows, cols, out, out2, out3, accumulate ? 1 : 0);
    return check_launch("colsum_det");
}

int colsum2(const float* X, long ld, int rows, int cols, float* out, float* out2, bool accumulate, hipStream_t st) {
    if (det_mode()) return colsum_det(X, ld, rows, cols, out, out2, nullptr, accumulate, st);
    const int chunks = (rows + CS_ROWS - 1) / CS_ROWS;
    int mode = (accumulate || chunks > 1) ? 1 : 0;
    if (!accumulate && chunks > 1) {
        int rc = fill_zero(out, cols, st); if (rc) return rc;
        if (out2) { rc = fill_zero(out2, cols, st); if (rc) return rc; }
    }
    hipLaunchKernelGGL(colsum_kernel, dim3((cols + 63) / 64, chunks), dim3(256), 0, st, X, ld, rows, cols, out, out2, mode);
    return check_launch("colsum");
}
int colsum(const float* X, long ld, int rows, int cols, float* out, bool accumulate, hipStream_t st) {
    return colsum2(X, ld, rows, cols, out, nullptr, accumulate, st);
}

// several column-sum problems in ONE launch; every output accumulates atomically (the caller owns the base values: gradient
// buffers that are already zeroed or being accumulated into).  Linear grid over (job, column tile, row chunk).
struct ColsumArgs { ColsumJob job[COLSUM_MAX_JOBS]; int start[COLSUM_MAX_JOBS + 1]; int njobs; };
__global__ __launch_bounds__(256) void colsum_multi_kernel(ColsumArgs a) {
    __shared__ float red[4][64];
    int ji = 0;
    while (ji + 1 < a.njobs && (int)blockIdx.x >= a.start[ji + 1]) ++ji;
    const ColsumJob J = a.job[ji];
    const int ci = (int)blockIdx.x - a.start[ji];
    const int ctiles = (J.cols + 63) / 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = (ci % ctiles) * 64 + lane;
    const int r0 = (ci / ctiles) * CS_ROWS, r1 = min(J.rows, r0 + CS_ROWS);
    const float s = col < J.cols ? colsum_rows(J.X, J.ld, col, r0, r1, wave) : 0.f;
    red[wave][lane] = s;
    __syncthreads();
    if (wave == 0 && col < J.cols) {
        const float t = red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane];
        atomicAdd(&J.out[col], t);
        if (J.out2) atomicAdd(&J.out2[col], t);
        if (J.out3) atomicAdd(&J.out3[col], t);
    }
}
int colsum_multi(const ColsumJob* jobs, int n, hipStream_t st) {
    ECHR_REQUIRE(jobs && n >= 1 && n <= COLSUM_MAX_JOBS, "colsum_multi: 1..%d jobs", COLSUM_MAX_JOBS);
    if (det_mode()) {          // one fixed-order launch per job; the outputs accumulate (single writer per column)
        for (int i = 0; i < n; ++i) {
            ECHR_REQUIRE(jobs[i].X && jobs[i].out && jobs[i].rows > 0 && jobs[i].cols > 0, "colsum_multi: bad job %d", i);
            if (int rc = colsum_det(jobs[i].X, jobs[i].ld, jobs[i].rows, jobs[i].cols, jobs[i].out, jobs[i].out2, jobs[i].out3, true, st)) return rc;
        }
        return 0;
    }
    ColsumArgs a;
    int total = 0;
    for (int i = 0; i < n; ++i) {
        ECHR_REQUIRE(jobs[i].X && jobs[i].out && jobs[i].rows > 0 && jobs[i].cols > 0, "colsum_multi: bad job %d", i);
        a.job[i] = jobs[i];
        a.start[i] = total;
        total += ((jobs[i].cols + 63) / 64) * ((jobs[i].rows + CS_ROWS - 1) / CS_ROWS);
    }
    for (int i = n; i < COLSUM_MAX_JOBS; ++i) a.job[i] = jobs[0];
    for (int i = n; i <= COLSUM_MAX_JOBS; ++i) a.start[i] = total;
    a.njobs = n;
    hipLaunchKernelGGL(colsum_multi_kernel, dim3(total), dim3(256), 0, st, a);
    return check_launch("colsum_multi");
}

__global__ void sum_over_time_kernel(const float* __restrict__ X, long ld, int S, int N, int cols,
                                     float* __restrict__ out, long ld_out) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)N * cols) return;
    const int n = (int)(idx / cols), j = (int)(idx % cols);
    // eight rows' loads in flight per round (one load per iteration is a chain of S dependent memory latencies); same order of additions
    float s = 0.f;
    int t = 0;
    for (; t + 8 <= S; t += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = X[((long)(t + u) * N + n) * ld + j];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; t < S; ++t) s += X[((long)t * N + n) * ld + j];
    out[(long)n * ld_out + j] = s;
}
int sum_over_time(const float* X, long ld, int S, int N, int cols, float* out, long ld_out, hipStream_t st) {
    const long tot = (long)N * cols;
    hipLaunchKernelGGL(sum_over_time_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, X, ld, S, N, cols, out, ld_out);
    return check_launch("sum_over_time");
}

// ---- rank-1 products of the scene-context path (one video vector per batch: OldModel_NEW.py:808-818 feeds it to stream 2 every step) ----
// C[r, c] = x[r] * y[c] (+ C[r, c] when accumulate): the K = 1 "product" d W_ih2[:, E:] = colsum(d G_2)^T . video
__global__ __launch_bounds__(256) void rank1_kernel(const float* __restrict__ x, const float* __restrict__ y, float* __restrict__ C, long ldc,
                                                    int M, int N, int accumulate) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)M * N) return;
    const int r = (int)(idx / N), c = (int)(idx % N);
    const float v = x[r] * y[c];
    C[(long)r * ldc + c] = accumulate ? C[(long)r * ldc + c] + v : v;
}
int rank1_update(const float* x, const float* y, float* C, long ldc, int M, int N, bool accumulate, hipStream_t st) {
    hipLaunchKernelGGL(rank1_kernel, dim3((unsigned)(((long)M * N + 255) / 256)), dim3(256), 0, st, x, y, C, ldc, M, N, accumulate ? 1 : 0);
    return check_launch("rank1_update");
}
// out[c] = sum_r x[r] * W[r, c]  (M rows, N <= a few hundred columns): one workgroup per 64 columns, 4 row groups reduced through LDS
__global__ __launch_bounds__(256) void vec_mat_kernel(const float* __restrict__ x, const float* __restrict__ W, long ldw, float* __restrict__ out, int M, int N) {
    __shared__ float red[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), part = threadIdx.x >> 6;
    float s = 0.f;
    if (c < N)
        for (int r = part; r < M; r += 4) s = fmaf(x[r], W[(long)r * ldw + c], s);
    red[part][threadIdx.x & 63] = s;
    __syncthreads();
    if (part == 0 && c < N) out[c] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}
int vec_mat(const float* x, const float* W, long ldw, float* out, int M, int N, hipStream_t st) {
    hipLaunchKernelGGL(vec_mat_kernel, dim3((N + 63) / 64), dim3(256), 0, st, x, W, ldw, out, M, N);
    return check_launch("vec_mat");
}

// out[c] = sum_k x[k] * W[c, k] + b1[c] + b2[c]  (one input row against N weight rows of K <= a few hundred contiguous floats): one WAVE per
// output (the lanes stride k, one reduction) -- the scene-context gate bias W_ih2[:, E:] . video + b_ih2 + b_hh2 without a GEMM launch.  (A
// quarter wave per output walked 25 dependent loads: 14 us on the decoder's prepare chain.)
__global__ __launch_bounds__(256) void row_matvec_kernel(const float* __restrict__ x, const float* __restrict__ W, long ldw, const float* __restrict__ b1,
                                                         const float* __restrict__ b2, float* __restrict__ out, int N, int K) {
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    float s = 0.f;
    if (c < N)
        for (int k = lane; k < K; k += 64) s = fmaf(x[k], W[(long)c * ldw + k], s);
    s = wave_sum(s);
    if (c < N && lane == 0) out[c] = s + (b1 ? b1[c] : 0.f) + (b2 ? b2[c] : 0.f);
}
int row_matvec(const float* x, const float* W, long ldw, const float* b1, const float* b2, float* out, int N, int K, hipStream_t st) {
    hipLaunchKernelGGL(row_matvec_kernel, dim3((N + 3) / 4), dim3(256), 0, st, x, W, ldw, b1, b2, out, N, K);
    return check_launch("row_matvec");
}

// ---- embedding gather / scatter-add ----------------------------------------------------------------
__global__ void embed_gather_kernel(const float* __restrict__ W, const int* __restrict__ tok, float* __restrict__ out,
                                    int rows, int E, int V1) {
    const int row = blockIdx.x;
    int t = tok[row];
    t = min(max(t, 0), V1 - 1);
    for (int j = threadIdx.x; j < E; j += blockDim.x) out[(long)row * E + j] = W[(long)t * E + j];
}
int embed_gather(const float* W, const int* tok, float* out, int rows, int E, int V1, hipStream_t st) {
    hipLaunchKernelGGL(embed_gather_kernel, dim3(rows), dim3(128), 0, st, W, tok, out, rows, E, V1);
    return check_launch("embed_gather");
}
__global__ void embed_scatter_add_kernel(const float* __restrict__ dX, const int* __restrict__ tok, float* __restrict__ gW,
                                         int rows, int E, int V1, const int* __restrict__ rowmap) {
    const int row = blockIdx.x;
    int t = tok[rowmap ? rowmap[row] : row];          // (rowmap: dX holds the listed positions only)
    t = min(max(t, 0), V1 - 1);
    // Rows whose gradient is exactly zero add nothing: the label positions behind a caption's end (zero-padded labels, train.py:298: half of
    // all (t, n) positions at S = 20) all carry token 0, and their 512-wide atomics serialise on the <bos> / padding row of the table while
    // contributing 0.  One block-wide test skips them (NaN counts as non-zero and still propagates).
    int any = 0;
    for (int j = threadIdx.x; j < E; j += blockDim.x) any |= dX[(long)row * E + j] != 0.f;
    if (!__syncthreads_or(any)) return;
    for (int j = threadIdx.x; j < E; j += blockDim.x) atomicAdd(&gW[(long)t * E + j], dX[(long)row * E + j]);
}
// fixed-order scatter-add ("deterministic" = 1): workgroup b owns the table rows of tokens [b * tpb, (b + 1) * tpb), walks ALL gradient rows in
// order, 256 at a time (their tokens staged through LDS), and adds the rows of its tokens into the table with plain read-modify-writes -- every
// table row has one writer and receives its rows in row order
__global__ __launch_bounds__(256) void embed_scatter_det_kernel(const float* __restrict__ dX, const int* __restrict__ tok, float* __restrict__ gW,
                                                                int rows, int E, int V1, const int* __restrict__ rowmap, int tpb) {
    __shared__ int stok[256];
    const int t0 = blockIdx.x * tpb, t1 = min(V1, t0 + tpb);
    for (int r0 = 0; r0 < rows; r0 += 256) {
        const int r = r0 + threadIdx.x;
        int t = -1;
        if (r < rows) { t = tok[rowmap ? rowmap[r] : r]; t = min(max(t, 0), V1 - 1); }
        const int mine = (t >= t0 && t < t1) ? 1 : 0;
        __syncthreads();          // the previous round's readers of stok are done
        stok[threadIdx.x] = mine ? t : -1;
        if (!__syncthreads_or(mine)) continue;
        const int nr = min(256, rows - r0);
        for (int i = 0; i < nr; ++i) {
            const int ti = stok[i];
            if (ti < 0) continue;
            for (int j = threadIdx.x; j < E; j += 256) gW[(long)ti * E + j] += dX[(long)(r0 + i) * E + j];
        }
    }
}
int embed_scatter_add(const float* dX, const int* tok, float* gW, int rows, int E, int V1, hipStream_t st, const int* rowmap) {
    if (config().diag_skip & 8) return 0;
    if (det_mode()) {
        const int tpb = max(1, (V1 + 1023) / 1024);
        hipLaunchKernelGGL(embed_scatter_det_kernel, dim3((V1 + tpb - 1) / tpb), dim3(256), 0, st, dX, tok, gW, rows, E, V1, rowmap, tpb);
        return check_launch("embed_scatter_det");
    }
    hipLaunchKernelGGL(embed_scatter_add_kernel, dim3(rows), dim3(128), 0, st, dX, tok, gW, rows, E, V1, rowmap);
    return check_launch("embed_scatter_add");
}

// ---- row log-softmax (in place) ---------------------------------------------------------------------
// rows are (n, t) pairs of a [N,S,cols] tensor restricted to timesteps [t0, t0+nt): block b -> n = b / nt, t = t0 + b % nt
__global__ __launch_bounds__(256) void logsoftmax_rows_kernel(float* __restrict__ X, long ld, int S, int t0, int nt, int cols) {
    __shared__ float red[4];
    float* x = X + ((long)(blockIdx.x / nt) * S + t0 + blockIdx.x % nt) * ld;
    float m = -INFINITY;
    for (int j = threadIdx.x; j < cols; j += 256) m = fmaxf(m, x[j]);
    m = block_max(m, red);
    float s = 0.f;
    for (int j = threadIdx.x; j < cols; j += 256) s += expf(x[j] - m);
    s = block_sum(s, red);
    const float lse = m + logf(s);
    for (int j = threadIdx.x; j < cols; j += 256) x[j] = x[j] - lse;
}
// same arithmetic with the row held in registers (one read, one write): rows of up to 256 * EPT columns
template <int EPT>
__global__ __launch_bounds__(256) void logsoftmax_rows_reg_kernel(float* __restrict__ X, long ld, int S, int t0, int nt, int cols) {
    __shared__ float red[4];
    float* x = X + ((long)(blockIdx.x / nt) * S + t0 + blockIdx.x % nt) * ld;
    float v[EPT];
    float m = -INFINITY;
#pragma unroll
    for (int i = 0; i < EPT; ++i) {
        const int j = threadIdx.x + i * 256;
        v[i] = j < cols ? x[j] : -INFINITY;
        m = fmaxf(m, v[i]);
    }
    m = block_max(m, red);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < EPT; ++i) s += ((int)threadIdx.x + i * 256 < cols) ? expf(v[i] - m) : 0.f;
    s = block_sum(s, red);
    const float lse = m + logf(s);
#pragma unroll
    for (int i = 0; i < EPT; ++i) {
        const int j = threadIdx.x + i * 256;
        if (j < cols) x[j] = v[i] - lse;
    }
}
int logsoftmax_rows(float* X, long ld, int N, int S, int t0, int nt, int cols, hipStream_t st) {
    if (cols <= 256 * 8) hipLaunchKernelGGL(logsoftmax_rows_reg_kernel<8>, dim3(N * nt), dim3(256), 0, st, X, ld, S, t0, nt, cols);
    else if (cols <= 256 * 20) hipLaunchKernelGGL(logsoftmax_rows_reg_kernel<20>, dim3(N * nt), dim3(256), 0, st, X, ld, S, t0, nt, cols);
    else if (cols <= 256 * 40) hipLaunchKernelGGL(logsoftmax_rows_reg_kernel<40>, dim3(N * nt), dim3(256), 0, st, X, ld, S, t0, nt, cols);
    else hipLaunchKernelGGL(logsoftmax_rows_kernel, dim3(N * nt), dim3(256), 0, st, X, ld, S, t0, nt, cols);
    return check_launch("logsoftmax_rows");
}

// d logits (time-major rows t*N+n, leading dim ldo, zero padded) from log-probs [N,S,V1] and either a
// dense upstream gradient G [N,S,V1] or the fused masked-NLL gradient (target/mask/g_loss/inv_den).
// target indices arrive as int32 or, straight from the reference's LongTensor labels, as int64 (no conversion pass)
__device__ __forceinline__ int load_index(const void* p, long i, int is64) {
    return is64 ? (int)reinterpret_cast<const long long*>(p)[i] : reinterpret_cast<const int*>(p)[i];
}
__global__ __launch_bounds__(256) void logsoftmax_bwd_kernel(const float* __restrict__ logp, const float* __restrict__ G,
                                                             const void* __restrict__ target, int tgt64, const float* __restrict__ mask,
                                                             const float* __restrict__ g_loss, const float* __restrict__ mask_sum,
                                                             float* __restrict__ out, long ldo, int N, int S, int V1) {
    __shared__ float red[4];
    const int row = blockIdx.x;             // time-major row = t*N + n
    const int t = row / N, n = row % N;
    const long src = ((long)n * S + t) * V1;
    float* o = out + (long)row * ldo;
    if (G) {
        float s = 0.f;
        for (int j = threadIdx.x; j < V1; j += 256) s += G[src + j];
        s = block_sum(s, red);
        for (int j = threadIdx.x; j < V1; j += 256) o[j] = G[src + j] - expf(logp[src + j]) * s;
    } else {
        // loss = -sum(logp[target]*mask)/(sum(mask)+1e-6)  =>  g[target] = -mask/(den) * g_loss, other entries 0
        const float gv = -mask[n * S + t] / (mask_sum[0] + 1e-6f) * g_loss[0];
        const int tg = load_index(target, n * S + t, tgt64);
        for (int j = threadIdx.x; j < V1; j += 256) o[j] = (j == tg ? gv : 0.f) - expf(logp[src + j]) * gv;
    }
    for (int j = V1 + threadIdx.x; j < ldo; j += 256) o[j] = 0.f;
}
// dense-G form with the row of G held in registers: one read of G and logp, one write (rows of up to 256 * EPT columns)
template <int EPT>
__global__ __launch_bounds__(256) void logsoftmax_bwd_reg_kernel(const float* __restrict__ logp, const float* __restrict__ G,
                                                                 float* __restrict__ out, long ldo, int N, int S, int V1) {
    __shared__ float red[4];
    const int row = blockIdx.x;             // time-major row = t*N + n
    const int t = row / N, n = row % N;
    const long src = ((long)n * S + t) * V1;
    float* o = out + (long)row * ldo;
    float g[EPT], lp[EPT];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < EPT; ++i) {
        const int j = threadIdx.x + i * 256;
        g[i] = j < V1 ? G[src + j] : 0.f;
        lp[i] = j < V1 ? logp[src + j] : -INFINITY;
        s += g[i];
    }
    s = block_sum(s, red);
#pragma unroll
    for (int i = 0; i < EPT; ++i) {
        const int j = threadIdx.x + i * 256;
        if (j < V1) o[j] = g[i] - expf(lp[i]) * s;
        else if (j < ldo) o[j] = 0.f;
    }
}
int logsoftmax_bwd(const float* logp, const float* G, const void* target, int tgt64, const float* mask, const float* g_loss,
                   const float* mask_sum, float* out, long ldo, int N, int S, int V1, hipStream_t st) {
    if (G && ldo <= 256 * 20 && ldo > 256 * 8) hipLaunchKernelGGL(logsoftmax_bwd_reg_kernel<20>, dim3(N * S), dim3(256), 0, st, logp, G, out, ldo, N, S, V1);
    else if (G && ldo <= 256 * 8) hipLaunchKernelGGL(logsoftmax_bwd_reg_kernel<8>, dim3(N * S), dim3(256), 0, st, logp, G, out, ldo, N, S, V1);
    else hipLaunchKernelGGL(logsoftmax_bwd_kernel, dim3(N * S), dim3(256), 0, st, logp, G, target, tgt64, mask, g_loss, mask_sum, out, ldo, N, S, V1);
    return check_launch("logsoftmax_bwd");
}

// log-softmax + masked NLL + its gradient in ONE pass over the logits (echr_train_step: nothing else reads the log-probs, so they are never
// written): block = row (n, t) of the [N,S,ld] logits held in registers -> lse; d logits row (time-major t*N+n, leading dimension ldo, zero
// padded) = (onehot(target) - softmax) * gv with gv = -mask[n,t] / (sum(mask) + 1e-6) * g_loss (misc/utils.py:66-75 and its backward);
// row_loss[t*N+n] = -logp[target] * mask[n,t].  Every block sums the N*S mask entries itself (same fixed order everywhere).
template <int EPT>
__global__ __launch_bounds__(256) void logsoftmax_nll_dlg_kernel(const float* __restrict__ X, long ld, const void* __restrict__ target, int tgt64,
                                                                 const float* __restrict__ mask, const float* __restrict__ g_loss, float* __restrict__ out,
                                                                 long ldo, float* __restrict__ row_loss, float* __restrict__ msum_out, int N, int S, int V1,
                                                                 const int* __restrict__ act) {
    __shared__ float red[4];
    // act != nullptr: block i handles the compacted row i = time-major row act[i] (only rows whose mask is non-zero exist: the logits arrive
    // as [n_active, ld], d logits / row_loss leave in the same compact order)
    const int row = blockIdx.x;             // output row
    const int tm = act ? act[row] : row;    // time-major row = t*N + n
    const int t = tm / N, n = tm % N;
    const float* x = act ? X + (long)row * ld : X + ((long)n * S + t) * ld;
    float v[EPT];
    float m = -INFINITY;
#pragma unroll
    for (int i = 0; i < EPT; ++i) {
        const int j = threadIdx.x + i * 256;
        v[i] = j < V1 ? x[j] : -INFINITY;
        m = fmaxf(m, v[i]);
    }
    float ms = 0.f;
    for (int i = threadIdx.x; i < N * S; i += 256) ms += mask[i];
    const int tg = min(max(load_index(target, n * S + t, tgt64), 0), V1 - 1);
    const float mk = mask[n * S + t];
    m = block_max(m, red);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < EPT; ++i) s += ((int)threadIdx.x + i * 256 < V1) ? expf(v[i] - m) : 0.f;
    s = block_sum(s, red);
    ms = block_sum(ms, red);
    const float lse = m + logf(s);
    const float gv = -mk / (ms + 1e-6f) * g_loss[0];
    float* o = out + (long)row * ldo;
#pragma unroll
    for (int i = 0; i < EPT; ++i) {
        const int j = threadIdx.x + i * 256;
        if (j < V1) {
            const float lp = v[i] - lse;
            o[j] = (j == tg ? gv : 0.f) - expf(lp) * gv;
            if (j == tg) row_loss[row] = -lp * mk;
        } else if (j < ldo) o[j] = 0.f;
    }
    if (row == 0 && threadIdx.x == 0) msum_out[0] = ms;
}
// loss[0] = sum(row_loss) / (msum + 1e-6), loss[1] = msum: one block, fixed order
__global__ __launch_bounds__(256) void nll_rows_sum_kernel(const float* __restrict__ row_loss, int NS, const float* __restrict__ msum, float* __restrict__ loss) {
    __shared__ float red[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < NS; i += 256) s += row_loss[i];
    s = block_sum(s, red);
    if (threadIdx.x == 0) { loss[0] = s / (msum[0] + 1e-6f); loss[1] = msum[0]; }
}
bool logsoftmax_nll_dlg_ok(int V1, long ldo) { return ldo <= 256 * 40; }
int logsoftmax_nll_dlg(const float* X, long ld, const void* target, int tgt64, const float* mask, const float* g_loss, float* out, long ldo,
                       float* row_loss, float* msum_out, int N, int S, int V1, hipStream_t st, const int* act, int n_active) {
    const int rows = act ? n_active : N * S;
    if (ldo <= 256 * 8) hipLaunchKernelGGL(logsoftmax_nll_dlg_kernel<8>, dim3(rows), dim3(256), 0, st, X, ld, target, tgt64, mask, g_loss, out, ldo, row_loss, msum_out, N, S, V1, act);
    else if (ldo <= 256 * 20) hipLaunchKernelGGL(logsoftmax_nll_dlg_kernel<20>, dim3(rows), dim3(256), 0, st, X, ld, target, tgt64, mask, g_loss, out, ldo, row_loss, msum_out, N, S, V1, act);
    else hipLaunchKernelGGL(logsoftmax_nll_dlg_kernel<40>, dim3(rows), dim3(256), 0, st, X, ld, target, tgt64, mask, g_loss, out, ldo, row_loss, msum_out, N, S, V1, act);
    return check_launch("logsoftmax_nll_dlg");
}
int nll_rows_sum(const float* row_loss, int NS, const float* msum, float* loss, hipStream_t st) {
    hipLaunchKernelGGL(nll_rows_sum_kernel, dim3(1), dim3(256), 0, st, row_loss, NS, msum, loss);
    return check_launch("nll_rows_sum");
}

// masked NLL (misc/utils.py:66-75): out[0] = loss, out[1] = sum(mask)
__global__ __launch_bounds__(256) void nll_loss_kernel(const float* __restrict__ logp, const void* __restrict__ target, int tgt64,
                                                       const float* __restrict__ mask, float* __restrict__ out, int NS, int V1) {
    __shared__ float red[4];
    float s = 0.f, ms = 0.f;
    // four rows per round: targets and masks first, then the four gathered log-probs in flight together (the gather depends on its target:
    // one row per iteration is a chain of 2 NS / 256 dependent latencies on the forward's critical path); same order of additions
    for (int i0 = threadIdx.x; i0 < NS; i0 += 1024) {
        float mk[4], lv[4];
        int tg[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + 256 * u;
            mk[u] = i < NS ? mask[i] : 0.f;
            tg[u] = i < NS ? min(max(load_index(target, i, tgt64), 0), V1 - 1) : 0;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int i = i0 + 256 * u; lv[u] = i < NS ? logp[(long)i * V1 + tg[u]] : 0.f; }
#pragma unroll
        for (int u = 0; u < 4; ++u) if (i0 + 256 * u < NS) { s -= lv[u] * mk[u]; ms += mk[u]; }
    }
    s = block_sum(s, red);
    ms = block_sum(ms, red);
    if (threadIdx.x == 0) { out[0] = s / (ms + 1e-6f); out[1] = ms; }
}

// backward of the masked NLL as ONE pass: g_logp[n,s,:] = 0 except g_logp[n,s,target] = -mask / (sum(mask) + 1e-6) * g_loss
__global__ __launch_bounds__(256) void nll_loss_bwd_kernel(const void* __restrict__ target, int tgt64, const float* __restrict__ mask,
                                                           const float* __restrict__ fwd_out, const float* __restrict__ g_loss,
                                                           float* __restrict__ g_logp, int V1) {
    const long row = blockIdx.x;
    float4* o = reinterpret_cast<float4*>(g_logp + row * V1);
    const float gv = -mask[row] / (fwd_out[1] + 1e-6f) * g_loss[0];
    const int tg = min(max(load_index(target, row, tgt64), 0), V1 - 1);
    float* of = g_logp + row * V1;
    for (int j = threadIdx.x; j < V1; j += 256) of[j] = (j == tg) ? gv : 0.f;
    (void)o;
}

// ---- event pooling + anchor gather (CaptionGenerator.py:111-114,121,128) -----------------------------
// grid (N, ceil(D/64)): a workgroup pools one event over one 64-column chunk -- 16 float4 lanes across the columns (256-byte row
// segments), 16 row groups across the other threads (eight or so independent loads in flight per thread), LDS combine.
__global__ __launch_bounds__(256) void event_pool_gather_kernel(const float* __restrict__ c3d, const float* __restrict__ tap,
                                                                const int* __restrict__ ev_start, const int* __restrict__ ev_len,
                                                                const int* __restrict__ ind, float* __restrict__ ech, int D, int Ht, int vec) {
    __shared__ __attribute__((aligned(16))) float red[16][64];
    const int n = blockIdx.x, c0 = blockIdx.y * 64;
    const int s = ev_start[n], len = ev_len[n];
    float* o = ech + (long)n * (D + Ht);
    const int c = threadIdx.x & 15, rg = threadIdx.x >> 4;
    const int col = c0 + 4 * c;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (vec) {
        if (col < D) {
#pragma unroll 4
            for (int a = rg; a < len; a += 16) {
                const float4 v = *reinterpret_cast<const float4*>(c3d + (long)(s + a) * D + col);
                acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
            }
        }
    } else {
        for (int a = rg; a < len; a += 16) {
            const float* r = c3d + (long)(s + a) * D + col;
            if (col < D) acc.x += r[0];
            if (col + 1 < D) acc.y += r[1];
            if (col + 2 < D) acc.z += r[2];
            if (col + 3 < D) acc.w += r[3];
        }
    }
    *reinterpret_cast<float4*>(&red[rg][4 * c]) = acc;
    __syncthreads();
    if (threadIdx.x < 64 && c0 + (int)threadIdx.x < D) {
        float t = 0.f;
#pragma unroll
        for (int g = 0; g < 16; ++g) t += red[g][threadIdx.x];
        o[c0 + threadIdx.x] = t / (float)len;
    }
    // the anchor's SST state: the chunks share the Ht columns
    if (Ht == 0) return;
    const long trow = ind[n];
    const int per = (Ht + gridDim.y - 1) / gridDim.y, j0 = blockIdx.y * per;
    for (int j = j0 + threadIdx.x; j < min(Ht, j0 + per); j += 256) o[D + j] = tap[trow * Ht + j];
}
__global__ void event_gather_bwd_kernel(const float* __restrict__ d_ech, const int* __restrict__ ind, float* __restrict__ d_tap,
                                        int D, int Ht) {
    const int n = blockIdx.x;
    const long trow = ind[n];
    for (int j = threadIdx.x; j < Ht; j += blockDim.x) atomicAdd(&d_tap[trow * Ht + j], d_ech[(long)n * (D + Ht) + D + j]);
}

// fixed-order form ("deterministic" = 1): the first event of every anchor row owns it and adds the events that share it in event order
__global__ void event_gather_bwd_det_kernel(const float* __restrict__ d_ech, const int* __restrict__ ind, float* __restrict__ d_tap,
                                            int D, int Ht, int N) {
    const int n = blockIdx.x;
    const long trow = ind[n];
    for (int m = 0; m < n; ++m)
        if (ind[m] == trow) return;          // uniform: an earlier event owns this row
    for (int j = threadIdx.x; j < Ht; j += blockDim.x) {
        float acc = 0.f;
        for (int m = n; m < N; ++m)
            if (ind[m] == trow) acc += d_ech[(long)m * (D + Ht) + D + j];
        d_tap[trow * Ht + j] += acc;
    }
}

// ---- greedy arg-max over logits rows: lowest index on ties (torch.max semantics, OldModel_NEW.py:158) ---
// Updates the sampler state: it_next[n] (int32), unfinished[n], seq/seq_logp column, n_unfinished counter.
// slabs != nullptr: the row is first formed from the nslab k-slice slabs of the logits product, four at a time in slab order, + bias -- one fixed
// order (bitwise reproducible), and written to `logits` by the thread that scans it -- the separate slab-sum launch of every decoder step folded in
template <int EPT>          // EPT > 0: the row is held in registers (V1 <= 256 EPT; every load of the thread in flight at once); 0: any V1, streamed
__global__ __launch_bounds__(256, 1) void greedy_step_kernel(float* __restrict__ logits, long ld, int V1, int t, int seq_len,
                                                          int* __restrict__ it_next, int* __restrict__ unfinished,
                                                          long long* __restrict__ seq, float* __restrict__ seq_logp,
                                                          int* __restrict__ n_unfinished, const float* __restrict__ slabs, long slab_stride,
                                                          const float* __restrict__ bias, int nslab) {
    __shared__ float red[4];
    __shared__ int redi[4];
    const int n = blockIdx.x;
    float* x = logits + (long)n * ld;
    float m = -INFINITY;
    int mi = 0x7fffffff;
    float rv[EPT > 0 ? EPT : 1];
    if (EPT > 0) {
        // a strided scan with one load per iteration is a chain of ~20 dependent memory latencies (measured 19.5 us per step at V1 = 5001);
        // with the row in registers the loads overlap
        if (slabs) {
            // nslab (a multiple of four) k-slice slabs added four at a time in slab order, then the bias: rounds of four slabs' loads in flight
            if constexpr (EPT <= 24) {
                float s0[EPT > 0 ? EPT : 1], s1[EPT > 0 ? EPT : 1], s2[EPT > 0 ? EPT : 1], s3[EPT > 0 ? EPT : 1];
                for (int sb = 0; sb < nslab; sb += 4) {
#pragma unroll
                    for (int i = 0; i < EPT; ++i) {
                        const int j = threadIdx.x + 256 * i;
                        const float* sp = slabs + (long)sb * slab_stride + (long)n * V1 + (j < V1 ? j : 0);
                        s0[i] = sp[0]; s1[i] = sp[slab_stride]; s2[i] = sp[2 * slab_stride]; s3[i] = sp[3 * slab_stride];
                    }
#pragma unroll
                    for (int i = 0; i < EPT; ++i) {
                        const float part = (s0[i] + s1[i]) + (s2[i] + s3[i]);
                        rv[i] = sb == 0 ? part : rv[i] + part;
                    }
                }
            } else {
                // long rows: two slabs' loads in flight per round (register budget), the same order of additions
                float s0[EPT > 0 ? EPT : 1], s1[EPT > 0 ? EPT : 1], pa[EPT > 0 ? EPT : 1];
                for (int sb = 0; sb < nslab; sb += 4) {
#pragma unroll
                    for (int hf = 0; hf < 2; ++hf) {
#pragma unroll
                        for (int i = 0; i < EPT; ++i) {
                            const int j = threadIdx.x + 256 * i;
                            const float* sp = slabs + (long)(sb + 2 * hf) * slab_stride + (long)n * V1 + (j < V1 ? j : 0);
                            s0[i] = sp[0]; s1[i] = sp[slab_stride];
                        }
#pragma unroll
                        for (int i = 0; i < EPT; ++i) {
                            if (hf == 0) pa[i] = s0[i] + s1[i];
                            else { const float part = pa[i] + (s0[i] + s1[i]); rv[i] = sb == 0 ? part : rv[i] + part; }
                        }
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < EPT; ++i) { const int j = threadIdx.x + 256 * i; rv[i] += bias ? bias[j < V1 ? j : 0] : 0.f; }
#pragma unroll
            for (int i = 0; i < EPT; ++i) { const int j = threadIdx.x + 256 * i; if (j < V1) x[j] = rv[i]; }
        } else {
#pragma unroll
            for (int i = 0; i < EPT; ++i) { const int j = threadIdx.x + 256 * i; rv[i] = x[j < V1 ? j : 0]; }
        }
#pragma unroll
        for (int i = 0; i < EPT; ++i) {
            const int j = threadIdx.x + 256 * i;
            if (j < V1 && rv[i] > m) { m = rv[i]; mi = j; }   // ascending j per thread: first maximum kept
        }
    } else {
        for (int j = threadIdx.x; j < V1; j += 256) {
            float v;
            if (slabs) {
                v = 0.f;
                for (int sb = 0; sb < nslab; sb += 4) {
                    const float* sp = slabs + (long)sb * slab_stride + (long)n * V1 + j;
                    const float part = (sp[0] + sp[slab_stride]) + (sp[2 * slab_stride] + sp[3 * slab_stride]);
                    v = sb == 0 ? part : v + part;
                }
                v += bias ? bias[j] : 0.f;
                x[j] = v;
            } else v = x[j];
            if (v > m) { m = v; mi = j; }   // ascending j per thread: first maximum kept
        }
    }
    // wave arg-max with (value, lowest index) ordering
    for (int off = 32; off > 0; off >>= 1) {
        const float om = __shfl_xor(m, off, 64);
        const int oi = __shfl_xor(mi, off, 64);
        if (om > m || (om == m && oi < mi)) { m = om; mi = oi; }
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) { red[wave] = m; redi[wave] = mi; }
    __syncthreads();
    float bm = red[0];
    int bi = redi[0];
    for (int w = 1; w < 4; ++w)
        if (red[w] > bm || (red[w] == bm && redi[w] < bi)) { bm = red[w]; bi = redi[w]; }
    float s = 0.f;
    if (EPT > 0) {
#pragma unroll
        for (int i = 0; i < EPT; ++i) if (threadIdx.x + 256 * i < V1) s += expf(rv[i] - bm);      // same order of additions as the streamed form
    } else {
        for (int j = threadIdx.x; j < V1; j += 256) s += expf(x[j] - bm);
    }
    s = block_sum(s, red);
    if (threadIdx.x == 0) {
        // step t produced logits(t); the token fed at step t+1 is argmax -> sample position t (0-based)
        const float lp = -logf(s);              // log-softmax value at the maximum
        int un = (t == 0) ? 1 : unfinished[n];
        un = un && (bi > 0);
        unfinished[n] = un;
        const int tok = un ? bi : 0;
        it_next[n] = bi;   // the network keeps consuming the raw arg-max; only the emitted seq is masked (:181 runs after :171)
        if (t < seq_len) {
            seq[(long)n * seq_len + t] = tok;
            seq_logp[(long)n * seq_len + t] = lp;
        }
        if (un) atomicAdd(&n_unfinished[t + 1], 1);
    }
}

// torch.clamp semantics: a NaN gradient stays NaN (fminf/fmaxf alone would turn it into -clip and hide a diverged run)
__device__ __forceinline__ float clamp_keep_nan(float g, float clip) { return (g != g) ? g : fminf(fmaxf(g, -clip), clip); }

// ---- fused clamp + Adam (misc/utils.py:107-111 + torch.optim.Adam) ------------------------------------
__device__ __forceinline__ float4 nt_load4(const float4* a) {
    typedef float v4 __attribute__((ext_vector_type(4)));
    const v4 x = __builtin_nontemporal_load(reinterpret_cast<const v4*>(a));
    return make_float4(x[0], x[1], x[2], x[3]);
}
__device__ __forceinline__ void nt_store4(float4* a, const float4 x) {
    typedef float v4 __attribute__((ext_vector_type(4)));
    __builtin_nontemporal_store((v4){x.x, x.y, x.z, x.w}, reinterpret_cast<v4*>(a));
}
template <int NT>
__global__ __launch_bounds__(256) void clamp_adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                         float* __restrict__ v, long n, float lr_over_bc1, float inv_sqrt_bc2,
                                                         float omb1, float b2, float omb2, float eps, float clip,
                                                         const unsigned* __restrict__ abort_word, unsigned* __restrict__ applied) {
    // a persistent recurrence launch of this iteration gave up (hand-off timeout): its gradients are garbage, so the update is skipped;
    // the host reports -ETIME at its next library call (persist_check_async)
    if (abort_word && *abort_word) {
        if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(const_cast<unsigned*>(abort_word) + ABORT_SKIPPED_WORD, 1u);          // counted for echr_async_skipped_updates
        return;
    }
    // the caller's own count of updates that WERE applied (one word per optimiser state: echr_clamp_adam_counted)
    if (applied && blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(applied, 1u);
    const long n4 = n >> 2;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 P = NT >= 2 ? nt_load4(reinterpret_cast<const float4*>(p) + i) : reinterpret_cast<float4*>(p)[i];
        float4 G, M, V;
        if (NT) {          // g, m, v are streamed once per step and never re-read before they are rewritten: non-temporal
            G = nt_load4(reinterpret_cast<const float4*>(g) + i);
            M = nt_load4(reinterpret_cast<const float4*>(m) + i);
            V = nt_load4(reinterpret_cast<const float4*>(v) + i);
        } else {
            G = reinterpret_cast<const float4*>(g)[i];
            M = reinterpret_cast<float4*>(m)[i];
            V = reinterpret_cast<float4*>(v)[i];
        }
        float* pp = &P.x; float* gg = &G.x; float* mm = &M.x; float* vv = &V.x;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float gk = clamp_keep_nan(gg[k], clip);
            mm[k] = mm[k] + (gk - mm[k]) * omb1;            // torch: exp_avg.lerp_(grad, 1 - beta1)
            vv[k] = b2 * vv[k] + omb2 * gk * gk;            // torch: exp_avg_sq.mul_(beta2).addcmul_(g, g, value=1 - beta2)
            const float denom = sqrtf(vv[k]) * inv_sqrt_bc2 + eps;
            pp[k] -= lr_over_bc1 * (mm[k] / denom);
        }
        if (NT >= 3) nt_store4(reinterpret_cast<float4*>(p) + i, P); else reinterpret_cast<float4*>(p)[i] = P;
        if (NT) { nt_store4(reinterpret_cast<float4*>(m) + i, M); nt_store4(reinterpret_cast<float4*>(v) + i, V); }
        else { reinterpret_cast<float4*>(m)[i] = M; reinterpret_cast<float4*>(v)[i] = V; }
    }
    // tail
    const long base = n4 << 2;
    const long i = base + (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        float gk = clamp_keep_nan(g[i], clip);
        float mk = m[i] + (gk - m[i]) * omb1;
        float vk = b2 * v[i] + omb2 * gk * gk;
        m[i] = mk; v[i] = vk;
        p[i] -= lr_over_bc1 * (mk / (sqrtf(vk) * inv_sqrt_bc2 + eps));
    }
}

__global__ void clamp_kernel(float* g, long n, float clip, const unsigned* __restrict__ abort_word) {
    if (abort_word && *abort_word) return;          // (a clamp is not an optimiser update: not counted)
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) g[i] = clamp_keep_nan(g[i], clip);
}

// ---- multinomial step (OldModel_NEW.py:160-168, sample_max = 0): token j with probability exp(logp_j / T) / sum_j exp(logp_j / T) ---
// by inverse CDF: thread th owns the contiguous indices [CH th, CH th + CH), a block scan of the per-thread masses finds the owner of
// u * total, which then walks its chunk.  u comes from the library's counter-based Philox stream keyed by (seed, row, step), so a
// decode is reproducible for a given seed (it is NOT torch.multinomial's stream: only the distribution matches the reference).
// The emitted log-prob is the un-tempered log-softmax value of the sampled token (`logprobs.gather`, :167).
constexpr unsigned SITE_SAMPLE = 6;
__global__ __launch_bounds__(256) void sample_step_kernel(const float* __restrict__ logits, long ld, int V1, int t, int seq_len,
                                                          int* __restrict__ it_next, int* __restrict__ unfinished,
                                                          long long* __restrict__ seq, float* __restrict__ seq_logp,
                                                          int* __restrict__ n_unfinished, float inv_temp, unsigned k0, unsigned k1) {
    __shared__ float red[4];
    __shared__ float pre[257];
    const int n = blockIdx.x, th = threadIdx.x;
    const float* x = logits + (long)n * ld;
    const int CH = (V1 + 255) / 256, j0 = th * CH, j1 = min(V1, j0 + CH);
    float m = -INFINITY;
    for (int j = th; j < V1; j += 256) m = fmaxf(m, x[j]);
    m = block_max(m, red);
    float s = 0.f;
    for (int j = th; j < V1; j += 256) s += expf(x[j] - m);
    s = block_sum(s, red);
    const float lz = m + logf(s);                        // log-softmax: logp_j = x_j - lz
    float mass = 0.f;
    // (weights relative to the row maximum: softmax(logp / T) does not change, and the largest weight is 1 at any temperature)
    for (int j = j0; j < j1; ++j) mass += expf((x[j] - m) * inv_temp);
    pre[th + 1] = mass;
    if (th == 0) pre[0] = 0.f;
    __syncthreads();
    if (th == 0) {                                       // 256 sequential adds: one fixed order, negligible beside the row reads
        float acc = 0.f;
        for (int i = 1; i <= 256; ++i) { acc += pre[i]; pre[i] = acc; }
    }
    __syncthreads();
    const float total = pre[256];
    const unsigned w = philox_word((unsigned)n, (unsigned)t, SITE_SAMPLE, 0u, k0, k1);
    const float target = (float)(w >> 8) * (1.0f / 16777216.0f) * total;
    // the owner: first thread whose inclusive prefix exceeds the target (the last thread with mass when rounding pushes it past the end)
    const bool mine = (pre[th] <= target && target < pre[th + 1]) || (th == 255 && target >= total);
    if (mine) {
        float acc = pre[th];
        int pick = -1;
        for (int j = j0; j < j1; ++j) {
            acc += expf((x[j] - m) * inv_temp);
            if (acc > target) { pick = j; break; }
        }
        if (pick < 0) {                                  // rounding at the chunk end: the last index with non-zero mass up to here
            pick = max(0, min(V1, j1) - 1);
            while (pick > 0 && !(expf((x[pick] - m) * inv_temp) > 0.f)) --pick;
        }
        const float lp = x[pick] - lz;
        int un = (t == 0) ? 1 : unfinished[n];
        un = un && (pick > 0);
        unfinished[n] = un;
        it_next[n] = pick;
        if (t < seq_len) {
            seq[(long)n * seq_len + t] = un ? pick : 0;
            seq_logp[(long)n * seq_len + t] = lp;
        }
        if (un) atomicAdd(&n_unfinished[t + 1], 1);
    }
}

int sample_step(const float* logits, long ld, int N, int V1, int t, int seq_len, int* it_next, int* unfinished, long long* seq,
                float* seq_logp, int* n_unfinished, float temperature, unsigned long long seed, hipStream_t st) {
    const float inv_temp = 1.0f / (temperature > 0.f ? temperature : 1.0f);
    hipLaunchKernelGGL(sample_step_kernel, dim3(N), dim3(256), 0, st, logits, ld, V1, t, seq_len, it_next, unfinished, seq, seq_logp, n_unfinished,
                       inv_temp, (unsigned)(seed & 0xFFFFFFFFull), (unsigned)(seed >> 32));
    return check_launch("sample_step");
}

int greedy_step(float* logits, long ld, int N, int V1, int t, int seq_len, int* it_next, int* unfinished,
                long long* seq, float* seq_logp, int* n_unfinished, hipStream_t st, const float* slabs, long slab_stride, const float* bias, int nslab) {
    if (V1 <= 256 * 8) hipLaunchKernelGGL(greedy_step_kernel<8>, dim3(N), dim3(256), 0, st, logits, ld, V1, t, seq_len, it_next, unfinished, seq, seq_logp,
                                          n_unfinished, slabs, slab_stride, bias, nslab);
    else if (V1 <= 256 * 20) hipLaunchKernelGGL(greedy_step_kernel<20>, dim3(N), dim3(256), 0, st, logits, ld, V1, t, seq_len, it_next, unfinished, seq,
                                                seq_logp, n_unfinished, slabs, slab_stride, bias, nslab);
    else if (V1 <= 256 * 48) hipLaunchKernelGGL(greedy_step_kernel<48>, dim3(N), dim3(256), 0, st, logits, ld, V1, t, seq_len, it_next, unfinished, seq,
                                                seq_logp, n_unfinished, slabs, slab_stride, bias, nslab);          // vocabularies up to 12 288 (ActivityNet Captions: ~10 k)
    else hipLaunchKernelGGL(greedy_step_kernel<0>, dim3(N), dim3(256), 0, st, logits, ld, V1, t, seq_len, it_next, unfinished, seq, seq_logp,
                            n_unfinished, slabs, slab_stride, bias, nslab);
    return check_launch("greedy_step");
}

}  // namespace echr

using namespace echr;

extern "C" int echr_version(void) { return ECHR_ABI_VERSION; }
extern "C" int64_t echr_abi_sizeof(const char* name) {
    if (!name) return -1;
#define ECHR_SZ(T) if (!strcmp(name, #T)) return (int64_t)sizeof(T)
    ECHR_SZ(echr_gemm_desc); ECHR_SZ(echr_dropout); ECHR_SZ(echr_tsrm_args); ECHR_SZ(echr_tsrm_grads); ECHR_SZ(echr_dec_args); ECHR_SZ(echr_dec_grads);
    ECHR_SZ(echr_sample_args); ECHR_SZ(echr_sst_args); ECHR_SZ(echr_sst_grads); ECHR_SZ(echr_train_step_args);
    ECHR_SZ(echr_init_state_args); ECHR_SZ(echr_init_state_grads);
#undef ECHR_SZ
    return -1;
}
extern "C" const char* echr_last_error(void) { return g_err; }

// ---- non-zero initial decoder state (OldModel.init_hidden, OldModel_NEW.py:72-96) ----
// feats[n] = [video | event[n] | sum over the event's C3D rows / A]   (one block per event; A = the PADDED clip length of clip.mean(1))
__global__ __launch_bounds__(256) void init_feats_kernel(echr_init_state_args a, int Dtot) {
    const int n = blockIdx.x;
    float* o = a.feats + (long)n * Dtot;
    int off = 0;
    if (a.use_v) { for (int j = threadIdx.x; j < a.Dv; j += 256) o[j] = a.video[j]; off += a.Dv; }
    if (a.use_e) { for (int j = threadIdx.x; j < a.De; j += 256) o[off + j] = a.event[(long)n * a.De + j]; off += a.De; }
    if (a.use_c) {
        const int s = a.ev_start[n], len = a.ev_len[n];
        const float inv = 1.0f / (float)a.A;
        for (int j = threadIdx.x; j < a.D; j += 256) {
            float acc = 0.f;
            for (int r = 0; r < len; ++r) acc += a.c3d[(long)(s + r) * a.D + j];
            o[off + j] = acc * inv;
        }
    }
}
// g_event[n, :] += dfeats[n, off : off + De]
__global__ __launch_bounds__(256) void init_event_grad_kernel(const float* __restrict__ dfeats, float* __restrict__ g_event, int N, int De, int Dtot, int off) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)N * De) return;
    const int n = (int)(i / De), j = (int)(i % De);
    g_event[i] += dfeats[(long)n * Dtot + off + j];
}
static int init_dtot(const echr_init_state_args* a) { return (a->use_v ? a->Dv : 0) + (a->use_e ? a->De : 0) + (a->use_c ? a->D : 0); }
extern "C" int echr_init_state_fwd(const echr_init_state_args* a, void* stream) {
    ECHR_REQUIRE(a && a->N > 0 && a->H3 > 0 && a->w && a->b && a->feats && a->h0 && init_dtot(a) > 0, "init_state_fwd: bad arguments");
    ECHR_REQUIRE((!a->use_v || a->video) && (!a->use_e || a->event) && (!a->use_c || (a->c3d && a->ev_start && a->ev_len && a->A > 0)), "init_state_fwd: missing inputs");
    hipStream_t st = (hipStream_t)stream;
    const int Dtot = init_dtot(a);
    hipLaunchKernelGGL(init_feats_kernel, dim3(a->N), dim3(256), 0, st, *a, Dtot);
    if (int rc = check_launch("init_feats")) return rc;
    echr_gemm_desc d = desc_nt(a->feats, Dtot, a->w, Dtot, a->h0, a->H3, a->N, a->H3, Dtot);          // init_linear (:89)
    d.bias = a->b;
    return gemm(d, st);
}
extern "C" int echr_init_state_bwd(const echr_init_state_args* a, const echr_init_state_grads* g, void* stream) {
    ECHR_REQUIRE(a && g && g->g_h0 && g->g_w && g->g_b && g->dfeats && a->feats && a->w, "init_state_bwd: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    const int Dtot = init_dtot(a);
    const bool z = g->zeroed != 0;
    echr_gemm_desc d = desc_tn(g->g_h0, a->H3, a->feats, Dtot, g->g_w, Dtot, a->H3, Dtot, a->N);          // d W = d h0^T . feats
    d.beta = z ? 1.f : 0.f;
    if (int rc = gemm(d, st)) return rc;
    if (int rc = colsum(g->g_h0, a->H3, a->N, a->H3, g->g_b, z, st)) return rc;
    if (!(g->g_video && a->use_v) && !(g->g_event && a->use_e)) return 0;
    d = desc_nn(g->g_h0, a->H3, a->w, Dtot, g->dfeats, Dtot, a->N, Dtot, a->H3);                          // d feats = d h0 . W
    if (int rc = gemm(d, st)) return rc;
    int off = 0;
    if (a->use_v) {
        if (g->g_video) { if (int rc = colsum(g->dfeats, Dtot, a->N, a->Dv, g->g_video, false, st)) return rc; }          // the scene vector is shared by the N rows
        off += a->Dv;
    }
    if (a->use_e && g->g_event) {
        const long n = (long)a->N * a->De;
        hipLaunchKernelGGL(init_event_grad_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, g->dfeats, g->g_event, a->N, a->De, Dtot, off);
        return check_launch("init_event_grad");
    }
    return 0;
}

// ---- scene context 'VC' / 'VH' (CaptionGenerator.py:95-99): the mean over all T_v rows of c3d_feats / tap_feats ----
__global__ __launch_bounds__(256) void col_scale_kernel(float* __restrict__ v, int n, float s) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) v[i] *= s;
}
// gx[t, :] += g[:] / T  (the mean's gradient, broadcast over the rows)
__global__ __launch_bounds__(256) void col_mean_bwd_kernel(const float* __restrict__ g, float* __restrict__ gx, int rows, int cols, long ld, float inv) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)rows * cols) return;
    const int r = (int)(i / cols), c = (int)(i % cols);
    gx[(long)r * ld + c] += g[c] * inv;
}
extern "C" int echr_col_mean_fwd(const float* x, int32_t rows, int32_t cols, int64_t ld, float* out, void* stream) {
    ECHR_REQUIRE(x && out && rows > 0 && cols > 0 && ld >= cols, "col_mean_fwd: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    if (int rc = colsum(x, (long)ld, rows, cols, out, false, st)) return rc;
    hipLaunchKernelGGL(col_scale_kernel, dim3((cols + 255) / 256), dim3(256), 0, st, out, cols, 1.0f / (float)rows);
    return check_launch("col_mean_fwd");
}
extern "C" int echr_col_mean_bwd(const float* g, int32_t rows, int32_t cols, int64_t ld, float* gx, void* stream) {
    ECHR_REQUIRE(g && gx && rows > 0 && cols > 0 && ld >= cols, "col_mean_bwd: bad arguments");
    const long n = (long)rows * cols;
    hipLaunchKernelGGL(col_mean_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, g, gx, rows, cols, (long)ld, 1.0f / (float)rows);
    return check_launch("col_mean_bwd");
}

extern "C" int echr_event_pool_gather_fwd(const float* c3d, const float* tap, const int32_t* ev_start, const int32_t* ev_len,
                                          const int32_t* ind, float* ech, int32_t N, int32_t D, int32_t Ht, void* stream) {
    // D = 0: the anchors' SST states only ('ER2', CaptionGenerator.py:120-125); Ht = 0: the pooled C3D rows only ('ER1', :109-117)
    ECHR_REQUIRE(ev_start && ev_len && ind && ech && N > 0 && D >= 0 && Ht >= 0 && D + Ht > 0 && (c3d || D == 0) && (tap || Ht == 0), "event_pool_gather_fwd: bad arguments");
    const int vec = (D % 4 == 0) && ((uintptr_t)c3d % 16 == 0) && D >= 4;
    hipLaunchKernelGGL(event_pool_gather_kernel, dim3(N, D > 0 ? (D + 63) / 64 : 1), dim3(256), 0, (hipStream_t)stream, c3d, tap, ev_start, ev_len, ind, ech, D,
                       Ht, vec);
    return check_launch("event_pool_gather_fwd");
}
extern "C" int echr_event_pool_gather_bwd(const float* d_ech, const int32_t* ind, float* d_tap, int32_t N, int32_t D, int32_t Ht,
                                          void* stream) {
    ECHR_REQUIRE(d_ech && ind && d_tap && N > 0, "event_pool_gather_bwd: bad arguments");
    if (det_mode()) hipLaunchKernelGGL(event_gather_bwd_det_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream, d_ech, ind, d_tap, D, Ht, N);
    else hipLaunchKernelGGL(event_gather_bwd_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream, d_ech, ind, d_tap, D, Ht);
    return check_launch("event_pool_gather_bwd");
}

static int nll_fwd(const float* logp, const void* target, int tgt64, const float* mask, float* loss, int32_t N, int32_t S, int32_t V1, void* stream) {
    ECHR_REQUIRE(logp && target && mask && loss && N > 0 && S > 0 && V1 > 0, "nll_loss_fwd: bad arguments");
    hipLaunchKernelGGL(nll_loss_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, logp, target, tgt64, mask, loss, N * S, V1);
    return check_launch("nll_loss_fwd");
}
extern "C" int echr_nll_loss_fwd(const float* logp, const int32_t* target, const float* mask, float* loss, int32_t N, int32_t S,
                                 int32_t V1, void* stream) { return nll_fwd(logp, target, 0, mask, loss, N, S, V1, stream); }
extern "C" int echr_nll_loss_fwd_i64(const float* logp, const int64_t* target, const float* mask, float* loss, int32_t N, int32_t S,
                                     int32_t V1, void* stream) { return nll_fwd(logp, target, 1, mask, loss, N, S, V1, stream); }

extern "C" int echr_clamp_adam(float* p, const float* g, float* m, float* v, int64_t n, int32_t step, double lr, double beta1,
                               double beta2, double eps, float clip, void* stream) {
    return echr_clamp_adam_counted(p, g, m, v, n, step, lr, beta1, beta2, eps, clip, nullptr, stream);
}
extern "C" int echr_clamp_adam_counted(float* p, const float* g, float* m, float* v, int64_t n, int32_t step, double lr, double beta1,
                                       double beta2, double eps, float clip, uint32_t* applied, void* stream) {
    ECHR_REQUIRE(p && g && m && v && n > 0 && step >= 1, "clamp_adam: bad arguments");
    if (int rc = join_tail((hipStream_t)stream)) return rc;
    ECHR_REQUIRE(((uintptr_t)p % 16 == 0) && ((uintptr_t)g % 16 == 0) && ((uintptr_t)m % 16 == 0) && ((uintptr_t)v % 16 == 0),
                 "clamp_adam: buffers must be 16-byte aligned");
    if (config().diag_skip & 2) return 0;
    const double bc1 = 1.0 - pow(beta1, (double)step);
    const double bc2 = 1.0 - pow(beta2, (double)step);
    const long n4 = n >> 2;
    int grid = (int)min(max((n4 + 255) / 256, 1L), 4096L);
    static const int nt = [] { const char* e = getenv("ECHR_ADAM_NT"); return e ? atoi(e) : 2; }();      // A/B switch: 0 = cached accesses, 1 = g / m / v non-temporal, 2 = + the load of p, 3 = + its store
#define ECHR_ADAM_LAUNCH(L) hipLaunchKernelGGL(clamp_adam_kernel<L>, dim3(grid), dim3(256), 0, (hipStream_t)stream, p, g, m, v, (long)n, (float)(lr / bc1), \
                       (float)(1.0 / sqrt(bc2)), (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)eps, clip, persist_abort_word(), applied)
    if (nt >= 3) ECHR_ADAM_LAUNCH(3); else if (nt == 2) ECHR_ADAM_LAUNCH(2); else if (nt == 1) ECHR_ADAM_LAUNCH(1); else ECHR_ADAM_LAUNCH(0);
#undef ECHR_ADAM_LAUNCH
    return check_launch("clamp_adam");
}

extern "C" int echr_clamp(float* g, int64_t n, float clip, void* stream) {
    ECHR_REQUIRE(g && n > 0, "clamp: bad arguments");
    if (int rc = join_tail((hipStream_t)stream)) return rc;
    int grid = (int)min(((long)n + 255) / 256, 4096L);
    hipLaunchKernelGGL(clamp_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, g, (long)n, clip, persist_abort_word());
    return check_launch("clamp");
}

extern "C" int echr_prof_enable(int on) {
    prof_resolve();
    g_prof_on = on != 0;
    if (on) for (int k = 0; k < PROF_KINDS; ++k) { g_prof_ms[k] = g_prof_flops[k] = g_prof_bytes[k] = 0.0; g_prof_n[k] = 0; }
    return 0;
}
extern "C" int echr_prof_read(int kind, double* ms, double* flops, double* bytes, int64_t* launches) {
    ECHR_REQUIRE(kind >= 0 && kind < PROF_KINDS && ms && flops && bytes && launches, "prof_read: bad arguments");
    prof_resolve();
    *ms = g_prof_ms[kind]; *flops = g_prof_flops[kind]; *bytes = g_prof_bytes[kind]; *launches = g_prof_n[kind];
    return 0;
}

// cost of one ProfScope event pair around NOTHING on an idle stream (the fixed per-launch overhead of event timing)
extern "C" int echr_prof_event_overhead(double* ms, int64_t* n) {
    if (!ms || !n) return -22;
    const int reps = 200;
    hipEvent_t e0[reps], e1[reps];
    hipStream_t st = nullptr;
    (void)hipDeviceSynchronize();
    for (int i = 0; i < reps; ++i) {
        (void)hipEventCreate(&e0[i]); (void)hipEventCreate(&e1[i]);
        (void)hipEventRecord(e0[i], st); (void)hipEventRecord(e1[i], st);
    }
    (void)hipDeviceSynchronize();
    double tot = 0;
    for (int i = 0; i < reps; ++i) {
        float t = 0.f;
        (void)hipEventElapsedTime(&t, e0[i], e1[i]);
        tot += t;
        (void)hipEventDestroy(e0[i]); (void)hipEventDestroy(e1[i]);
    }
    *ms = tot; *n = reps;
    return 0;
}

extern "C" int echr_check_async(void) { return persist_check_async(); }
extern "C" int64_t echr_async_skipped_updates(void) { return (int64_t)persist_take_skipped_updates(); }

extern "C" int echr_persist_read_stamps(uint64_t* dst, int32_t max_entries) {
    return persist_read_stamps(reinterpret_cast<unsigned long long*>(dst), max_entries);
}

extern "C" int echr_config_set(const char* key, int32_t value) {
    ECHR_REQUIRE(key, "config_set: null key");
    Config& c = config();
    if (!strcmp(key, "gemm_bf16x3")) c.gemm_bf16x3 = value;
    else if (!strcmp(key, "overlap")) c.overlap = value;
    else if (!strcmp(key, "chains2")) c.chains2 = value;
    else if (!strcmp(key, "gemm_h2")) c.gemm_h2 = value;
    else if (!strcmp(key, "persist")) c.persist = value;
    else if (!strcmp(key, "persist_stamps")) c.persist_stamps = value;
    else if (!strcmp(key, "persist_bwd")) c.persist_bwd = value;
    else if (!strcmp(key, "persist_split")) c.persist_split = value;
    else if (!strcmp(key, "persist_h2")) c.persist_h2 = value;
    else if (!strcmp(key, "persist_merge")) c.persist_merge = value;
    else if (!strcmp(key, "persist_kgroups")) c.persist_kgroups = value;
    else if (!strcmp(key, "tsrm_fork")) c.tsrm_fork = value;
    else if (!strcmp(key, "persist_coop")) c.persist_coop = value;
    else if (!strcmp(key, "sst_persist")) c.sst_persist = value;
    else if (!strcmp(key, "tail_early")) c.tail_early = value;
    else if (!strcmp(key, "diag_skip")) c.diag_skip = value;
    else if (!strcmp(key, "embed_fused")) c.embed_fused = value;
    else if (!strcmp(key, "persist_sample")) c.persist_sample = value;
    else if (!strcmp(key, "posemb_rows")) c.posemb_rows = value;
    else if (!strcmp(key, "gemm_skinny")) c.gemm_skinny = value;
    else if (!strcmp(key, "posemb_packed")) c.posemb_packed = value;
    else if (!strcmp(key, "pair_tables")) c.pair_tables = value;
    else if (!strcmp(key, "persist_sample_force_eos")) c.persist_sample_force_eos = value;
    else if (!strcmp(key, "persist_sample_max")) c.persist_sample_max = value;
    else if (!strcmp(key, "persist_inject_timeout")) c.persist_inject_timeout = value;
    else if (!strcmp(key, "persist_spin_limit")) c.persist_spin_limit = value;
    else if (!strcmp(key, "deterministic")) c.deterministic = value ? 1 : 0;
    else if (!strcmp(key, "gemm_tile")) c.gemm_tile = value;          // tuning only: ASCII code of the tile selector ('1','6','a','b','c','s'), 0 = heuristics
    else if (!strcmp(key, "gemm_split")) c.gemm_split = value;        // tuning only: forced k-slice count of auto-split products, 0 = heuristics
    else if (!strcmp(key, "att_slots")) { ECHR_REQUIRE(value == 2 || value == 4 || value == 8, "config_set: att_slots must be 2, 4 or 8"); c.att_slots = value; }
    else { set_error("config_set: unknown key %s", key); return -22; }
    return 0;
}

static int nll_bwd(const void* target, int tgt64, const float* mask, const float* fwd_out, const float* g_loss, float* g_logp,
                   int32_t N, int32_t S, int32_t V1, void* stream) {
    ECHR_REQUIRE(target && mask && fwd_out && g_loss && g_logp && N > 0 && S > 0 && V1 > 0, "nll_loss_bwd: bad arguments");
    hipLaunchKernelGGL(nll_loss_bwd_kernel, dim3(N * S), dim3(256), 0, (hipStream_t)stream, target, tgt64, mask, fwd_out, g_loss, g_logp, V1);
    return check_launch("nll_loss_bwd");
}
extern "C" int echr_nll_loss_bwd(const int32_t* target, const float* mask, const float* fwd_out, const float* g_loss, float* g_logp,
                                 int32_t N, int32_t S, int32_t V1, void* stream) { return nll_bwd(target, 0, mask, fwd_out, g_loss, g_logp, N, S, V1, stream); }
extern "C" int echr_nll_loss_bwd_i64(const int64_t* target, const float* mask, const float* fwd_out, const float* g_loss, float* g_logp,
                                     int32_t N, int32_t S, int32_t V1, void* stream) { return nll_bwd(target, 1, mask, fwd_out, g_loss, g_logp, N, S, V1, stream); }
