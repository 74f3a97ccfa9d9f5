// Internal (C++) interfaces between the translation units of libechr_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/echr_hip.h"

namespace echr {

// Optional per-kernel-class timing with HIP events on the launch stream (bench.py's roofline leg).
// Disabled by default: ProfScope is then a no-op.  Events are resolved in echr_prof_read after a stream sync.
enum ProfKind { PROF_GEMM = 0, PROF_ATT_FWD = 1, PROF_ATT_BWD = 2, PROF_ATT_POST = 3, PROF_LSTM = 4, PROF_OTHER = 5, PROF_GEMM_SPLIT = 6, PROF_GEMM_H2 = 7, PROF_PACK = 8, PROF_PERSIST = 9, PROF_SST = 10, PROF_KINDS = 11 };
struct ProfScope {
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipStream_t st;
    ProfScope(int kind, double flops, double bytes, hipStream_t st);
    ~ProfScope();
    int kind; double flops, bytes;
};

// runtime switches (initial values from the environment, changeable through echr_config_set)
struct Config { int gemm_bf16x3; int overlap; int att_slots; int chains2; int gemm_h2; int persist; int persist_stamps; int gemm_tile; int gemm_split; int persist_bwd; int persist_split; int persist_h2; int persist_merge; int persist_kgroups; int tsrm_fork; int persist_coop; int persist_inject_timeout; int persist_spin_limit; int sst_persist; int tail_early; int diag_skip; int embed_fused; int persist_sample; int posemb_rows; int gemm_skinny; int posemb_packed; int pair_tables; int persist_sample_force_eos; int persist_sample_max; int deterministic; };
// diag_skip (diagnostic, tools/skip_bounds.py; results are WRONG while a bit is set): 1 = h2 operand packs, 2 = clamp+Adam kernel, 4 = att_post, 16 = every fp32-path product (gemm_f32 / t128 / bf16x3), 32 = every h2 product, 64 / 128 = the h2m16 product kernel loads only / computes only (tools/h2_ablate.py),
// 8 = embedding scatter-add -- the launch is skipped, which bounds what removing / hiding that work could gain
Config& config();

int gemm(const echr_gemm_desc& d, hipStream_t st);
// while one is alive on the calling thread, auto split-K (fp32 atomics, order-dependent last bits) is disabled: every product is a
// single fixed-order k loop per output tile
struct DeterministicScope { DeterministicScope(); ~DeterministicScope(); };
bool deterministic_gemm();
// "deterministic" = 1 (echr_config_set / ECHR_DETERMINISTIC): every order-dependent sum of a training iteration (fp32 atomics: split-K epilogues,
// k-slices of the recurrence products, column sums, scatter-adds, the attention backward's shared rows, the persistent kernels' exchange adds) is
// replaced by a fixed-order one -- two runs on the same inputs then agree bit for bit.  Slower; the persistent recurrence kernels are not used.
inline bool det_mode() { return config().deterministic != 0; }
// scratch slabs of the fixed-order variants: one lazily grown device buffer per use (`kind`), owned by the library, never shrunk; the call
// that grows one synchronises the device (hipFree / hipMalloc).  One calling thread, like the rest of the library's static state.
enum { DET_DQ = 0, DET_DPALL = 1, DET_ALPHA = 2, DET_REC = 3, DET_KINDS = 4 };
float* det_scratch(int kind, size_t floats);

// persistent (one launch for all timesteps) recurrence of the decoder, csrc/persist.hip
struct DropCfg;
struct PersistFwdBufs { float* GATES[3]; float* CS[3]; float *HS, *OUTD, *QS, *WT, *ATT, *PALL, *xws; bool prezeroed = false;
                        const float* EVB0 = nullptr; };      // EVB0: event part of stream 0's gates, still to be added (persist_fwd_adds_evb0)
bool persist_fwd_adds_evb0();          // the persistent forward launch can add EVB0 itself (fp16-pair LSTM role)
void persist_fwd_zero_range(const echr_dec_args* a, float* xws, float** ptr, long* count);
long persist_fwd_ws_floats(int S);
bool persist_fwd_eligible(const echr_dec_args* a);
int persist_fwd(const echr_dec_args* a, const PersistFwdBufs& B, const DropCfg& dh, const DropCfg& dout, hipStream_t st);
// The forward attention chain's weight images (fp16-pair B fragments of W_hh1 / W_att / W_h2a with their column scales: parameters only) built
// AHEAD of the launch into the exchange workspace, by a small kernel on `st`; the next persist_fwd on the same workspace copies them into LDS
// (2 x 64 KB per gate workgroup from L2) instead of converting them from the strided weight rows in front of its first step.  No-op (returns 0)
// for shapes / configurations the persistent fp16-pair forward kernels do not take.
int persist_fwd_prebuild(const echr_dec_args* a, float* xws, hipStream_t st);
struct PersistBwdBufs { const float* GATES[3]; const float* CS[3]; const float *QS, *WT, *ATT, *PALL, *DOUT; float* DG[3]; float *DQ, *DSC, *xws; bool prezeroed = false; };
void persist_bwd_zero_range(const echr_dec_args* a, float* xws, float** ptr, long* count);
long persist_bwd_ws_floats(int S);
bool persist_bwd_eligible(const echr_dec_args* a);
int persist_bwd(const echr_dec_args* a, const PersistBwdBufs& B, const DropCfg& dh, const DropCfg& dout, hipStream_t st);
// greedy decoding on the persistent kernels (64 events per launch, every step on device): see PersistS in csrc/persist.hip
struct PersistSampleBufs { float* PALL; const float* EVB0; const float* VIDB; float* xws; const float* TG[3]; const float* limg; float* sws; long long* seq; float* seq_logp; int* n_unfinished; };
long persist_sample_ws_floats(int S, int V1);          // per group of 64 events
long persist_sample_x_floats(int S);                   // per group of 64 events
long persist_logit_image_floats(int V1);
bool persist_sample_shape_ok(const echr_dec_args* a);
bool persist_sample_eligible(const echr_dec_args* a);
int persist_logit_image(const float* w_logit, int V1, float* img, hipStream_t st);
int persist_sample(const echr_dec_args* a, const PersistSampleBufs& B, hipStream_t st);
int persist_check_async();
long long persist_take_skipped_updates();          // updates the optimiser kernels skipped under the abort the last -62 reported (cleared by the read)
void coop_refused(const char* who, const char* why);          // one stderr line the first time a cooperative launch is refused
// device word that is non-zero from the moment a persistent launch aborts until the host has reported it (persist_check_async):
// kernels that would apply results (clamp_adam, clamp) skip their update while it is set; nullptr when the state is unavailable
const unsigned* persist_abort_word();
// word of the 256-byte abort block in which the optimiser kernels count the updates they skipped while the abort word was set
constexpr int ABORT_SKIPPED_WORD = 16;
unsigned* persist_host_flag();          // device view of the host-mapped flag persist_check_async reads (nullptr when unavailable)
// the library's helper stream outside a backward pass (the one the asynchronous decoder-backward tail uses): `aux_fork` makes it continue
// after everything queued on `from` and returns it, `aux_join` makes `to` wait for what was queued on it since
// Flags of every event the library uses to order its own streams on ONE device: no timing, and no system-scope fence -- the default event
// release writes the caches back and invalidates them for the host and other devices, which neither a stream-to-stream edge nor the staging
// ring's "this copy kernel has finished" needs (ECHR_EVENT_SYSTEM_FENCE=1 restores the default)
inline unsigned sync_event_flags() {
    static const unsigned f = [] {
        const char* e = getenv("ECHR_EVENT_SYSTEM_FENCE");
        if (e && e[0] == '1') return (unsigned)hipEventDisableTiming;
        return (unsigned)(hipEventDisableTiming | hipEventDisableSystemFence);
    }();
    return f;
}
void fork_event(hipEvent_t ev);                  // forks that follow wait for `ev` (the caller's stream's last record) instead of recording their own; nullptr ends it
hipStream_t aux_fork(hipStream_t from);          // nullptr when unavailable
int aux_join(hipStream_t to);
hipStream_t aux2_fork(hipStream_t from);         // the same on the decoder's prepare stream (idle during a backward pass); nullptr when unavailable
int aux2_join(hipStream_t to);
int aux2_publish();                             // instead of a join: echr_stream_join / the next library call wait for what the prepare stream carries now
bool helpers_available();
hipStream_t aux2_stream();
hipStream_t tail_stream_raw();                   // the tail stream itself, NOT ordered behind anything (echr_train_step's stage-ahead form); nullptr when unavailable
int prep_stream_wait(hipEvent_t ev);             // the prepare stream waits for `ev` (in addition to whatever its next fork waits for)
hipStream_t helpers_merge_to_tail();
int tail_publish();
int tsrm_position_early(const echr_tsrm_args* a, hipStream_t from);          // echr_train_step: start the event encoder's position branch right behind the index staging
void tsrm_bwd_defer_join(bool on);          // echr_train_step: the position branch's stream is joined by echr_stream_join, not inside echr_tsrm_bwd
int tsrm_bwd_parts(const echr_tsrm_args* a, const echr_tsrm_grads* g, const echr_dropout* drop, void* stream, int part);
int decoder_bwd_scratch_ahead(const echr_dec_args* a, const echr_dec_grads* g);
int decoder_bwd_parts(const echr_dec_args* a, const echr_dec_grads* g, const echr_dropout* drop, void* stream, int part);
void handover_close();                   // stop recording; the recorded events stay valid for echr_handover_wait
void handover_request(bool on, echr_handover_fn cb = nullptr, void* user = nullptr);          // the next decoder backward records the data-parallel hand-over events (decoder.hip)
int join_tail(hipStream_t st);          // make st wait for an asynchronous decoder-backward tail (decoder.hip); no-op when none is pending
int persist_read_stamps(unsigned long long* dst, int max_entries);
unsigned long long* persist_stamp_buffer(int S, hipStream_t st);      // diagnostic: [4][S <= 256][16] stamps, zeroed on st (nullptr: unavailable)
int gemm_grouped(const echr_gemm_desc* ds, int ng, hipStream_t st);
// C[M, Nc <= 16] = A[M, K] . W[Nc, K]^T + bias for very tall A (a stream over A; exact fp32 MFMA, plain stores)
bool gemm_skinny_ok(int M, int Nc, int K, long lda, long ldw, const float* A, const float* W);
int gemm_skinny_nt(const float* A, long lda, const float* W, long ldw, const float* bias, float* C, long ldc, int M, int Nc, int K, hipStream_t st);
// h2-packed operands (csrc/gemm.hip: two block-scaled fp16 planes): bytes of the packed image of a [rows x cols] operand (cols =
// contraction axis), the packing pass, and a multi-operand packing launch
struct H2PackJob { const float* src; unsigned char* dst; int R, K; long s_row, s_col;
                   const int* gather = nullptr; };      // optional: source row i (the strided axis: operand rows of a row pack, k of a transposing one) is read from row gather[i]
long h2_bytes(int rows, int cols);
int h2_pack(const float* src, int rows, int cols, long s_row, long s_col, void* dst, hipStream_t st);
int h2_pack_multi(const H2PackJob* jobs, int n, hipStream_t st);
inline long h2_floats(int rows, int cols) { return (h2_bytes(rows, cols) + 3) / 4; }
// packed operand of the rows x K matrix stored k-contiguous with leading dimension ld ...
inline H2PackJob pack_rows(const float* src, long ld, int rows, int K, float* dst) {
    return H2PackJob{src, reinterpret_cast<unsigned char*>(dst), rows, K, ld, 1};
}
// ... or stored transposed: element (r, k) at src[k * ld + r]
inline H2PackJob pack_cols(const float* src, long ld, int rows, int K, float* dst) {
    return H2PackJob{src, reinterpret_cast<unsigned char*>(dst), rows, K, 1, ld};
}


// C[M,N] = A[M,K] . W[N,K]^T  (nn.Linear forward; row-major operands with leading dimensions)
inline echr_gemm_desc desc_nt(const float* A, long lda, const float* W, long ldw, float* C, long ldc, int M, int N, int K) {
    echr_gemm_desc d{};
    d.A = A; d.B = W; d.C = C; d.M = M; d.N = N; d.K = K;
    d.sam = lda; d.sak = 1; d.sbk = 1; d.sbn = ldw; d.ldc = ldc;
    d.batch = 1; d.alpha = 1.f; d.beta = 0.f; d.split_k = 1;
    return d;
}
// C[M,N] = A[M,K] . B[K,N]   (data gradient dX = dY . W)
inline echr_gemm_desc desc_nn(const float* A, long lda, const float* B, long ldb, float* C, long ldc, int M, int N, int K) {
    echr_gemm_desc d = desc_nt(A, lda, B, 1, C, ldc, M, N, K);
    d.sbk = ldb; d.sbn = 1;
    return d;
}
// C[M,N] = A[K,M]^T . B[K,N] (weight gradient dW = dY^T . X)
inline echr_gemm_desc desc_tn(const float* A, long lda, const float* B, long ldb, float* C, long ldc, int M, int N, int K) {
    echr_gemm_desc d = desc_nn(A, 1, B, ldb, C, ldc, M, N, K);
    d.sam = 1; d.sak = lda;
    return d;
}

// C[M,N] = A . B^T on two h2-packed operands
inline echr_gemm_desc desc_h2(const float* Apk, const float* Bpk, float* C, long ldc, int M, int N, int K) {
    echr_gemm_desc d = desc_nt(Apk, K, Bpk, K, C, ldc, M, N, K);
    d.split_k = -1; d.algo = ECHR_GEMM_H2;
    return d;
}

// column sums: out[j] (+)= sum_i X[i*ld + j], i < rows
int colsum(const float* X, long ld, int rows, int cols, float* out, bool accumulate, hipStream_t st);
int colsum2(const float* X, long ld, int rows, int cols, float* out, float* out2, bool accumulate, hipStream_t st);
// several accumulating column sums in one launch (out / out2 / out3 += column sums of X; out2, out3 optional)
constexpr int COLSUM_MAX_JOBS = 8;
struct ColsumJob { const float* X; long ld; int rows, cols; float* out; float* out2; float* out3; };
int colsum_multi(const ColsumJob* jobs, int n, hipStream_t st);
// out[n*ld_out + j] = sum_t X[(t*N + n)*ld + j]
int sum_over_time(const float* X, long ld, int S, int N, int cols, float* out, long ld_out, hipStream_t st);
int fill_zero(float* p, long n, hipStream_t st);
int fill_zero_2d(float* p, int rows, int cols, long ld, hipStream_t st);
constexpr int FILL_MAX_JOBS = 8;
int fill_zero_multi(float* const* ptrs, const long* counts, int n, hipStream_t st);     // several ranges, one launch
constexpr int TRANSPOSE_MAX_JOBS = 8;
struct TransposeJob { const float* in; long ld_in; float* out; long ld_out; int rows, cols; };
int transpose_multi(const TransposeJob* jobs, int n, hipStream_t st);                    // several out[c,r] = in[r,c], one launch
// out[c, r] = in[r, c] for r < rows, c < cols; rows..rows_pad-1 of the output's row are zero-filled (k padding)
int transpose(const float* in, long ld_in, float* out, long ld_out, int rows, int cols, int rows_pad, hipStream_t st);
int embed_gather(const float* W, const int* tok, float* out, int rows, int E, int V1, hipStream_t st);
int row_matvec(const float* x, const float* W, long ldw, const float* b1, const float* b2, float* out, int N, int K, hipStream_t st);
int rank1_update(const float* x, const float* y, float* C, long ldc, int M, int N, bool accumulate, hipStream_t st);
int vec_mat(const float* x, const float* W, long ldw, float* out, int M, int N, hipStream_t st);
int embed_scatter_add(const float* dX, const int* tok, float* gW, int rows, int E, int V1, hipStream_t st, const int* rowmap = nullptr);
int logsoftmax_rows(float* X, long ld, int N, int S, int t0, int nt, int cols, hipStream_t st);
int logsoftmax_bwd(const float* logp, const float* G, const void* target, int tgt64, const float* mask, const float* g_loss,
                   const float* mask_sum, float* out, long ldo, int N, int S, int V1, hipStream_t st);
// log-softmax + masked NLL + d logits in one pass over the logits (echr_train_step); rows_sum turns the per-row terms into (loss, sum(mask))
bool logsoftmax_nll_dlg_ok(int V1, long ldo);
int logsoftmax_nll_dlg(const float* X, long ld, const void* target, int tgt64, const float* mask, const float* g_loss, float* out, long ldo,
                       float* row_loss, float* msum_out, int N, int S, int V1, hipStream_t st, const int* act = nullptr, int n_active = 0);
int nll_rows_sum(const float* row_loss, int NS, const float* msum, float* loss, hipStream_t st);
// echr_decoder_fwd with the criterion fused behind the logits product: g carries nll_target / nll_mask / g_loss / ws_bwd; d logits land in ws_bwd
// (echr_dec_grads.dlg_ready = 1 for the echr_decoder_bwd that follows); returns through *fused whether the fused form applied
int decoder_fwd_fused(const echr_dec_args* a, const echr_dec_grads* g, const echr_dropout* drop, void* stream, bool* fused, bool* compact);
int decoder_fused_loss(const echr_dec_args* a, const echr_dec_grads* g, float* loss, hipStream_t st);
int sample_step(const float* logits, long ld, int N, int V1, int t, int seq_len, int* it_next, int* unfinished, long long* seq,
                float* seq_logp, int* n_unfinished, float temperature, unsigned long long seed, hipStream_t st);
// slabs != nullptr: logits rows are formed here from four k-slice slabs (+ bias) in a fixed order and written to `logits`
int greedy_step(float* logits, long ld, int N, int V1, int t, int seq_len, int* it_next, int* unfinished,
                long long* seq, float* seq_logp, int* n_unfinished, hipStream_t st, const float* slabs = nullptr, long slab_stride = 0,
                const float* bias = nullptr, int nslab = 0);

}  // namespace echr
