/*
 * echr_hip.h -- C ABI of libechr_hip.so: the MI355X (gfx950) implementation of ECHR's
 * hierarchical-encoder + attention caption-decoder hot path.
 *
 * The reference (ttengwang/ECHR) has no FFI layer of its own: its boundary is the Python module API
 * (CaptionGenerator.py:17, models/__init__.py:6-29).  echr_amd/ keeps that Python API and calls the
 * entry points below through ctypes; each entry point names the reference code it replaces.
 *
 * Conventions
 *  - every pointer is a BORROWED device pointer (PyTorch owns all memory; the library never
 *    allocates or frees user-visible buffers); tensors are row-major contiguous fp32 unless a
 *    leading dimension is given; index tensors are int32.
 *  - `stream` is a hipStream_t passed as void*; calls are asynchronous w.r.t. the host, re-entrant,
 *    never synchronise, and are legal inside hipGraph stream capture.
 *  - return 0 on success, negative errno-style code on failure; echr_last_error() returns the
 *    calling thread's last message.
 */
#ifndef ECHR_HIP_H
#define ECHR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2 (round 5): echr_train_step_args.handover, echr_handover_wait, echr_async_skipped_updates; round 4 had grown the decoder's
 * argument struct (train, zero_extra, zero_extra_count) and its gradient struct (dlg_ready, active_rows, n_active) under version 1.  A binding that restates
 * the structs MUST also compare its sizeof() of each with echr_abi_sizeof(): the version alone does not describe the layouts. */
/* 3 (round 6): echr_clamp_adam_counted, echr_train_step_args.adam_applied (a per-optimiser count of APPLIED updates; the process-wide
 * echr_async_skipped_updates now counts echr_clamp_adam launches only); echr_train_step_args.event_parts / w_init.. / vh_offset: the
 * reference's 'ER1' / 'ER2' event contexts, CG_init_feats_type and the 'VH' scene context's gradient on the one-call path. */
#define ECHR_ABI_VERSION 3

int echr_version(void);
/* sizeof() of an argument struct of this header by its type name ("echr_dec_args", ...), -1 for an unknown name: lets a binding that
 * restates the structs (ctypes, cgo, JNI) check its layout against the library it loaded */
int64_t echr_abi_sizeof(const char* name);
const char* echr_last_error(void);
/* Asynchronous failures.  The persistent recurrence kernels (csrc/persist.hip) bound every inter-workgroup wait; when one gives up the
 * launch drains, a device word stays set (echr_clamp_adam / echr_clamp then skip their update, so no parameter is touched by the
 * invalid gradients) and the NEXT library call that takes a stream returns -ETIME (-62) once, naming the edge and timestep.
 * echr_check_async() is that check on its own, for callers that want it right after a synchronisation point: 0 or -62. */
int echr_check_async(void);
/* Number of echr_clamp_adam launches (of ANY optimiser state of the process; echr_clamp is not an update and is not counted) that skipped
 * their update because of the asynchronous failure the last -62 reported (read once: the count is cleared).  Process-wide, therefore only
 * a diagnostic when several optimiser states step in one process: a caller that keeps step counts winds each of them back from its OWN
 * count of applied updates (echr_clamp_adam_counted / echr_train_step_args.adam_applied). */
int64_t echr_async_skipped_updates(void);

/* ------------------------------------------------------------------------------------------------
 * Dense fp32 projection on the matrix cores (v_mfma_f32_32x32x2_f32: exact fp32, k-ordered fma chain).
 * Replaces every nn.Linear / torch.bmm / addmm site of the path.
 *   C[b][rowmap(i)][j] = act( alpha * sum_k A[b](i,k) * B[b](k,j) + beta * C_old + bias[j] + bias2[j]
 *                             + addend[(i % add_mod)][j] )
 * A(i,k) = A[i*sam + k*sak], B(k,j) = B[k*sbk + j*sbn]; in each pair one stride must be 1.
 * split_k > 1: partial sums are atomically added into C (which must already hold its base value);
 *              act must be ECHR_ACT_NONE and beta is ignored.
 * ---------------------------------------------------------------------------------------------- */
enum { ECHR_GEMM_F32 = 0, ECHR_GEMM_BF16X3 = 1, ECHR_GEMM_H2 = 2 };
enum { ECHR_ACT_NONE = 0, ECHR_ACT_TANH = 1, ECHR_ACT_MUL_DTANH = 2 /* acc * (1 - aux^2) */ };

typedef struct {
    const float* A;
    const float* B;
    float* C;
    int32_t M, N, K;
    int64_t sam, sak, sbk, sbn, ldc;
    int32_t batch;
    int64_t bsa, bsb, bsc;
    float alpha, beta;
    const float* bias;
    int64_t bs_bias;
    const float* bias2;
    const float* addend;
    int32_t add_mod;
    int64_t ld_add;
    int32_t act;
    const float* aux;
    int64_t ld_aux;
    int32_t rowmap_mod, rowmap_mul; /* out row = (i % mod) * mul + i / mod ; mod = 0 -> identity */
    int32_t split_k;
    int32_t algo; /* ECHR_GEMM_F32 (exact v_mfma_f32_32x32x2_f32) or ECHR_GEMM_BF16X3 (fp32 operands split exactly into three
                     bf16 planes, six plane products on v_mfma_f32_32x32x16_bf16, fp32 accumulate: fp32-accurate, 2.7x the rate;
                     needs both operands k-contiguous and 16-byte aligned, else the library silently uses ECHR_GEMM_F32),
                     or ECHR_GEMM_H2: A and B point at operands that echr_h2_pack has already rewritten as two block-scaled fp16
                     planes (below); three fp16 MFMA products per k block, fp32-grade accuracy at 5x the fp32 MFMA rate and
                     fp32's byte count (strides are ignored; batch = 1, no activation) */
    const int32_t* row_index; /* optional output-row scatter: row i of the product is ADDED (fp32 atomics; several rows may share a target) into
                                 C[clamp(row_index[i], 0, row_index_max)] -- the token-embedding gradient d E[tok] += d X[row] without a
                                 materialised d X and a scatter pass (OldModel_NEW.py:110 backward).  Needs beta = 1, act = NONE, no rowmap */
    int32_t row_index_max;
} echr_gemm_desc;

int echr_gemm_f32(const echr_gemm_desc* d, void* stream);

/* h2 operands: a logical [rows x cols] fp32 operand (cols = the contraction axis) rewritten as two fp16 planes with one shared
 * power-of-two scale per row and 256-wide k segment (8 blocks of 32): xs = x * 2^(14 - floor(log2 blockmax)), h1 = fp16(xs), h2 = fp16(xs - h1).
 * No value is held in fp16 unscaled, so the fp32 exponent range survives; h1 + h2 carries 22..24 significand bits of every
 * element within 2^-17 of its segment maximum.  Laid out as zero-padded 128 x 32 chunks in the order the GEMM's direct-to-LDS
 * loads consume.  Element (r, k) is read from src[r*s_row + k*s_col] (one stride must be 1, so a transposed source packs
 * without a separate transpose).  dst: echr_h2_bytes(rows, cols) bytes, 16-byte aligned. */
int64_t echr_h2_bytes(int32_t rows, int32_t cols);
int echr_h2_pack(const float* src, int32_t rows, int32_t cols, int64_t s_row, int64_t s_col, void* dst, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Dropout configuration shared by all training-mode entry points (counter-based Philox-4x32-10;
 * bit-identical host implementation: echr_amd/philox.py).  Sites: reference nn.Dropout instances at
 * models/MA_attention_8_NEW.py:162 (p=0.3), models/OldModel_NEW.py:810,814,818 (p=0.5), :136 (CG_drop_prob).
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    uint64_t seed;
    uint32_t offset;   /* forward-call counter */
    int32_t training;  /* 0: every dropout is the identity */
    float p_tsrm, p_h, p_out;
} echr_dropout;

/* ------------------------------------------------------------------------------------------------
 * Event-level context inputs: mean-pool of each event's C3D rows + gather of the proposal-LSTM
 * state at the event's anchor.  Replaces CaptionGenerator.py:111-114,121,128.
 *   ech[n, 0:D]     = mean(c3d[ev_start[n] : ev_start[n]+ev_len[n], :])
 *   ech[n, D:D+Ht]  = tap[ind[n], :]
 * bwd scatters d_ech[:, D:] into d_tap rows (atomic add; d_tap pre-zeroed) -- gradients to c3d are
 * not produced (features are data).
 * ---------------------------------------------------------------------------------------------- */
int echr_event_pool_gather_fwd(const float* c3d, const float* tap, const int32_t* ev_start, const int32_t* ev_len,
                               const int32_t* ind, float* ech, int32_t N, int32_t D, int32_t Ht, void* stream);
int echr_event_pool_gather_bwd(const float* d_ech, const int32_t* ind, float* d_tap, int32_t N, int32_t D, int32_t Ht,
                               void* stream);
/* Non-zero initial decoder state (OldModel.init_hidden, OldModel_NEW.py:72-96; CG_init_feats_type from 'V', 'E', 'C'):
 *   feats = cat([video broadcast over the N events | event | clip.mean(1)], 1);   h0 = init_linear(feats)   [N, 3H]
 * `clip.mean(1)` is the mean over the A PADDED slots of CaptionGenerator.get_clip_context (zeros behind an event's end): sum over the event's
 * rows / A.  bwd: gradients of init_linear (written, or accumulated when `zeroed`), d video (written), d event (ADDED to g_event); c3d is data. */
typedef struct {
    int32_t N, Dv, De, D, H3;                  /* events; widths of the scene / event context, of a C3D row; 3 * rnn_size */
    int32_t use_v, use_e, use_c;               /* 'V' / 'E' / 'C' in CG_init_feats_type */
    int32_t A;                                 /* padded clip length (max event length of the batch) */
    const float *video, *event, *c3d;          /* [Dv], [N,De], [Tv,D] */
    const int32_t *ev_start, *ev_len;
    const float *w, *b;                        /* init_linear.weight [H3, Dtot], .bias [H3] */
    float* feats;                              /* [N, Dtot] scratch; the backward pass reads it again */
    float* h0;                                 /* out [N, H3] */
} echr_init_state_args;
typedef struct {
    const float* g_h0;                         /* [N, H3] */
    float *g_w, *g_b;                          /* [H3, Dtot], [H3] */
    float *g_video, *g_event;                  /* [Dv] written (NULL: not wanted), [N,De] added into (NULL: not wanted) */
    float* dfeats;                             /* [N, Dtot] scratch */
    int32_t zeroed;                            /* 1: g_w / g_b are pre-zeroed accumulation targets */
} echr_init_state_grads;
int echr_init_state_fwd(const echr_init_state_args* a, void* stream);
int echr_init_state_bwd(const echr_init_state_args* a, const echr_init_state_grads* g, void* stream);

/* Scene context 'VC' / 'VH' (CaptionGenerator.get_video_context, CaptionGenerator.py:95-99: `c3d_feats.mean(0)`, `tap_feats.mean(0)`):
 * out[c] = mean over the `rows` rows of x[:, c] (x row-major with leading dimension ld); bwd: gx[r, c] += g[c] / rows. */
int echr_col_mean_fwd(const float* x, int32_t rows, int32_t cols, int64_t ld, float* out, void* stream);
int echr_col_mean_bwd(const float* g, int32_t rows, int32_t cols, int64_t ld, float* gx, void* stream);

/* ------------------------------------------------------------------------------------------------
 * TSRM event-relation encoder.  Replaces MA_Attention8.forward (MA_attention_8_NEW.py:35-49) and
 * attention_module_multi_head.forward (:101-177), fST0 / use_posit=1.
 * Parameter pointers use the reference state_dict names (fusion_model.*).
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    int32_t N, Din, Df, Do, G;
    /* parameters */
    const float *w_emb, *b_emb;      /* event_emb      [Df,Din],[Df] */
    const float *w_fc1, *b_fc1;      /* pair_pos_fc1   [Df,Df],[Df]  */
    const float *w_fc2, *b_fc2;      /* pair_pos_fc2   [G,Df],[G]    */
    const float *w_q, *b_q;          /* query_1        [Df,Df],[Df]  */
    const float *w_k, *b_k;          /* key_1          [Df,Df],[Df]  */
    const float *w_out, *b_out;      /* linear_out_1   [Do,Df(,1,1)],[Do] */
    /* inputs */
    const float* ech;                /* [N,Din] */
    const int32_t *ev_start, *ev_len;
    /* saved activations (caller-allocated; sizes via echr_tsrm_ws_floats) */
    float* ws;
    /* output */
    float* out;                      /* [N,Do] */
    int32_t inference;               /* echr_tsrm_fwd: 1 = no echr_tsrm_bwd will follow on this workspace (activations that only the backward
                                        pass reads -- the fp32 position embedding of >= 4096 pairs -- are not materialised) */
    int32_t max_len, max_span;       /* optional, known to the host from the index lists: max ev_len and max |(2 start_i + len_i) - (2 start_j +
                                        len_j)| over the events (0 / -1... leave max_len = 0 when unknown).  With `inference` and >= 16384 pairs
                                        they let the pair MLP run from tables over the distinct (|2 dc|, l_i) and (l_i, l_j) keys */
    int32_t fst_mode;                /* how position gate and scaled affinity combine ahead of the softmax (MA_attention_8_NEW.py:148-157):
                                        0 = fST0 gate * aff (the recipe), 1 = fST1 gate + aff, 2 = fST2 log(clamp(gate, 1e-6)) + aff,
                                        3 = fST3 the gate alone, 4 = use_posit = 0: the affinity alone (the position branch is not run and
                                        pair_pos_fc1 / fc2 receive no gradient) */
} echr_tsrm_args;

typedef struct {
    /* parameter gradients (written, not accumulated, unless `zeroed`) */
    float *g_w_emb, *g_b_emb, *g_w_fc1, *g_b_fc1, *g_w_fc2, *g_b_fc2, *g_w_q, *g_b_q, *g_w_k, *g_b_k, *g_w_out, *g_b_out;
    float* g_ech;                    /* [N,Din] */
    const float* g_out;              /* [N,Do] */
    float* ws_bwd;                   /* scratch, echr_tsrm_ws_bwd_floats */
    int32_t zeroed;                  /* 1: the caller zero-filled every parameter-gradient buffer (flat arena): the
                                        library accumulates into them and skips its own zero fills */
} echr_tsrm_grads;

int64_t echr_tsrm_ws_floats(int32_t N, int32_t Din, int32_t Df, int32_t Do, int32_t G);
int64_t echr_tsrm_ws_bwd_floats(int32_t N, int32_t Din, int32_t Df, int32_t Do, int32_t G);
int echr_tsrm_fwd(const echr_tsrm_args* a, const echr_dropout* drop, void* stream);
int echr_tsrm_bwd(const echr_tsrm_args* a, const echr_tsrm_grads* g, const echr_dropout* drop, void* stream);
/* attention_module_multi_head.forward on its own (MA_attention_8_NEW.py:101-177, fST0, use_posit = 1): roi_feat [N,Df] = the embedded
 * events, pos_emb [N,N,Df] = the pairwise position embedding; a->ech / ev_start / ev_len / w_emb / b_emb are not read (Din only
 * sizes the workspace).  Forward only. */
int echr_tsrm_attn_fwd(const echr_tsrm_args* a, const float* roi_feat, const float* pos_emb, const echr_dropout* drop, void* stream);
/* position embedding alone (float64 math on device, fp32 result [N,N,Df]); replaces the numpy
 * extract_position_matrix / extract_position_embedding (:51-79) + the host->device copy at :41 */
int echr_tsrm_posemb(const int32_t* ev_start, const int32_t* ev_len, float* pos, int32_t N, int32_t Df, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Three-stream attention caption decoder.  Replaces OldModel.forward (teacher forcing,
 * models/OldModel_NEW.py:98-130), get_logprobs_state (:133-137), ThreeStream_Core.forward (:801-823),
 * Attention.forward (:376-401), the greedy branch of OldModel.sample (:139-187), and, optionally
 * fused, LanguageModelCriterion (misc/utils.py:66-75).
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    int32_t N, A, Tv, D, H, E, Ha, De, Dv, V1, S;
    int32_t rows_disjoint;                     /* 1: no two events share a video row (lets d P_all be stored instead of atomically added) */
    /* parameters (reference state_dict names under lm_model.) */
    const float* embed;                        /* embed.weight [V1,E] */
    const float *w_logit, *b_logit;            /* logit [V1,3H],[V1] */
    const float* w_ih[3];                      /* core.layer{k}.weight_ih [4H, E+ctx_k] */
    const float* w_hh[3];                      /* [4H,H] */
    const float* b_ih[3];
    const float* b_hh[3];
    const float *w_c2a, *b_c2a;                /* core.attention.ctx2att [Ha,D],[Ha] */
    const float *w_h2a, *b_h2a;                /* h2att [Ha,H],[Ha] */
    const float *w_alpha, *b_alpha;            /* alpha_net [1,Ha],[1] */
    /* inputs */
    const float* c3d;                          /* [Tv,D] */
    const int32_t *ev_start, *ev_len;          /* [N] (clip context = rows ev_start..ev_start+ev_len) */
    const float* event;                        /* [N,De] */
    const float* video;                        /* [Dv]   */
    const int32_t* tokens;                     /* [S,N] time-major input tokens (labels[:, t]) */
    /* workspace (saved for backward), echr_decoder_ws_floats */
    float* ws;
    /* output: log-probs [N,S,V1] */
    float* logp;
    int32_t prepared;                          /* echr_decoder_fwd only: 1 = echr_decoder_fwd_prepare already ran on this workspace */
    int32_t train;                             /* 1 = echr_decoder_bwd will follow on this workspace: what only the backward pass reads and depends on
                                                  parameters alone (the transposed h2 image of W_logit for d OUTD = d logits . W_logit) is packed by the
                                                  forward's own packing launch, off the critical path in front of the reverse recurrence.  Pass the
                                                  SAME value to echr_decoder_fwd_prepare / echr_decoder_fwd / echr_decoder_bwd.  0: the backward packs it */
    float* zero_extra;                         /* echr_decoder_fwd / _prepare, optional: a range the caller wants zero-filled before the backward pass (its
                                                  gradient arena), folded into the forward's first fill launch; NULL = none */
    int64_t zero_extra_count;                  /* floats */
    const float* h0;                           /* optional [N, 3H]: the initial state, `init_linear(cat(...))` BEFORE its view / transpose
                                                  (OldModel_NEW.py:72-96, CG_init_feats_type non-empty): stream k starts from h = c = h0[:, kH:(k+1)H].
                                                  NULL = the recipe's zero state.  With it the recurrences run launch per phase (the persistent
                                                  kernels assume h(-1) = 0) */
} echr_dec_args;

typedef struct {
    float* g_embed;                            /* [V1,E] must be zeroed by the caller (scatter-add) */
    float *g_w_logit, *g_b_logit;
    float* g_w_ih[3];
    float* g_w_hh[3];
    float* g_b_ih[3];
    float* g_b_hh[3];
    float *g_w_c2a, *g_b_c2a, *g_w_h2a, *g_b_h2a, *g_w_alpha, *g_b_alpha;
    float* g_event;                            /* [N,De] */
    float* g_video;                            /* [Dv] or NULL */
    const float* g_logp;                       /* [N,S,V1] upstream gradient, or NULL when nll_* are set */
    /* fused criterion path: d loss / d logp = -mask/(sum(mask)+1e-6) * g_loss at the target entries */
    const int32_t* nll_target;                 /* [N,S] or NULL */
    const float* nll_mask;                     /* [N,S] */
    const float* g_loss;                       /* device scalar */
    float* ws_bwd;                             /* scratch, echr_decoder_ws_bwd_floats */
    int32_t zeroed;                            /* 1: parameter-gradient buffers arrive zero-filled (see echr_tsrm_grads) */
    int32_t phase;                             /* 0: the whole backward.  A data-parallel caller may run it in stages ON THE SAME ws_bwd and
                                                  start reducing gradients as they become final:
                                                    1 = late-fusion stage (d logits; g_w_logit, g_b_logit final: 35 % of the bytes),
                                                    3 = reverse recurrence + every gradient of the three LSTM layers (g_w_ih, g_w_hh,
                                                        g_b_ih, g_b_hh, plus g_w_h2a, g_b_h2a, g_event, g_video): 39 % of the bytes,
                                                    4 = the rest (attention parameters, token embedding);
                                                    2 = 3 followed by 4.   1, 3, 4 in this order equal 0. */
    int32_t async_tail;                        /* phase 0 only: 1 = the last stage (attention-parameter and token-embedding gradients; nothing else
                                                  in a backward pass depends on them) runs on a library-owned second stream and the call returns
                                                  without joining it, so that the caller's next backward kernels (event encoder, proposal encoder)
                                                  overlap it; the logit-layer gradients are formed there too.  The caller MUST call
                                                  echr_stream_join(stream) before anything reads g_w_logit, g_b_logit, g_w_c2a,
                                                  g_b_c2a, g_w_alpha, g_b_alpha, g_embed or frees ws / ws_bwd (every library entry that takes a
                                                  stream joins first as a safety net).
                                                  2 = additionally only g_event (and what it needs) is formed on `stream`; every other gradient
                                                  of the three LSTM layers (g_w_ih, g_w_hh, g_b_ih, g_b_hh, g_w_h2a, g_b_h2a, g_video) is formed on
                                                  a second library-owned stream and is final only after echr_stream_join as well (needs zeroed = 1) */
    const float* nll_msum;                     /* fused criterion path, optional: device pointer to sum(nll_mask) (the second output of
                                                  echr_nll_loss_fwd); NULL = the library sums the mask itself */
    float* zero_extra;                         /* optional: a range the caller wants zero-filled before any gradient is written (its gradient
                                                  arena span when `zeroed` = 1): folded into the backward's own multi-range fill launch
                                                  (stage 1) instead of a separate fill; NULL = none */
    int64_t zero_extra_count;                  /* floats */
    int32_t nll_target_i64;                    /* 1: nll_target points at int64 indices (the reference's LongTensor labels as they are: no
                                                  conversion pass), 0: int32 */
    int32_t dlg_ready;                         /* 1: d logits already sit in ws_bwd (echr_train_step forms them in the pass that reads the logits:
                                                  log-softmax + criterion + its gradient in one kernel); callers of echr_decoder_bwd leave it 0 */
    const int32_t* active_rows;                /* echr_train_step only (with dlg_ready): the n_active time-major rows t*N + n with t <= the last
                                                  timestep at which caption n's criterion mask is non-zero, ascending.  Rows outside carry exactly zero
                                                  d logits (misc/utils.py:66-75: the mask multiplies the log-prob) and, as nothing later in the caption
                                                  feeds back into them, exactly zero d gates / d q, so the late-fusion products run on the compacted
                                                  rows (logits and d logits exist as [n_active, V1], d OUTD is scattered back by row, d W_logit contracts
                                                  over n_active rows) and the recurrent weight gradients and d XT contract / run over the same rows */
    int32_t n_active;
    float* g_h0;                               /* optional [N, 3H] out (written): d loss / d h0 = d h(-1) + d c(-1) of every stream (echr_dec_args.h0) */
} echr_dec_grads;

/* make `stream` wait for an asynchronous decoder-backward tail (echr_dec_grads.async_tail); no-op when none is pending */
int echr_stream_join(void* stream);
/* Create the library's helper streams now (they are otherwise created by the first call that needs them) and submit one marker on each.
 * Call it once per process AFTER selecting the device and BEFORE anything else creates streams -- in particular before the RCCL communicator
 * is set up (torch.distributed.init_process_group with device_id): this runtime spreads streams over a handful of hardware queues in creation
 * order, and helper streams created behind RCCL's land on queues they share with each other -- the backward tail's three streams then
 * serialise (measured on one MI355X, single-rank communicator: 1.64 vs 1.49 ms per c3 iteration).  Returns 0, or -ENODEV without a device. */
int echr_streams_init(void);

int64_t echr_decoder_ws_floats(const echr_dec_args* a);
int64_t echr_decoder_ws_bwd_floats(const echr_dec_args* a);
int echr_decoder_fwd(const echr_dec_args* a, const echr_dropout* drop, void* stream);
/* The part of echr_decoder_fwd that does not read the event context (operand packs, P_all = ctx2att over the video rows, token
 * embedding, the token-side gate pre-activations of all S*N rows): started on a library-owned second stream, forked from `stream`, so
 * that it overlaps the event encoder (echr_event_pool_gather_fwd + echr_tsrm_fwd) the caller runs next on `stream`.  a->event may be
 * NULL here.  The following echr_decoder_fwd call must pass the same workspace with a->prepared = 1; it joins the streams. */
int echr_decoder_fwd_prepare(const echr_dec_args* a, void* stream);
/* Abandon a prepare whose echr_decoder_fwd will not follow (the caller changed its mind, or failed in between): `stream` waits for the
 * second stream, so the workspace may be released or reused in `stream` order afterwards.  A no-op when nothing is pending. */
int echr_decoder_fwd_prepare_cancel(void* stream);
int echr_decoder_bwd(const echr_dec_args* a, const echr_dec_grads* g, const echr_dropout* drop, void* stream);

/* ONE decoder timestep with the recurrent state passed in and out: OldModel.get_logprobs_state (models/OldModel_NEW.py:133-137) =
 * embed -> ThreeStream_Core.forward (:801-823, incl. Attention.forward :376-401) -> dropout -> logit -> log_softmax.
 * a: S is ignored (treated as 1), tokens = it [N] int32, logp = out [N,V1], ws from echr_decoder_ws_floats with S = 1.
 * h_in / c_in / h_out / c_out: [3,N,H] (the reference's `state` tuple; h holds the DROPPED outputs, :810,814,818,821).
 * drop->offset selects the dropout stream of this call (training mode).  Forward only; bitwise reproducible. */
int echr_decoder_step(const echr_dec_args* a, const float* h_in, const float* c_in, float* h_out, float* c_out,
                      const echr_dropout* drop, void* stream);

/* masked NLL of LanguageModelCriterion on log-probs [N,S,V1]: loss (device scalar) */
int echr_nll_loss_fwd(const float* logp, const int32_t* target, const float* mask, float* loss, int32_t N, int32_t S,
                      int32_t V1, void* stream);
/* the same on int64 targets (torch.LongTensor labels as train.py:298-303 passes them) */
int echr_nll_loss_fwd_i64(const float* logp, const int64_t* target, const float* mask, float* loss, int32_t N, int32_t S,
                          int32_t V1, void* stream);

/* backward of echr_nll_loss_fwd in one pass: g_logp [N,S,V1] (fully written); fwd_out = the 2-float output of the forward
 * (loss, sum(mask)), g_loss = upstream scalar gradient (device) */
int echr_nll_loss_bwd(const int32_t* target, const float* mask, const float* fwd_out, const float* g_loss, float* g_logp,
                      int32_t N, int32_t S, int32_t V1, void* stream);
int echr_nll_loss_bwd_i64(const int64_t* target, const float* mask, const float* fwd_out, const float* g_loss, float* g_logp,
                          int32_t N, int32_t S, int32_t V1, void* stream);

/* sampler (greedy or multinomial): runs seq_len+1 decoder steps on device without host syncs.
 * seq [N,seq_len] int64 (zero after a row finished), seq_logp [N,seq_len] fp32,
 * n_unfinished [seq_len+1] int32: number of unfinished rows after step t (the host trims the output
 * at the first t >= 1 with n_unfinished[t] == 0, as OldModel.sample's break does). */
typedef struct {
    echr_dec_args dec;       /* S, tokens, logp unused */
    int32_t seq_len;
    int64_t* seq;
    float* seq_logp;
    int32_t* n_unfinished;
    float* ws_sample;        /* echr_sampler_ws_floats */
    int32_t multinomial;     /* 0 = greedy arg-max (sample_max = 1, lowest index on ties); 1 = draw from softmax(logp / temperature)
                                (OldModel_NEW.py:160-168, sample_max = 0) with the library's Philox stream keyed by (seed, row, step):
                                reproducible per seed, the reference's distribution but not torch.multinomial's random stream */
    float temperature;       /* multinomial only; <= 0 means 1 */
    uint64_t seed;
    float* tables;           /* optional cache of the parameter-only operands of the persistent greedy decoder (token-side gate tables
                                embed . W_ih_k[:, :E]^T, logit-weight image): echr_sampler_table_floats floats, owned by the caller.  NULL:
                                they are rebuilt inside ws_sample on every call (~0.15 ms at V1 = 5001) */
    int32_t tables_valid;    /* 1: `tables` holds the operands of the CURRENT parameters (skip the rebuild); 0: rebuild into `tables` */
} echr_sample_args;
int64_t echr_sampler_ws_floats(const echr_dec_args* a);
int64_t echr_sampler_table_floats(const echr_dec_args* a);      /* 0 when the persistent decoder does not apply to these shapes / settings */
int echr_decoder_sample(const echr_sample_args* a, void* stream);

/* ------------------------------------------------------------------------------------------------
 * SST proposal encoder.  Replaces models/sst_model.py:31-40 (nn.LSTM over one video + Linear + sigmoid)
 * and TAPModelCriterion (misc/utils.py:78-99).  Parameters use nn.LSTM's layout: w_ih[l] [4H, D or H], w_hh[l] [4H,H],
 * gate order i,f,g,o.  p_drop = inter-layer dropout (applied to layer 0's output while training; site 5 of echr_dropout).
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    int32_t T, D, H, K;
    float p_drop;
    const float* w_ih[2];
    const float* w_hh[2];
    const float* b_ih[2];
    const float* b_hh[2];
    const float *w_sc, *b_sc;       /* scores Linear [K,H],[K] */
    const float* x;                 /* [T,D] C3D features of one video */
    float* ws;                      /* saved activations, echr_sst_ws_floats */
    float* tap_feats;               /* out [T,H]: top-layer hidden states */
    float* scores;                  /* out [T,K]: sigmoid proposal scores */
} echr_sst_args;

typedef struct {
    float* g_w_ih[2];
    float* g_w_hh[2];
    float* g_b_ih[2];
    float* g_b_hh[2];
    float *g_w_sc, *g_b_sc;
    const float* g_tap;             /* [T,H] or NULL */
    const float* g_scores;          /* [T,K] or NULL */
    float* ws_bwd;                  /* scratch, echr_sst_ws_bwd_floats */
    int32_t zeroed;                 /* 1: every parameter-gradient buffer arrives zero-filled (flat arena, one fill by the caller): the library
                                       accumulates into them instead of zero-filling each split-K product's output itself */
} echr_sst_grads;

int64_t echr_sst_ws_floats(int32_t T, int32_t D, int32_t H, int32_t K);
int64_t echr_sst_ws_bwd_floats(int32_t T, int32_t D, int32_t H, int32_t K);
int echr_sst_fwd(const echr_sst_args* a, const echr_dropout* drop, void* stream);
/* The two halves of echr_sst_fwd as calls of their own (round 6): the recurrence -> tap_feats (models/sst_model.py:33-37), and the proposal head
 * scores = sigmoid(tap_feats . W_sc^T + b_sc) (:38-39) from the tap_feats of the same arguments.  The joint iteration's caption side waits for
 * tap_feats alone; its caller queues the head behind the caption call (fused.JointTrainStep). */
int echr_sst_fwd_states(const echr_sst_args* a, const echr_dropout* drop, void* stream);
int echr_sst_head_fwd(const echr_sst_args* a, void* stream);
int echr_sst_bwd(const echr_sst_args* a, const echr_sst_grads* g, const echr_dropout* drop, void* stream);
/* weighted BCE of the proposal head: loss (device scalar) and its gradient w.r.t. the scores */
int echr_tap_bce_fwd(const float* scores, const float* masks, const float* labels, const float* w1, float* loss, int32_t T,
                     int32_t K, void* stream);
/* the same loss with a caller-provided scratch of 64 floats (`partials`): 64 workgroups + a final fixed-order add instead of one workgroup
 * (bit-reproducible either way; what echr_amd uses) */
int echr_tap_bce_fwd_ws(const float* scores, const float* masks, const float* labels, const float* w1, float* loss, float* partials,
                        int32_t T, int32_t K, void* stream);
int echr_tap_bce_bwd(const float* scores, const float* masks, const float* labels, const float* w1, const float* g_loss,
                     float* g_scores, int32_t T, int32_t K, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Proposal selection: the index outputs of eval_utils.gettop1000 (eval_utils.py:259-287), bit-exact.
 * Output buffers must hold T*K entries (ties at the threshold are all kept); out_count[0] = number written.
 * ---------------------------------------------------------------------------------------------- */
int echr_top_proposals(const float* scores, const float* mask, int32_t T, int32_t K, int32_t topN, float val_thres,
                       int32_t* out_ind, int32_t* out_feat, float* out_conf, int32_t* out_count, void* stream);

/* Greedy temporal NMS variant (eval_utils.gettop1000_nms, eval_utils.py:290-331): candidates (n, k < min(n,K)) = [n-k, n+1], picked by
 * descending score until topN, suppressing inclusive-IoU > overlap (float64, the reference's operation order).  Equal scores: the
 * later candidate wins (the reference's unstable argsort leaves ties undefined).  scratch: T*K floats.  out_feat [topN,2], out_conf
 * [topN] in pick order; out_count[0] = number of picks. */
int echr_top_proposals_nms(const float* scores, int32_t T, int32_t K, int32_t topN, double overlap, float* scratch,
                           int32_t* out_feat, float* out_conf, int32_t* out_count, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Fused element-wise clamp(+-clip) + Adam (betas, eps, no weight decay, no amsgrad) over a flat
 * buffer.  Replaces misc/utils.py:107-111 + torch.optim.Adam.step as wired at train.py:209,315-317.
 * `step` is the 1-based step count; lr/betas/eps are doubles because torch derives 1-beta and the bias corrections
 * from the Python doubles (1 - 0.999f != 0.001).
 * ---------------------------------------------------------------------------------------------- */
int echr_clamp_adam(float* p, const float* g, float* m, float* v, int64_t n, int32_t step, double lr, double beta1,
                    double beta2, double eps, float clip, void* stream);
/* The same, counting itself: `applied` (optional device word, one per optimiser state, zeroed by its owner) is incremented by the launch
 * iff the update was applied, i.e. not skipped under an unacknowledged asynchronous failure (see echr_check_async).  After a -62 the owner
 * reads the word (everything queued has retired by then) and sets its step count to the number of updates that really happened. */
int echr_clamp_adam_counted(float* p, const float* g, float* m, float* v, int64_t n, int32_t step, double lr, double beta1,
                            double beta2, double eps, float clip, uint32_t* applied, void* stream);

/* ------------------------------------------------------------------------------------------------
 * One training iteration of the caption path as ONE call.  Replaces the per-iteration protocol of train.py:281-317 around the hot
 * path -- zero_grad; cg_model(tap_feats, c3d_feats, lda_feats, labels, ind, soi, 'train') (CaptionGenerator.py:17-30);
 * LanguageModelCriterion (misc/utils.py:66-75); backward; clip_gradient; optimizer.step() -- for m_batch = 1 with the caption
 * model's parameters and gradients in flat buffers.  The entry sequences the library's own pieces (event pooling, TSRM encoder,
 * decoder forward / backward with the fused criterion gradient, clamp + Adam) on `stream`; the host pays one call instead of ~75.
 *
 * The caller fills: the parameter pointers and shapes of `tsrm` / `dec` (N, A, Tv, S, rows_disjoint included), dec.c3d, dec.video,
 * the gradient pointers of `tsrm_g` / `dec_g` (views of flat_g; dec_g.g_video optional), `drop`, and the fields below.  Every
 * other pointer of the four embedded structs (index vectors, event context, tokens, workspaces, log-probs, upstream gradients) is
 * set by the library from `ws`.
 * ---------------------------------------------------------------------------------------------- */
/* hand-over callback (echr_train_step_args.handover_cb); which = ECHR_HANDOVER_LOGIT or ECHR_HANDOVER_LSTM */
typedef void (*echr_handover_fn)(int32_t which, void* stream, void* user);
/* mid-call hook of the joint form (echr_train_step_args.mid_cb) */
typedef void (*echr_mid_fn)(void* stream, void* user);
typedef struct {
    echr_tsrm_args tsrm;
    echr_tsrm_grads tsrm_g;
    echr_dec_args dec;
    echr_dec_grads dec_g;
    echr_dropout drop;
    const float* tap;              /* [Tv,Ht] proposal-encoder states (tap_feats) */
    int32_t Ht;
    float* g_tap;                  /* optional [Tv,Ht], zero-filled by the caller: receives d loss / d tap_feats (joint training); NULL = tap is data */
    const int32_t* host_index;     /* HOST memory (any kind; copied into a pinned ring inside the call):
                                      ev_start[N] | ev_len[N] | ind[N] | tokens[S*N] (time-major input tokens, labels[:, t])
                                      | active[n_active] (below) | and, with host_nll = 1, targets int32 [N*S] | mask fp32 [N*S] */
    const void* nll_target;        /* device [N,S] targets (labels[:, 1:1+S]), int64 or int32 */
    int32_t nll_target_i64;
    const float* nll_mask;         /* device [N,S] */
    const float* g_loss;           /* device scalar: d objective / d loss (1 for plain training) */
    float* loss;                   /* device [2]: out = (loss, sum(mask)) */
    float* ws;                     /* workspace, echr_train_step_ws_floats floats, 256-byte aligned */
    int64_t ws_floats;
    float* flat_g;                 /* the gradient arena (n_flat floats): zero-filled by the call, then accumulated into */
    int64_t n_flat;
    float *flat_p, *adam_m, *adam_v; /* parameter arena and Adam moments (do_step = 1) */
    int32_t adam_step;             /* 1-based step count of THIS update */
    double lr, beta1, beta2, eps;
    float clip;                    /* clip_gradient value (inf = none) */
    int32_t do_step;               /* 0: stop after the backward pass (the caller reduces / inspects flat_g and steps itself) */
    int32_t overlap_encoder;       /* 1: the decoder forward's event-independent part runs on the library's second stream beside the event encoder */
    int32_t forward_only;          /* 1: stop after the criterion (validation loss, eval_utils.py:148-152) */
    int32_t n_active;              /* > 0: host_index carries the n_active time-major rows t*N + n with a non-zero criterion mask (ascending);
                                      training then forms logits, d logits and the logit-layer products on those rows only (the masked-out
                                      label positions behind a caption's end -- half of all rows at S = 20 -- cannot reach the loss).  0: all rows */
    int32_t host_nll;              /* 1: targets and mask travel in host_index too (nll_target / nll_mask are ignored) */
    int32_t prepared;              /* 1: echr_train_step_prepare already ran with these arguments (see there) */
    int32_t defer_update;          /* 1 (with g_tap and do_step): joint 'tap_cg' iteration -- the call returns as soon as g_tap and the loss are
                                      final in `stream` order; the parameter gradients and the clamp + Adam update complete on library-owned
                                      streams beside whatever the caller queues next (the proposal encoder's backward, train.py:313).  The
                                      caller MUST call echr_stream_join(stream) before reading parameters or gradients or freeing ws;
                                      the next echr_train_step joins by itself */
    int32_t handover;              /* 1 (with do_step = 0): data-parallel hand-over points.  The backward pass records an event where the
                                      logit layer's gradients (g_w_logit, g_b_logit) become final and one where the three LSTM layers'
                                      gradients (g_w_ih / g_w_hh / g_b_ih / g_b_hh) do -- both long before the call's last kernel -- so the
                                      caller can start the collectives on those ranges of flat_g while the rest of the backward tail runs:
                                      echr_handover_wait(which, s) makes stream s wait for the point (train.py:281-283,313-317: the reference
                                      sums m_batch gradients before one clamp + step; the data-parallel form sums over ranks) */
    echr_handover_fn handover_cb;  /* optional (with handover = 1): called on the HOST, from inside the call, at each hand-over point -- right
                                      after the last launch that writes the range has been queued on `stream` (a library-owned stream).  What the
                                      callback queues on `stream`, or orders behind it, runs as soon as the range is final and beside the rest of
                                      the backward tail: a collective library that orders its own stream behind the caller's CURRENT stream needs
                                      no further stream or event (RCCL through torch.distributed: make `stream` current for the call).  The
                                      callback must not call back into this library and must not block on the device */
    void* handover_user;           /* passed through to handover_cb */
    uint32_t* adam_applied;        /* optional (do_step = 1): the optimiser state's count of applied updates, see echr_clamp_adam_counted */
    /* ---- the reference's non-recipe options on the one-call path (round 6; all zero / NULL = the ECHR recipe) ---- */
    int32_t event_parts;           /* event_context_type (CaptionGenerator.py:106-130): 0 or 3 = 'ER3' (pooled C3D rows | anchor state), 1 = 'ER1'
                                      (pooled rows only: tsrm.Din = dec.D), 2 = 'ER2' (anchor states only: tsrm.Din = Ht) */
    const float *w_init, *b_init;  /* CG_init_feats_type (OldModel_NEW.py:72-96): init_linear.weight [3H, Dtot] / .bias [3H]; h(-1) = c(-1) =
                                      init_linear(cat([video | event | clip.mean(1)])) with the parts init_use_* select.  NULL = zero state.
                                      (The persistent recurrence kernels start from zero: such a call runs the launch-per-phase recurrences.) */
    float *g_w_init, *g_b_init;    /* their gradient slots in flat_g */
    int32_t init_use_v, init_use_e, init_use_c;
    int32_t vh_offset;             /* 'VH' in video_context_type with g_tap: dec.video[vh_offset : vh_offset + Ht] is tap.mean(0) over tap_rows rows
                                      (formed by the caller, CaptionGenerator.py:95-99); d loss / d video of that span -- from the decoder and from
                                      init_linear -- is spread back over the rows of g_tap.  With it the call does not defer its update.  -1 or
                                      g_tap = NULL: the scene vector is data */
    int32_t tap_rows;              /* rows of tap / g_tap (only read with vh_offset >= 0) */
    echr_mid_fn mid_cb;            /* optional, joint form (defer_update = 1): called on the HOST from inside the call once everything that leads to
                                      g_tap and the loss has been queued on `stream`, BEFORE the parameter-gradient tail and the update are forked onto
                                      the library's streams.  What the callback queues on `stream` -- the proposal encoder's backward and update,
                                      train.py:313 -- starts right behind g_tap, and the tail is ordered BEHIND it (the proposal encoder's reverse
                                      recurrence is a 64-workgroup latency chain that loses more beside chip-filling kernels than the overlap
                                      gains).  Called exactly once per call that takes the deferred form, never otherwise; may call other entry
                                      points of this library on `stream`; must not block on the device */
    void* mid_user;
} echr_train_step_args;
int64_t echr_train_step_ws_floats(const echr_train_step_args* a);
/* Back-to-back calls (round 6, "stage-ahead"): a call that ends with its own update (do_step = 1, not deferred) lets the NEXT call on the same
 * stream, workspace and arena start its index staging copy and the event encoder's position embedding -- which read host_index and nothing the
 * update writes -- on a library stream beside that update; everything else of the next call is ordered behind the update as before.  Nothing
 * changes for the caller: host_index is copied inside the call, and `ws` must stay untouched between calls as it always had to (it holds the
 * saved activations until the call's last kernel).  ECHR_STAGE_AHEAD=0 in the environment switches it off. */
int echr_train_step(const echr_train_step_args* a, void* stream);
/* Optional first half for the joint 'tap_cg' iteration (train.py:300-313): stages the index vectors and starts everything of the call that
 * does not read tap_feats (the decoder's event-independent part, the gradient-arena fill) on the library's prepare stream, so that it runs
 * beside the proposal encoder's forward the caller queues next.  Takes the arguments of the following echr_train_step (tap, g_tap, loss
 * slots may still be unset), which must then pass prepared = 1 and the same workspace.  Needs overlap_encoder = 1. */
int echr_train_step_prepare(const echr_train_step_args* a, void* stream);
/* Hand-over points of the LAST echr_train_step issued with handover = 1 (which: 0 = logit layer, 1 = LSTM layers): makes `stream` wait
 * until that range of flat_g is final.  0 = `stream` now waits; 1 = the call recorded no such point (a configuration without the
 * asynchronous tail: the range is final when the call's own stream reaches its end, like every other); < 0 error. */
/* Memory scope of the library's ordering points.  echr_handover_wait's events carry a system-scope release (their consumers sit outside the
 * library: collectives read by peers, copy engines).  echr_stream_join -- and every entry that joins the helper streams by itself -- orders
 * `stream` behind the helper streams with AGENT-scope events (no cache write-back for the host or other devices: ~2 us less per edge, a dozen
 * edges per iteration): kernels queued on `stream` afterwards see every result; a consumer on another device or the host must follow the join
 * with a system-scope point of its own on `stream` -- hipStreamSynchronize, or an event created without hipEventDisableSystemFence (what
 * torch.distributed records in front of every collective).  ECHR_EVENT_SYSTEM_FENCE=1 in the environment makes every library event system-scope. */
enum { ECHR_HANDOVER_LOGIT = 0, ECHR_HANDOVER_LSTM = 1 };
int echr_handover_wait(int which, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Optional per-kernel-class timing (HIP events recorded on the launch stream around every launch of the
 * class) for bench.py's roofline leg.  kind: 0 fp32 MFMA GEMM, 1 attention fwd, 2 attention bwd,
 * 3 attention post-pass, 4 recurrent grouped GEMM, 5 other, 6 bf16x3-split GEMM, 7 h2 (fp16-pair) GEMM, 8 h2 operand packing,
 * 9 persistent recurrence kernels (csrc/persist.hip).  echr_prof_read synchronises the recorded events and
 * returns the totals since echr_prof_enable(1): elapsed ms, algorithmic flops / bytes, launches.
 * Not thread-safe; never enabled on the product path.
 * ---------------------------------------------------------------------------------------------- */
int echr_prof_enable(int on);
int echr_prof_read(int kind, double* ms, double* flops, double* bytes, int64_t* launches);
/* total ms of `n` back-to-back event pairs around nothing: the fixed per-launch cost of event timing, which bench.py subtracts */
int echr_prof_event_overhead(double* ms, int64_t* n);

/* Runtime switches (defaults from the environment variables ECHR_GEMM_H2=1, ECHR_GEMM_BF16X3=1, ECHR_OVERLAP=0, ECHR_ATT_SLOTS=2):
 *   "gemm_h2"     0/1  run the decoder's large projections on h2-packed operands (two block-scaled fp16 planes, fp32-grade) or not
 *   "gemm_bf16x3" 0/1  (gemm_h2 = 0) use the three-plane bf16 split product for the large projections or the native fp32 MFMA
 *   "overlap"     0/1  run recurrence-independent GEMMs on a second HIP stream
 *   "att_slots"   2/4/8 attention slots per wave
 *   "persist"     0/1  (default 1, ECHR_PERSIST) run the teacher-forced recurrence as a pair of persistent launches (csrc/persist.hip)
 *                      when the shape allows (N <= 64, A <= 129, H = Ha = 512, D <= 512, a full 256-CU device), else one launch per phase
 *   "persist_bwd" 0/1  (default 1, ECHR_PERSIST_BWD) the same for the reverse recurrence of echr_decoder_bwd (independent of "persist")
 *   "persist_split" 0/1 (default 1, ECHR_PERSIST_SPLIT) attention chain as two half-chip machines of 32 event rows (0: one machine of 64)
 *   "persist_h2"  0/1  (default 1, ECHR_PERSIST_H2) fp16-pair (fp32-grade) MFMA products in the forward persistent kernels (0: exact
 *                      fp32 MFMAs); the reverse kernels always use fp32 MFMAs
 *   "persist_merge" 0/1 (default 1, ECHR_PERSIST_MERGE) each direction's pair (attention chain + the two plain LSTM streams) as ONE launch
 *                      of 256 workgroups on the caller's stream (0: two concurrent launches on two streams; needs two free hardware queues)
 *   "persist_kgroups" 0/1 (default 1, ECHR_PERSIST_KGROUPS) reverse LSTM role with the contraction split over 4 workgroup groups (128 KB of
 *                      ingest per workgroup and step plus a small partial-tile exchange) instead of 512 KB per workgroup
 *   "tsrm_fork"   0/1  (default 1, ECHR_TSRM_FORK) echr_tsrm_fwd runs its position branch (pair embedding -> fc1 -> fc2 gates) on the
 *                      library's helper stream beside the event-embedding / query / key products
 *   "persist_coop" 0/1  (default 0, ECHR_PERSIST_COOP) launch the persistent pairs with hipLaunchCooperativeKernel: the dispatch starts only
 *                      when all 256 workgroups can be co-resident, whatever else holds CUs (RCCL kernels of a data-parallel run, another
 *                      process on the device).  echr_amd.parallel / bench.py switch it on when the world size is > 1
 *   "persist_spin_limit" n  bound of every hand-off spin in polls (0 = default, about seconds; ECHR_PERSIST_SPIN_LIMIT)
 *   "persist_inject_timeout" code  diagnostic: the hand-off wait with this code (attention chain: 100000 * edge + timestep, edge 1 h1 / 2 q / 3 context / 4 d q / 5 d h / 6 d G / 7 d ATT) never completes, so the
 *                      launch aborts through its time-out path (tests/test_gpu_parity.py::test_persistent_abort_path); 0 = off
 *   "sst_persist" 0/1   (default 1, ECHR_SST_PERSIST) proposal encoder's recurrences as one persistent launch per direction (H = 512)
 *   "persist_sample" 0/1 (default 1, ECHR_PERSIST_SAMPLE) greedy decoding (echr_decoder_sample) as one persistent launch per 64 events, every step
 *                      on device (vocabularies of up to 15 360 words, the persistent forward kernel's shapes); 0 = one launch chain per step
 *   "posemb_rows" 0/1   (default 1, ECHR_POSEMB_ROWS) pairwise position embedding with one thread per frequency (contiguous stores); 0 = one
 *                      thread per 16 frequencies of a pair
 *   "posemb_packed" 0/1 (default 1, ECHR_POSEMB_PACKED) >= 4096 event pairs: the position embedding is written directly as the packed fc1 operand
 *   "persist_sample_max" n (default 512, ECHR_PERSIST_SAMPLE_MAX) greedy decoding of more than n events takes one batched launch chain per step (every
 *                      product an h2 GEMM over the N rows, fixed-order k loops: bitwise reproducible) instead of one persistent launch per 64 events;
 *                      0 = always the persistent form
 *   "persist_sample_force_eos" k  diagnostic: the persistent greedy decoder's logits of steps >= k - 1 favour <eos> for every event, so the launch takes
 *                      its early-stop path at a known step (tests); 0 = off
 *   "pair_tables" 0/1   (default 1, ECHR_PAIR_TABLES) inference over >= 16384 event pairs with known index bounds: fc1 tabulated over the distinct keys
 *   "gemm_skinny" 0/1   (default 1, ECHR_GEMM_SKINNY) the event encoder's fc2 over >= 4096 event pairs (512 -> <= 16 columns) as a streaming
 *                      16-row-tile kernel instead of the general tiles
 *   "embed_fused" 0/1   (default 0, ECHR_EMBED_FUSED) token-embedding gradient through echr_gemm_desc.row_index instead of d XT + scatter pass
 *                      (measured slower: atomics of all k-slices contend on the <bos> / frequent-word rows)
 *   "deterministic" 0/1 (default 0, ECHR_DETERMINISTIC) fixed-order accumulation: every order-dependent fp32 sum of a training iteration is replaced
 *                      by a fixed-order one -- products run one k loop per output tile (no split-K atomics; grouped problems that share an output run
 *                      one after the other), the recurrences run launch-per-phase with the accumulator products as one k loop per tile (the
 *                      persistent kernels' exchange adds are atomics: not used), column sums have one owner workgroup per 64 columns, the token
 *                      scatter-add and the anchor-row scatter have one owner per destination row and add in source order, the attention backward
 *                      writes per-chunk / per-(event, position) / per-workgroup slabs that fold launches sum in index order (scratch owned by the
 *                      library, grown on demand).  Two runs on the same inputs, parameters and dropout seed then agree bit for bit in loss and
 *                      every gradient (the reference's CPU path is run-to-run deterministic at a fixed thread count); cost: see DESIGN.md section 4h
 *   "tail_early"  0/1   (default 0, ECHR_TAIL_EARLY) fork the asynchronous decoder-backward tail ahead of the LSTM-layer gradient stage
 *   "persist_stamps" 0/1/2 diagnostic phase stamps of the forward (1) / reverse (2) pair, see echr_persist_read_stamps
 *   "gemm_tile", "gemm_split"  tuning overrides of the GEMM tile / split-K heuristics (0 = heuristics; tools/gemm_bench.py only) */
int echr_config_set(const char* key, int32_t value);

/* Diagnostic (never on the product path): with echr_config_set("persist_stamps", 1) the persistent recurrence kernels record
 * s_memrealtime stamps (100 MHz) at their phase boundaries for one workgroup per role; this copies the last launch's stamps
 * ([4 roles][S][16] uint64) to host memory (synchronises the device) and returns S (0 = nothing recorded). */
/* diagnostic: role index of workgroup `block` of the merged 256-workgroup recurrence launches under placement `mode` (csrc/persist.hip,
 * persist_role_index); a bijection of 0..255 for every mode. */
int32_t echr_persist_role_index(int32_t block, int32_t mode);
int echr_persist_read_stamps(uint64_t* dst, int32_t max_entries);

/* stand-alone element-wise clamp (misc/utils.py:107-111) for optimisers other than the fused one */
int echr_clamp(float* g, int64_t n, float clip, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ECHR_HIP_H */
