// One training iteration of the caption hot path as ONE library call (echr_train_step, include/echr_hip.h).
//
// Reference protocol: train.py:281-283 (zero_grad), :298 (cg_model forward = CaptionGenerator.py:17-30), :300 (LanguageModelCriterion,
// misc/utils.py:66-75), :313 (backward), :315-317 (clip_gradient + optimizer.step).  The pieces are the library's own entry points
// (event pooling, TSRM encoder, decoder forward, criterion, decoder backward with the fused criterion gradient, TSRM backward,
// clamp + Adam); what this file adds is the sequencing that echr_amd/functional.py + autograd otherwise do from Python -- ~75 ctypes
// calls, four autograd nodes and their callbacks per iteration, 1.2 ms of host time against 1.75 ms of GPU time -- as host C++:
// one call, the index vectors staged through a pinned ring, every buffer carved from one caller-owned workspace.
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "echr_common.h"
#include "echr_internal.h"

namespace echr {

static inline long up64(long n) { return (n + 63) / 64 * 64; }
// echr_dec_grads.async_tail of the step's backward: 2 (default) = only d event on the caller's stream, 1 = the tail alone (ECHR_ASYNC_LEVEL: A/B)
static int async_level() {
    static const int lv = [] { const char* e = getenv("ECHR_ASYNC_LEVEL"); return (e && e[0] == '1') ? 1 : 2; }();
    return lv;
}

struct StepWs { long idx, ech, tsrm_ws, event, logp, dec_ws, dec_ws_bwd, g_event, g_ech, tsrm_ws_bwd, h0, g_h0, init_feats, init_dfeats, g_video, g_video_init, total; };

static inline int init_feats_width(const echr_train_step_args* a) {
    return (a->init_use_v ? a->dec.Dv : 0) + (a->init_use_e ? a->dec.De : 0) + (a->init_use_c ? a->dec.D : 0);
}

static StepWs carve_step(const echr_train_step_args* a) {
    StepWs w;
    long off = 0;
    auto take = [&](long n) { long o = off; off += up64(n); return o; };
    const echr_tsrm_args& t = a->tsrm;
    const echr_dec_args& d = a->dec;
    w.idx = take((long)(3 + 4 * d.S) * d.N);              // int32: ev_start | ev_len | ind | tokens [S,N] | active rows [<= S*N] | targets [N,S] | mask fp32 [N,S]
    w.ech = take((long)t.N * t.Din);
    w.tsrm_ws = take(echr_tsrm_ws_floats(t.N, t.Din, t.Df, t.Do, t.G));
    w.event = take((long)t.N * t.Do);
    w.logp = take((long)d.N * d.S * d.V1);
    w.dec_ws = take(echr_decoder_ws_floats(&d));
    w.dec_ws_bwd = take(echr_decoder_ws_bwd_floats(&d));
    w.g_event = take((long)d.N * d.De);
    w.g_ech = take((long)t.N * t.Din);
    w.tsrm_ws_bwd = take(echr_tsrm_ws_bwd_floats(t.N, t.Din, t.Df, t.Do, t.G));
    // non-recipe options: initial state (h0, its gradient, init_linear's input rows and their gradient), d video
    w.h0 = w.g_h0 = w.init_feats = w.init_dfeats = -1;
    if (a->w_init) {
        const long h3 = 3L * d.H, dt = init_feats_width(a);
        w.h0 = take((long)d.N * h3); w.g_h0 = take((long)d.N * h3); w.init_feats = take((long)d.N * dt); w.init_dfeats = take((long)d.N * dt);
    }
    w.g_video = take(d.Dv); w.g_video_init = take(d.Dv);
    w.total = off;
    return w;
}

// Host -> device staging of the per-iteration index vectors: a ring of pinned buffers, each guarded by the event recorded behind the
// copy that reads it, so the host may run several iterations ahead of the GPU without ever rewriting a buffer a copy still reads.
struct PinRing {
    static constexpr int SLOTS = 8;
    void* buf[SLOTS] = {};
    size_t cap[SLOTS] = {};
    hipEvent_t done[SLOTS] = {};
    bool used[SLOTS] = {};
    int next = 0;
};
static PinRing& ring() { static PinRing r; return r; }
static hipEvent_t& ring_last() { static hipEvent_t e = nullptr; return e; }          // the event behind the last staging copy

__global__ __launch_bounds__(256) void stage_copy_kernel(const int32_t* __restrict__ src, int32_t* __restrict__ dst, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] = __builtin_nontemporal_load(src + i);
}
static bool stage_kernel_ok() {
    static const bool on = [] { const char* e = getenv("ECHR_STAGE_KERNEL"); return !(e && e[0] == '0'); }();      // A/B switch
    return on;
}
static int stage_indices(const int32_t* host, int32_t* dev, size_t bytes, hipStream_t st) {
    PinRing& r = ring();
    const int s = r.next;
    r.next = (r.next + 1) % PinRing::SLOTS;
    if (r.used[s] && hipEventSynchronize(r.done[s]) != hipSuccess) { set_error("train_step: staging event failed"); return -5; }
    if (r.cap[s] < bytes) {
        if (r.buf[s]) (void)hipHostFree(r.buf[s]);
        r.buf[s] = nullptr; r.cap[s] = 0;
        const size_t want = bytes < 65536 ? 65536 : bytes * 2;
        if (hipHostMalloc(&r.buf[s], want, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); set_error("train_step: pinned staging buffer (%zu bytes) unavailable", want); return -12; }
        r.cap[s] = want;
    }
    if (!r.done[s] && hipEventCreateWithFlags(&r.done[s], echr::sync_event_flags()) != hipSuccess) { set_error("train_step: event create failed"); return -5; }
    memcpy(r.buf[s], host, bytes);
    // the pinned buffer is read by a copy KERNEL (host-coherent memory is device-visible at the same address): an in-stream launch of ~3 us.
    // hipMemcpyAsync takes the DMA-engine path for transfers of this size (~19 KB), whose start-up latency (tens of us) sat at the head of
    // every iteration in front of everything else
    const int n = (int)(bytes / 4);
    if (stage_kernel_ok()) {
        hipLaunchKernelGGL(stage_copy_kernel, dim3((n + 255) / 256), dim3(256), 0, st, static_cast<const int32_t*>(r.buf[s]), dev, n);
        if (hipGetLastError() != hipSuccess) { set_error("train_step: index upload failed"); return -5; }
    } else if (hipMemcpyAsync(dev, r.buf[s], bytes, hipMemcpyHostToDevice, st) != hipSuccess) {
        (void)hipGetLastError();
        set_error("train_step: index upload failed");
        return -5;
    }
    if (hipEventRecord(r.done[s], st) != hipSuccess) { (void)hipGetLastError(); set_error("train_step: index upload failed"); return -5; }
    r.used[s] = true;
    ring_last() = r.done[s];
    return 0;
}

// Stage-ahead (round 6): the LAST kernel of an iteration is clamp + Adam, ~95 us of pure HBM streaming on the caller's stream, and the FIRST
// things of the next one -- the index staging copy and, behind it, the event encoder's position embedding (indices only: the head of the chain
// the forward recurrence waits for) -- need nothing that update writes.  When a call ends with its own update it records an event in front of
// it (every reader of the workspace and the index region is complete there: the helper streams were just joined); the next call on the same
// stream, workspace and arena stages its indices on the TAIL stream behind that event, so that copy and embedding run beside the update, and
// orders everything else behind the caller's stream's position at its entry (the update and whatever the caller queued since).
struct StageAhead { hipEvent_t pre = nullptr, post = nullptr; bool valid = false; hipStream_t st = nullptr; const void* ws = nullptr; const void* flat_g = nullptr; bool init = false, ok = false; };
static StageAhead& stage_ahead() {
    static StageAhead s;
    if (!s.init) {
        s.init = true;
        const char* e = getenv("ECHR_STAGE_AHEAD");          // A/B switch
        s.ok = !(e && e[0] == '0') && hipEventCreateWithFlags(&s.pre, echr::sync_event_flags()) == hipSuccess &&
               hipEventCreateWithFlags(&s.post, echr::sync_event_flags()) == hipSuccess;
        if (!s.ok) (void)hipGetLastError();
    }
    return s;
}

}  // namespace echr

using namespace echr;

#define RC(x) do { int rc__ = (x); if (rc__) return rc__; } while (0)

extern "C" int64_t echr_train_step_ws_floats(const echr_train_step_args* a) { return a ? carve_step(a).total : -1; }

// diagnostic (ECHR_STEP_TIMING=1): HIP events at the phase boundaries of the call on the caller's stream; every 50th call prints the averages
struct StepTiming { hipEvent_t e[6][8] = {}; int calls = 0; bool on = false, init = false; double acc[5] = {0, 0, 0, 0, 0}; int n = 0; };
static StepTiming& step_timing() {
    static StepTiming t;
    if (!t.init) {
        t.init = true;
        const char* e = getenv("ECHR_STEP_TIMING");
        t.on = e && e[0] == '1';
        if (t.on) for (auto& row : t.e) for (auto& ev : row) if (hipEventCreate(&ev) != hipSuccess) t.on = false;
    }
    return t;
}
static void step_mark(int k, hipStream_t st) {
    StepTiming& t = step_timing();
    if (t.on) (void)hipEventRecord(t.e[k][t.calls % 8], st);
}
static void step_timing_end() {
    StepTiming& t = step_timing();
    if (!t.on) return;
    const int slot = (t.calls + 1) % 8;          // the oldest recorded call: long finished
    if (t.calls >= 8) {
        float ms;
        bool ok = true;
        double d[5];
        for (int k = 0; k < 5 && ok; ++k) { ok = hipEventElapsedTime(&ms, t.e[k][slot], t.e[k + 1][slot]) == hipSuccess; d[k] = ms; }
        if (ok) { for (int k = 0; k < 5; ++k) t.acc[k] += d[k]; ++t.n; } else (void)hipGetLastError();
    }
    ++t.calls;
    if (t.n && t.calls % 50 == 0) {
        fprintf(stderr, "[train_step] encoder + forward (to d logits) %.3f ms | decoder backward on this stream %.3f | event encoder backward %.3f | wait for the helper streams %.3f | clamp + Adam %.3f\n",
                t.acc[0] / t.n, t.acc[1] / t.n, t.acc[2] / t.n, t.acc[3] / t.n, t.acc[4] / t.n);
        for (double& x : t.acc) x = 0;
        t.n = 0;
    }
}

// joint mode: the helper streams' work of one call
struct JointPending { echr_dec_args d; echr_dec_grads g; echr_tsrm_args t; echr_tsrm_grads tg; echr_dropout drop; echr_train_step_args args; };
// every parameter gradient (decoder: helper streams forked from `src`; event encoder: behind them on the prepare stream), then clamp + Adam on
// the tail stream, published for echr_stream_join
static int joint_finish(JointPending& jp, hipStream_t src) {
    RC(decoder_bwd_parts(&jp.d, &jp.g, &jp.drop, src, 2));
    hipStream_t s2 = aux2_stream();
    RC(tsrm_bwd_parts(&jp.t, &jp.tg, &jp.drop, s2, 2));
    hipStream_t ts = helpers_merge_to_tail();
    if (!ts) return -5;
    const echr_train_step_args& a = jp.args;
    RC(echr_clamp_adam_counted(a.flat_p, a.flat_g, a.adam_m, a.adam_v, a.n_flat, a.adam_step, a.lr, a.beta1, a.beta2, a.eps, a.clip, a.adam_applied, ts));
    return tail_publish();
}

// the decoder arguments of one call (shared by echr_train_step_prepare and echr_train_step: both must describe the same launch)
static echr_dec_args step_dec_args(const echr_train_step_args* a, const StepWs& L, const int32_t* idx) {
    const int N = a->dec.N;
    echr_dec_args d = a->dec;
    d.ev_start = idx; d.ev_len = idx + N; d.tokens = idx + 3 * N;
    d.ws = a->ws + L.dec_ws; d.logp = a->ws + L.logp; d.event = nullptr; d.prepared = 0;
    d.train = a->forward_only ? 0 : 1;
    // the gradient arena is zero-filled by the forward's first fill launch (beside the event encoder), not in front of the reverse recurrence
    d.zero_extra = a->forward_only ? nullptr : a->flat_g; d.zero_extra_count = a->n_flat;
    // (the initial-state buffer is filled behind the event encoder; its ADDRESS is part of the description from the start, so that the prepare
    // half and the forward agree on which recurrence kernels run -- the persistent ones start from zero and decline h0)
    d.h0 = a->w_init ? a->ws + L.h0 : nullptr;
    return d;
}
static size_t step_index_count(const echr_train_step_args* a) {
    return (size_t)(3 + a->dec.S) * a->dec.N + (size_t)a->n_active + (a->host_nll ? 2 * (size_t)a->dec.S * a->dec.N : 0);
}

// Optional first half of echr_train_step for the joint 'tap_cg' iteration (train.py:300-313): everything of the call that does not read
// tap_feats -- index staging and the decoder's event-independent part (attention projections of the video, token-side gates, operand packs,
// the gradient-arena fill) -- is started on the library's prepare stream BEFORE the caller queues the proposal encoder's forward, a 64-workgroup
// persistent launch that leaves three quarters of the chip idle.  The following echr_train_step takes the same arguments plus prepared = 1
// (tap / g_tap may be filled in only then).
extern "C" int echr_train_step_prepare(const echr_train_step_args* a, void* stream) {
    ECHR_REQUIRE(a && a->ws && a->host_index && a->flat_g, "train_step_prepare: missing buffers");
    ECHR_REQUIRE(a->overlap_encoder, "train_step_prepare: needs overlap_encoder = 1");
    hipStream_t st = (hipStream_t)stream;
    RC(join_tail(st));
    const StepWs L = carve_step(a);
    ECHR_REQUIRE(a->ws_floats >= L.total, "train_step_prepare: workspace holds %lld floats, %ld needed (echr_train_step_ws_floats)", (long long)a->ws_floats, L.total);
    ECHR_REQUIRE(a->n_active >= 0 && a->n_active <= a->dec.S * a->dec.N, "train_step: n_active out of range");
    int32_t* idx = reinterpret_cast<int32_t*>(a->ws + L.idx);
    RC(stage_indices(a->host_index, idx, sizeof(int32_t) * step_index_count(a), st));
    echr_dec_args d = step_dec_args(a, L, idx);
    // the event encoder's position branch reads indices and parameters only: it starts here too (ECHR_PREPARE_POS=0: with the second half, behind
    // the proposal encoder's forward, where its ~70 us chain sits in front of the forward recurrence)
    static const bool prep_pos = [] { const char* e = getenv("ECHR_PREPARE_POS"); return !(e && e[0] == '0'); }();      // A/B switch
    if (prep_pos) {
        echr_tsrm_args t = a->tsrm;
        t.ech = a->ws + L.ech; t.ev_start = idx; t.ev_len = idx + a->dec.N; t.ws = a->ws + L.tsrm_ws; t.out = a->ws + L.event;
        t.inference = 0; t.max_len = 0; t.max_span = 0;
        const int rc = tsrm_position_early(&t, st);
        if (rc) { (void)tsrm_position_early(nullptr, nullptr); (void)aux_join(st); return rc; }
    }
    const int rc = echr_decoder_fwd_prepare(&d, stream);
    if (rc && prep_pos) { (void)tsrm_position_early(nullptr, nullptr); (void)aux_join(st); }
    return rc;
}

extern "C" int echr_train_step(const echr_train_step_args* a, void* stream) {
    ECHR_REQUIRE(a && a->ws && a->host_index && a->loss && a->g_loss && a->flat_g && (a->tap || a->event_parts == 1), "train_step: missing buffers");
    ECHR_REQUIRE(!a->prepared || a->overlap_encoder, "train_step: prepared = 1 needs overlap_encoder = 1");
    const int parts = a->event_parts ? a->event_parts : 3;
    ECHR_REQUIRE(parts >= 1 && parts <= 3, "train_step: event_parts must be 0..3");
    const int De_c3d = (parts & 1) ? a->dec.D : 0, De_tap = (parts & 2) ? a->Ht : 0;          // the two halves of the event encoder's input rows
    ECHR_REQUIRE(a->tsrm.N == a->dec.N && a->tsrm.Do == a->dec.De && a->tsrm.Din == De_c3d + De_tap, "train_step: encoder / decoder shapes disagree");
    const bool vh = a->g_tap && a->vh_offset >= 0;
    ECHR_REQUIRE(!vh || (a->vh_offset + a->Ht <= a->dec.Dv && a->tap_rows > 0), "train_step: vh_offset / tap_rows do not describe a span of dec.video");
    ECHR_REQUIRE(!a->w_init || (a->b_init && (a->forward_only || (a->g_w_init && a->g_b_init)) && init_feats_width(a) > 0), "train_step: init_linear pointers incomplete");
    ECHR_REQUIRE(!a->do_step || (a->flat_p && a->adam_m && a->adam_v && a->adam_step >= 1), "train_step: optimiser state missing");
    hipStream_t st = (hipStream_t)stream;
    RC(join_tail(st));          // (a deferred update of the previous call: it reads the index region this call is about to restage)
    const StepWs L = carve_step(a);
    ECHR_REQUIRE(a->ws_floats >= L.total, "train_step: workspace holds %lld floats, %ld needed (echr_train_step_ws_floats)", (long long)a->ws_floats, L.total);
    float* ws = a->ws;
    const int N = a->dec.N, S = a->dec.S;
    step_mark(0, st);
    int32_t* idx = reinterpret_cast<int32_t*>(ws + L.idx);
    ECHR_REQUIRE(a->n_active >= 0 && a->n_active <= S * N, "train_step: n_active out of range");
    StageAhead& sa = stage_ahead();
    const bool ahead = sa.ok && sa.valid && !a->prepared && a->overlap_encoder && sa.st == st && sa.ws == a->ws && sa.flat_g == a->flat_g &&
                       helpers_available() && tail_stream_raw();
    sa.valid = false;          // (consumed, or void: whatever this call does, the recorded pair no longer brackets the stream's last work)
    if (ahead) {
        hipStream_t ts = tail_stream_raw();
        if (hipStreamWaitEvent(ts, sa.pre, 0) != hipSuccess) { set_error("train_step: stream wait failed"); return -5; }
        RC(stage_indices(a->host_index, idx, sizeof(int32_t) * step_index_count(a), ts));
        // the caller's stream: behind the staging copy (its own position is already behind the update).  The prepare stream: behind the
        // caller's stream's position AT ENTRY -- the update, and whatever the caller queued since (this call's inputs: an upload of the next
        // video's features, the proposal encoder's forward; a write to the parameters) -- since the staging event it forks from no longer
        // implies that position
        if (hipEventRecord(sa.post, st) != hipSuccess) { set_error("train_step: event record failed"); return -5; }
        RC(prep_stream_wait(sa.post));
        if (hipStreamWaitEvent(st, ring_last(), 0) != hipSuccess) { set_error("train_step: stream wait failed"); return -5; }
    } else if (!a->prepared) RC(stage_indices(a->host_index, idx, sizeof(int32_t) * step_index_count(a), st));
    const int32_t *ev_start = idx, *ev_len = idx + N, *ind = idx + 2 * N, *active = idx + (3 + S) * N;          // (tokens at idx + 3 N: step_dec_args)
    const void* nll_target = a->nll_target;
    const float* nll_mask = a->nll_mask;
    int nll_i64 = a->nll_target_i64;
    if (a->host_nll) {          // targets / mask came with the index vectors
        nll_target = active + a->n_active;
        nll_mask = reinterpret_cast<const float*>(active + a->n_active + (size_t)S * N);
        nll_i64 = 0;
    }
    ECHR_REQUIRE(nll_target && nll_mask, "train_step: criterion targets / mask missing");

    echr_dec_args d = step_dec_args(a, L, idx);
    echr_tsrm_args t = a->tsrm;
    t.ech = ws + L.ech; t.ev_start = ev_start; t.ev_len = ev_len; t.ws = ws + L.tsrm_ws; t.out = ws + L.event;
    t.inference = 0; t.max_len = 0; t.max_span = 0;
    // the event encoder's position branch (pair embedding -> fc1 -> fc2 gates: indices and parameters only) starts right behind the staging
    // CaptionGenerator.forward (:23-30): the decoder's event-independent part starts on the library's second stream and overlaps the event encoder
    if (a->overlap_encoder && !a->prepared) {
        static const bool one_event = [] { const char* e = getenv("ECHR_ONE_FORK_EVENT"); return !(e && e[0] == '0'); }();      // A/B switch
        if (one_event || ahead) fork_event(ring_last());          // both forks hang off the staging copy's event: no further record in front of event pooling
        int rc2 = tsrm_position_early(&t, st);
        if (!rc2) rc2 = echr_decoder_fwd_prepare(&d, stream);
        fork_event(nullptr);
        if (rc2) {
            // the position branch may already be queued on the tail stream: forget it (a later echr_tsrm_fwd on this workspace must not take the
            // `early` form with stale gates) and order this stream behind what was queued, as the event encoder's own failure path does
            (void)tsrm_position_early(nullptr, nullptr);
            (void)aux_join(st);
            (void)aux2_join(st);          // (a prepare chain that failed half-way published nothing: join the stream itself)
            (void)echr_decoder_fwd_prepare_cancel(stream);
            return rc2;
        }
    }
    int rc = echr_event_pool_gather_fwd(a->dec.c3d, a->tap, ev_start, ev_len, ind, ws + L.ech, N, De_c3d, De_tap, stream);       // :106-128
    if (!rc) rc = echr_tsrm_fwd(&t, &a->drop, stream);                                                                       // :129
    if (rc) { (void)tsrm_position_early(nullptr, nullptr); if (a->overlap_encoder) (void)echr_decoder_fwd_prepare_cancel(stream); return rc; }
    d.event = ws + L.event; d.prepared = a->overlap_encoder ? 1 : 0;
    // OldModel.init_hidden with CG_init_feats_type (OldModel_NEW.py:72-96): h(-1) = c(-1) = init_linear(cat(selected contexts))
    echr_init_state_args ia;
    memset(&ia, 0, sizeof(ia));
    if (a->w_init) {
        ia.N = N; ia.Dv = a->dec.Dv; ia.De = a->dec.De; ia.D = a->dec.D; ia.H3 = 3 * a->dec.H;
        ia.use_v = a->init_use_v; ia.use_e = a->init_use_e; ia.use_c = a->init_use_c; ia.A = a->dec.A;
        ia.video = a->dec.video; ia.event = ws + L.event; ia.c3d = a->dec.c3d; ia.ev_start = ev_start; ia.ev_len = ev_len;
        ia.w = a->w_init; ia.b = a->b_init; ia.feats = ws + L.init_feats; ia.h0 = ws + L.h0;
        rc = echr_init_state_fwd(&ia, stream);
        if (rc) { if (a->overlap_encoder) (void)echr_decoder_fwd_prepare_cancel(stream); return rc; }
    }
    echr_dec_grads g = a->dec_g;
    g.g_event = ws + L.g_event; g.g_logp = nullptr;
    g.nll_target = static_cast<const int32_t*>(nll_target); g.nll_target_i64 = nll_i64; g.nll_mask = nll_mask;
    g.active_rows = a->n_active > 0 ? active : nullptr; g.n_active = a->n_active;
    g.g_loss = a->g_loss; g.nll_msum = a->loss + 1;
    g.g_h0 = a->w_init ? ws + L.g_h0 : nullptr;
    if (vh) g.g_video = ws + L.g_video;          // (the caller's own g_video, if any, is not filled then: the span is routed into g_tap)
    g.ws_bwd = ws + L.dec_ws_bwd; g.zeroed = 1; g.phase = 0; g.async_tail = async_level();
    if (!a->forward_only && a->overlap_encoder) RC(decoder_bwd_scratch_ahead(&d, &g));          // the backward's scratch fill: behind the prepare chain, not between the recurrences
    // forward (:30) + LanguageModelCriterion (misc/utils.py:66-75).  Training: log-softmax, criterion and its gradient are ONE pass over the
    // logits (d logits land in the backward workspace, the log-probs are never written; the loss is summed behind the backward pass, where
    // this stream waits for the helper stream anyway).  forward_only: the plain log-softmax + criterion, loss[0] = loss, loss[1] = sum(mask)
    bool fused_nll = false, compact = false;
    if (a->forward_only) RC(echr_decoder_fwd(&d, &a->drop, stream));
    else RC(decoder_fwd_fused(&d, &g, &a->drop, stream, &fused_nll, &compact));
    if (!fused_nll) {
        if (nll_i64) RC(echr_nll_loss_fwd_i64(d.logp, static_cast<const int64_t*>(nll_target), nll_mask, a->loss, N, S, d.V1, stream));
        else RC(echr_nll_loss_fwd(d.logp, static_cast<const int32_t*>(nll_target), nll_mask, a->loss, N, S, d.V1, stream));
    }
    if (a->forward_only) return 0;
    step_mark(1, st);
    g.dlg_ready = fused_nll ? 1 : 0;
    if (fused_nll) g.nll_msum = nullptr;
    if (!compact) { g.active_rows = nullptr; g.n_active = 0; }          // (the forward kept all rows: vocabulary beyond the register-resident kernels, h2 off)

    // backward (train.py:313): criterion gradient in fused form
    g.zero_extra = nullptr; g.zero_extra_count = 0;
    handover_request(false);          // (hand-over points of an earlier call are void from here on)
    if (a->defer_update && a->g_tap && a->do_step && g.async_tail == 2 && fused_nll && config().gemm_h2 && helpers_available() && !vh && !a->w_init) {
        // Joint 'tap_cg' iteration (train.py:300-313): the proposal encoder's backward -- a 64-workgroup persistent launch that leaves three
        // quarters of the chip idle -- waits for d tap_feats alone.  The chain that leads to it (late fusion, reverse recurrence, d event,
        // the event encoder's attention backward, d ech) runs first and alone on the caller's stream; every parameter gradient and the
        // clamp + Adam update follow on the helper streams, forked behind it, and are NOT joined here: they overlap whatever the caller
        // queues next.  echr_stream_join (and the next echr_train_step / decoder call) waits for them.
        echr_tsrm_grads tg = a->tsrm_g;
        tg.g_ech = ws + L.g_ech; tg.g_out = ws + L.g_event; tg.ws_bwd = ws + L.tsrm_ws_bwd; tg.zeroed = 1;
        RC(decoder_bwd_parts(&d, &g, &a->drop, stream, 1));
        RC(tsrm_bwd_parts(&t, &tg, &a->drop, stream, 1));
        if (De_tap > 0) RC(echr_event_pool_gather_bwd(ws + L.g_ech, ind, a->g_tap, N, De_c3d, De_tap, stream));
        RC(decoder_fused_loss(&d, &g, a->loss, st));
        // the caller's hook: the proposal encoder's backward (+ update) goes onto `stream` HERE, right behind g_tap; the helper streams fork
        // behind it (joint_finish), so the chip-filling tail never shares CUs with that 64-workgroup latency chain
        if (a->mid_cb) a->mid_cb(stream, a->mid_user);
        JointPending jp;
        jp.d = d; jp.g = g; jp.t = t; jp.tg = tg; jp.drop = a->drop; jp.args = *a;
        // (issuing the helper work only after the caller has queued the proposal encoder's backward -- a second entry point, tried -- is worse:
        // 3.21 vs 3.00 ms on c5.  The host needs ~0.5 ms to get from here to that launch anyway, the helpers fill exactly that gap, and a
        // persistent recurrence that shares its CUs with GEMM workgroups from its first step on loses more than the gap is worth)
        return joint_finish(jp, st);
    }
    handover_request(a->handover && !a->do_step, a->handover_cb, a->handover_user);
    rc = echr_decoder_bwd(&d, &g, &a->drop, stream);
    handover_close();          // (the events stay valid for echr_handover_wait; later backward passes do not re-record them)
    RC(rc);
    step_mark(2, st);
    if (a->w_init) {
        // d h0 -> init_linear's gradients, d event (ADDED to what the decoder left there, ahead of the event encoder's backward), d video
        echr_init_state_grads ig;
        ig.g_h0 = ws + L.g_h0; ig.g_w = a->g_w_init; ig.g_b = a->g_b_init; ig.zeroed = 1;
        ig.g_video = (vh && a->init_use_v) ? ws + L.g_video_init : nullptr;
        ig.g_event = a->init_use_e ? ws + L.g_event : nullptr;
        ig.dfeats = ws + L.init_dfeats;
        RC(echr_init_state_bwd(&ia, &ig, stream));
    }
    echr_tsrm_grads tg = a->tsrm_g;
    // d ech (the gradient of the event encoder's INPUT rows) only matters when d tap_feats is asked for: c3d features are data
    tg.g_ech = (a->g_tap && De_tap > 0) ? ws + L.g_ech : nullptr; tg.g_out = ws + L.g_event; tg.ws_bwd = ws + L.tsrm_ws_bwd; tg.zeroed = 1;
    tsrm_bwd_defer_join(true);          // (this call's workspace outlives the echr_stream_join below)
    rc = echr_tsrm_bwd(&t, &tg, &a->drop, stream);
    tsrm_bwd_defer_join(false);
    RC(rc);
    if (a->g_tap && De_tap > 0) RC(echr_event_pool_gather_bwd(ws + L.g_ech, ind, a->g_tap, N, De_c3d, De_tap, stream));
    if (fused_nll) RC(decoder_fused_loss(&d, &g, a->loss, st));
    step_mark(3, st);
    RC(echr_stream_join(stream));          // the decoder backward's asynchronous tail: every gradient is final in `stream` order now
    if (vh) {
        // scene context 'VH' = tap.mean(0) (CaptionGenerator.py:95-99): d tap[r, :] += d video[vh span] / rows, for the decoder's d video (final
        // behind the join: it is formed in the LSTM-layer stage on a helper stream) and init_linear's
        RC(echr_col_mean_bwd(ws + L.g_video + a->vh_offset, a->tap_rows, a->Ht, a->Ht, a->g_tap, stream));
        if (a->w_init && a->init_use_v) RC(echr_col_mean_bwd(ws + L.g_video_init + a->vh_offset, a->tap_rows, a->Ht, a->Ht, a->g_tap, stream));
    }
    step_mark(4, st);
    if (a->do_step) {                      // clip_gradient + Adam (misc/utils.py:107-111, train.py:315-317)
        const bool rec = sa.ok && hipEventRecord(sa.pre, st) == hipSuccess;          // (stage-ahead: every helper stream was joined just above)
        RC(echr_clamp_adam_counted(a->flat_p, a->flat_g, a->adam_m, a->adam_v, a->n_flat, a->adam_step, a->lr, a->beta1, a->beta2, a->eps, a->clip, a->adam_applied, stream));
        if (rec) { sa.valid = true; sa.st = st; sa.ws = a->ws; sa.flat_g = a->flat_g; }
        else (void)hipGetLastError();
    }
    step_mark(5, st);
    step_timing_end();
    return 0;
}
