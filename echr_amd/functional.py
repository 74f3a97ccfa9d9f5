"""torch.autograd bindings of the HIP hot path (one Function per fused region).

PyTorch is plumbing here: it owns device memory, streams and the autograd graph edges; all arithmetic
happens in libechr_hip.so (include/echr_hip.h).  Every call fails loudly when the library or the GPU
is missing -- there is no eager/PyTorch fallback.
"""
import ctypes as C
import os

import numpy as np
import torch

from . import _lib as L


class DropState:
    """Dropout configuration of one forward call (counter-based; see echr_amd/philox.py)."""

    def __init__(self, seed=0, offset=0, training=False, p_tsrm=0.3, p_h=0.5, p_out=0.5):
        self.seed, self.offset, self.training = int(seed), int(offset), bool(training)
        self.p_tsrm, self.p_h, self.p_out = float(p_tsrm), float(p_h), float(p_out)

    def c(self):
        return L.Dropout(self.seed & 0xFFFFFFFFFFFFFFFF, self.offset & 0xFFFFFFFF, 1 if self.training else 0,
                         self.p_tsrm, self.p_h, self.p_out)


def event_index_tensors(soi_select_list, ind_select_list, device, n_rows=None):
    """(ev_start, ev_len, ind) int32 device tensors + max length A from the reference's list inputs
    (numpy int arrays or python lists; CaptionGenerator.py:17, train.py:265-271)."""
    soi = np.asarray(soi_select_list, dtype=np.int64).reshape(-1, 2)
    ind = np.asarray(ind_select_list, dtype=np.int64).reshape(-1)
    lens = soi[:, 1] - soi[:, 0]
    if len(soi) == 0 or lens.min() <= 0:
        raise ValueError('every event needs at least one segment (soi=%s)' % (soi.tolist(),))
    if len(ind) != len(soi):
        raise ValueError('ind_select_list and soi_select_list differ in length (%d vs %d)' % (len(ind), len(soi)))
    if n_rows is not None and (soi.min() < 0 or soi[:, 1].max() > n_rows or ind.min() < 0 or ind.max() >= n_rows):
        raise ValueError('event intervals / anchors fall outside the %d feature rows' % n_rows)
    packed = np.stack([soi[:, 0], lens, ind]).astype(np.int32)
    t = upload(torch.from_numpy(packed), device)
    ev_start, ev_len = t[0].contiguous(), t[1].contiguous()
    c2 = 2 * soi[:, 0] + lens
    ev_len.echr_bounds = (int(lens.max()), int(c2.max() - c2.min()))      # host-known index bounds (echr_tsrm_args.max_len / max_span)
    return ev_start, ev_len, t[2].contiguous(), int(lens.max())


def upload(host_tensor, device):
    """Small host -> device copy that does not stall the host: staged through a pinned buffer from PyTorch's caching host allocator (it
    recycles a block only after the copy that reads it has completed).  A copy from pageable memory makes the host wait until the
    stream has drained, i.e. for the whole previous iteration, and the GPU then idles while the host issues the next one."""
    if host_tensor.is_cuda:
        return host_tensor
    if device.type != 'cuda':
        return host_tensor.to(device)
    stage = torch.empty(host_tensor.shape, dtype=host_tensor.dtype, pin_memory=True)
    stage.copy_(host_tensor)
    return stage.to(device, non_blocking=True)


def rows_disjoint(soi_select_list):
    """True when no two events share a segment row (then d P_all rows are stored instead of atomically added)."""
    soi = np.asarray(soi_select_list, dtype=np.int64).reshape(-1, 2)
    o = soi[np.argsort(soi[:, 0], kind='stable')]
    return bool(np.all(o[1:, 0] >= o[:-1, 1]))


FUSED_NLL = [os.environ.get('ECHR_FUSED_NLL', '1') != '0']        # see MaskedNLL.backward
ASYNC_LEVEL = [1 if os.environ.get('ECHR_ASYNC_LEVEL', '2') == '1' else 2]
ASYNC_TAIL = [os.environ.get('ECHR_ASYNC_TAIL', '1') != '0']        # decoder backward: run the last stage on a second stream (arena path); tests may switch it off


def _f32c(t):
    # (inside autograd.Function.forward / backward grad mode is off: inputs can be used and saved as they are)
    if t.dtype is torch.float32 and t.is_contiguous() and (not t.requires_grad or not torch.is_grad_enabled()):
        return t
    return t.detach().to(torch.float32).contiguous()


class GradSink(object):
    """Where a backward Function should write its parameter gradients: the module's flat arena (echr_amd/arena.py).

    Used only when EVERY parameter of the group currently has .grad None -- then the Function zero-fills the group's span
    once, lets the library accumulate into it (`zeroed` = 1) and returns fresh arena views that autograd adopts as .grad
    without a copy.  If any .grad already exists (gradient accumulation) the Function falls back to private buffers so
    that autograd's `grad += new` sees two distinct tensors."""

    def __init__(self, arena, params):
        self.arena, self.params = arena, list(params)
        self.slots = [arena.slot(p) for p in self.params]

    def usable(self):
        ok = all(s is not None for s in self.slots) and all(p.grad is None for p in self.params)
        if not ok:
            # gradient accumulation (a second backward before the optimiser step): a clamp that clip_gradient left to the fused step
            # kernel has to be applied to the running gradient NOW, before autograd adds this backward's gradients to it -- the
            # reference computes clamp(clamp(g1) + g2), not clamp(g1 + g2) (train.py:313-317)
            self.arena.flush_deferred_clamp()
        return ok

    def take(self, defer_zero=False):
        """Fresh arena views of the group's gradients over a zero-filled span.  defer_zero: the caller zero-fills the span itself (the
        decoder backward folds it into its own multi-range fill launch) -- returns (views, span tensor)."""
        lo, hi = self.arena.span(self.slots)
        if self.arena.whole_zero_pass:
            # an earlier Function of THIS backward pass (the decoder, which runs first) zero-filled the whole arena: nothing to fill
            views = [self.arena.grad_view(s) for s in self.slots]
            return (views, None) if defer_zero else views
        if defer_zero and all(p.grad is None for p in self.arena.params):
            lo, hi = 0, self.arena.total      # fresh step: ONE fill covers every group's span and the never-used parameters' slots
            self.arena.whole_zero_pass = True  # valid until this backward pass ends (autograd end-of-pass callback)
            torch.autograd.Variable._execution_engine.queue_callback(self.arena.end_backward_pass)
        span = self.arena.flat_g[lo:hi]
        if not defer_zero:
            span.zero_()
        self.arena.note_zeroed(lo, hi)
        views = [self.arena.grad_view(s) for s in self.slots]
        return (views, span) if defer_zero else views

    def has_hooks(self):
        """True when a parameter of the group carries tensor hooks (DDP / FSDP style post-accumulate hooks, register_hook): they read the
        gradient inside the backward pass, so it must be final when the Function returns (no asynchronous tail)."""
        return any(getattr(p, '_backward_hooks', None) or getattr(p, '_post_accumulate_grad_hooks', None) for p in self.params)


# --------------------------------------------------------------------------------------------------
class EventPoolGather(torch.autograd.Function):
    """ech = [mean-pooled C3D rows | tap[ind]]  (CaptionGenerator.py:111-114,121,128); `parts`: 3 = both ('ER3'), 1 = the pooled rows only
    ('ER1'), 2 = the anchors' SST states only ('ER2')."""

    @staticmethod
    def forward(ctx, c3d, tap, ev_start, ev_len, ind, parts=3):
        lib = L.load()
        c3d, tap = _f32c(c3d), _f32c(tap)
        N = ev_start.numel()
        D, Ht = (c3d.shape[1] if parts & 1 else 0), (tap.shape[1] if parts & 2 else 0)
        ech = torch.empty(N, D + Ht, device=c3d.device, dtype=torch.float32)
        L.check(lib.echr_event_pool_gather_fwd(L.ptr(c3d), L.ptr(tap), L.ptr(ev_start, torch.int32), L.ptr(ev_len, torch.int32),
                                               L.ptr(ind, torch.int32), L.ptr(ech), N, D, Ht, L.stream_ptr()), 'event_pool_gather_fwd')
        ctx.save_for_backward(ind)
        ctx.shape = (tap.shape, D, Ht, N)
        return ech

    @staticmethod
    def backward(ctx, g_ech):
        lib = L.load()
        (ind,) = ctx.saved_tensors
        tap_shape, D, Ht, N = ctx.shape
        g_tap = None
        if ctx.needs_input_grad[1] and Ht > 0:
            g_tap = torch.zeros(tap_shape, device=g_ech.device, dtype=torch.float32)
            L.check(lib.echr_event_pool_gather_bwd(L.ptr(_f32c(g_ech)), L.ptr(ind, torch.int32), L.ptr(g_tap), N, D, Ht,
                                                   L.stream_ptr()), 'event_pool_gather_bwd')
        return None, g_tap, None, None, None, None


# --------------------------------------------------------------------------------------------------
TSRM_PARAMS = ('w_emb', 'b_emb', 'w_fc1', 'b_fc1', 'w_fc2', 'b_fc2', 'w_q', 'b_q', 'w_k', 'b_k', 'w_out', 'b_out')


class TSRMFunction(torch.autograd.Function):
    """MA_Attention8.forward (MA_attention_8_NEW.py:35-49, :101-177)."""

    @staticmethod
    def forward(ctx, ech, ev_start, ev_len, n_head, drop, sink, bounds, *params):
        # bounds: None or (inference, max_len, max_span[, fst_mode]) = echr_tsrm_args' last fields.  `inference` must be decided by the CALLER
        # (grad mode is always off inside forward, and needs_input_grad ignores torch.no_grad())
        lib = L.load()
        ctx.sink = sink
        ech = _f32c(ech)
        ps = [_f32c(p) for p in params]
        N, Din = ech.shape
        Df, Do = ps[0].shape[0], ps[10].shape[0]
        if ps[0].shape[1] != Din:
            raise L.EchrHipError('event features are %d wide, fusion_model.event_emb expects %d (video_dim + hidden_dim: CaptionGenerator.py:121-125)'
                                 % (Din, ps[0].shape[1]))
        ws = torch.empty(lib.echr_tsrm_ws_floats(N, Din, Df, Do, n_head), device=ech.device, dtype=torch.float32)
        out = torch.empty(N, Do, device=ech.device, dtype=torch.float32)
        a = L.TsrmArgs(N, Din, Df, Do, n_head, *[L.ptr(p) for p in ps], L.ptr(ech), L.ptr(ev_start, torch.int32),
                       L.ptr(ev_len, torch.int32), L.ptr(ws), L.ptr(out), *(bounds or (0, 0, 0)))
        d = drop.c()
        L.check(lib.echr_tsrm_fwd(C.byref(a), C.byref(d), L.stream_ptr()), 'tsrm_fwd')
        ctx.save_for_backward(ech, ev_start, ev_len, ws, out, *ps)
        ctx.meta = (N, Din, Df, Do, n_head, drop, int(a.fst_mode))
        return out

    @staticmethod
    def backward(ctx, g_out):
        lib = L.load()
        ech, ev_start, ev_len, ws, out, *ps = ctx.saved_tensors
        N, Din, Df, Do, G, drop, fst_mode = ctx.meta
        g_out = _f32c(g_out)
        zeroed = 1 if (ctx.sink is not None and ctx.sink.usable()) else 0
        grads = ctx.sink.take() if zeroed else [torch.empty_like(p) for p in ps]
        if zeroed:
            grads[10] = grads[10].view(ps[10].shape)              # linear_out_1.weight [d_o, d_feats, 1, 1] -> [d_o, d_feats]
        g_ech = torch.empty_like(ech)
        wsb = torch.empty(lib.echr_tsrm_ws_bwd_floats(N, Din, Df, Do, G), device=ech.device, dtype=torch.float32)
        a = L.TsrmArgs(N, Din, Df, Do, G, *[L.ptr(p) for p in ps], L.ptr(ech), L.ptr(ev_start, torch.int32),
                       L.ptr(ev_len, torch.int32), L.ptr(ws), L.ptr(out), 0, 0, 0, fst_mode)
        g = L.TsrmGrads(*[L.ptr(x) for x in grads], L.ptr(g_ech), L.ptr(g_out), L.ptr(wsb), zeroed)
        if not zeroed and fst_mode in (3, 4):          # parameters the chosen combination does not reach: the library writes nothing there
            for i in ((6, 7, 8, 9) if fst_mode == 3 else (2, 3, 4, 5)):
                grads[i].zero_()
        d = drop.c()
        L.check(lib.echr_tsrm_bwd(C.byref(a), C.byref(g), C.byref(d), L.stream_ptr()), 'tsrm_bwd')
        return (g_ech, None, None, None, None, None, None) + tuple(grads)


def position_embedding(ev_start, ev_len, d_pos):
    """[N,N,d_pos] fp32 pairwise position embedding generated on device (MA_attention_8_NEW.py:39-41)."""
    lib = L.load()
    N = ev_start.numel()
    pos = torch.empty(N, N, d_pos, device=ev_start.device, dtype=torch.float32)
    L.check(lib.echr_tsrm_posemb(L.ptr(ev_start, torch.int32), L.ptr(ev_len, torch.int32), L.ptr(pos), N, d_pos, L.stream_ptr()),
            'tsrm_posemb')
    return pos


# --------------------------------------------------------------------------------------------------
# parameter order of the decoder Functions
DEC_PARAMS = ('embed', 'w_logit', 'b_logit',
              'w_ih0', 'w_ih1', 'w_ih2', 'w_hh0', 'w_hh1', 'w_hh2', 'b_ih0', 'b_ih1', 'b_ih2', 'b_hh0', 'b_hh1', 'b_hh2',
              'w_c2a', 'b_c2a', 'w_h2a', 'b_h2a', 'w_alpha', 'b_alpha')


def _dec_args(ps, c3d, ev_start, ev_len, event, video, tokens, A, S, ws, logp, disjoint=False, n_de=None, prepared=0, train=0, h0=None):
    (embed, w_logit, b_logit, wi0, wi1, wi2, wh0, wh1, wh2, bi0, bi1, bi2, bh0, bh1, bh2, w_c2a, b_c2a, w_h2a, b_h2a,
     w_alpha, b_alpha) = ps
    N, De = event.shape if event is not None else n_de
    Tv, D = c3d.shape
    H = wh0.shape[1]
    E = embed.shape[1]
    Ha = w_c2a.shape[0]
    Dv = video.numel()
    V1 = embed.shape[0]
    assert wi0.shape[1] == E + De and wi1.shape[1] == E + D and wi2.shape[1] == E + Dv, 'LSTM input widths do not match the contexts'
    return L.DecArgs(N, A, Tv, D, H, E, Ha, De, Dv, V1, S, 1 if disjoint else 0,
                     L.ptr(embed), L.ptr(w_logit), L.ptr(b_logit),
                     L.ptr3((wi0, wi1, wi2), 'w_ih'), L.ptr3((wh0, wh1, wh2), 'w_hh'), L.ptr3((bi0, bi1, bi2), 'b_ih'),
                     L.ptr3((bh0, bh1, bh2), 'b_hh'),
                     L.ptr(w_c2a), L.ptr(b_c2a), L.ptr(w_h2a), L.ptr(b_h2a), L.ptr(w_alpha), L.ptr(b_alpha),
                     L.ptr(c3d), L.ptr(ev_start, torch.int32), L.ptr(ev_len, torch.int32), L.ptr(event), L.ptr(video),
                     L.ptr(tokens, torch.int32) if tokens is not None else None, L.ptr(ws), L.ptr(logp) if logp is not None else None,
                     int(prepared), int(train), None, 0, L.ptr(h0) if h0 is not None else None)


def decoder_prepare(video, c3d, ev_start, ev_len, tokens, A, disjoint, params):
    """Start the event-independent part of the decoder forward (echr_decoder_fwd_prepare) on the library's second stream, BEFORE the
    event encoder is launched on the current stream; returns the handle DecoderFunction.forward continues from."""
    lib = L.load()
    video, c3d = _f32c(video), _f32c(c3d)
    ps = [_f32c(p) for p in params]
    S, N = tokens.shape
    V1, E = ps[0].shape
    De = ps[3].shape[1] - E                       # layer0.weight_ih is [4H, E + event_context_dim]
    logp = torch.empty(N, S, V1, device=c3d.device, dtype=torch.float32)
    # echr_dec_args.train: a backward pass can follow (decided here: grad mode is off inside the Function's forward)
    train = 1 if (torch.is_grad_enabled() and any(p.requires_grad for p in params)) else 0
    a = _dec_args(ps, c3d, ev_start, ev_len, None, video, tokens, A, S, None, logp, disjoint, n_de=(N, De), train=train)
    ws = torch.empty(lib.echr_decoder_ws_floats(C.byref(a)), device=c3d.device, dtype=torch.float32)
    a.ws = L.ptr(ws)
    L.check(lib.echr_decoder_fwd_prepare(C.byref(a), L.stream_ptr()), 'decoder_fwd_prepare')
    return dict(ws=ws, logp=logp, video=video, c3d=c3d, ps=ps, tokens=tokens, A=A, disjoint=disjoint, train=train)


def decoder_prepare_cancel():
    """Drop a decoder_prepare() handle that no forward will consume: the current stream waits for the library's second stream, so the
    handle's buffers can go back to the allocator."""
    L.check(L.load().echr_decoder_fwd_prepare_cancel(L.stream_ptr()), 'decoder_fwd_prepare_cancel')


class InitState(torch.autograd.Function):
    """OldModel.init_hidden with CG_init_feats_type (OldModel_NEW.py:79-96): h0 [N, 3H] = init_linear(cat([video | event | clip.mean(1)])) -- the
    tensor the reference then views as (N, 3, H) and transposes; the decoder entry points take it in this layout (echr_dec_args.h0)."""

    @staticmethod
    def forward(ctx, video, event, c3d, ev_start, ev_len, A, use, sink, w, b):
        lib = L.load()
        use_v, use_e, use_c = use
        video, event, c3d, w, b = _f32c(video), _f32c(event), _f32c(c3d), _f32c(w), _f32c(b)
        N, De = event.shape
        Dv, D, H3, Dtot = video.numel(), c3d.shape[1], w.shape[0], w.shape[1]
        if Dtot != (Dv if use_v else 0) + (De if use_e else 0) + (D if use_c else 0):
            raise L.EchrHipError('init_linear is %d wide, the selected contexts give %d' % (Dtot, (Dv if use_v else 0) + (De if use_e else 0) + (D if use_c else 0)))
        feats = torch.empty(N, Dtot, device=event.device, dtype=torch.float32)
        h0 = torch.empty(N, H3, device=event.device, dtype=torch.float32)
        a = L.InitStateArgs(N, Dv, De, D, H3, int(use_v), int(use_e), int(use_c), int(A), L.ptr(video), L.ptr(event), L.ptr(c3d),
                            L.ptr(ev_start, torch.int32), L.ptr(ev_len, torch.int32), L.ptr(w), L.ptr(b), L.ptr(feats), L.ptr(h0))
        L.check(lib.echr_init_state_fwd(C.byref(a), L.stream_ptr()), 'init_state_fwd')
        ctx.save_for_backward(video, event, c3d, ev_start, ev_len, w, b, feats)
        ctx.meta = (A, use)
        ctx.sink = sink
        return h0

    @staticmethod
    def backward(ctx, g_h0):
        lib = L.load()
        video, event, c3d, ev_start, ev_len, w, b, feats = ctx.saved_tensors
        A, (use_v, use_e, use_c) = ctx.meta
        N, De = event.shape
        g_h0 = _f32c(g_h0)
        zeroed = 1 if (ctx.sink is not None and ctx.sink.usable()) else 0
        g_w, g_b = ctx.sink.take() if zeroed else (torch.empty_like(w), torch.empty_like(b))
        g_video = torch.empty_like(video) if (use_v and ctx.needs_input_grad[0]) else None
        g_event = torch.zeros_like(event) if (use_e and ctx.needs_input_grad[1]) else None
        dfeats = torch.empty_like(feats)
        a = L.InitStateArgs(N, video.numel(), De, c3d.shape[1], w.shape[0], int(use_v), int(use_e), int(use_c), int(A), L.ptr(video), L.ptr(event),
                            L.ptr(c3d), L.ptr(ev_start, torch.int32), L.ptr(ev_len, torch.int32), L.ptr(w), L.ptr(b), L.ptr(feats), None)
        g = L.InitStateGrads(L.ptr(g_h0), L.ptr(g_w), L.ptr(g_b), L.ptr(g_video) if g_video is not None else None,
                             L.ptr(g_event) if g_event is not None else None, L.ptr(dfeats), zeroed)
        L.check(lib.echr_init_state_bwd(C.byref(a), C.byref(g), L.stream_ptr()), 'init_state_bwd')
        return g_video, g_event, None, None, None, None, None, None, g_w, g_b


class ColMean(torch.autograd.Function):
    """x.mean(0) of a [T, D] feature matrix: the 'VC' / 'VH' scene contexts (CaptionGenerator.py:95-99)."""

    @staticmethod
    def forward(ctx, x):
        x = _f32c(x)
        out = torch.empty(x.shape[1], device=x.device, dtype=torch.float32)
        L.check(L.load().echr_col_mean_fwd(L.ptr(x), x.shape[0], x.shape[1], x.shape[1], L.ptr(out), L.stream_ptr()), 'col_mean_fwd')
        ctx.shape = tuple(x.shape)
        return out

    @staticmethod
    def backward(ctx, g):
        T, D = ctx.shape
        gx = torch.zeros(T, D, device=g.device, dtype=torch.float32)
        L.check(L.load().echr_col_mean_bwd(L.ptr(_f32c(g)), T, D, D, L.ptr(gx), L.stream_ptr()), 'col_mean_bwd')
        return gx


class DecoderFunction(torch.autograd.Function):
    """OldModel.forward with the ThreeStream core (OldModel_NEW.py:98-137, :376-401, :801-823): log-probs [N,S,V1]."""

    @staticmethod
    def forward(ctx, video, event, c3d, ev_start, ev_len, tokens, A, disjoint, drop, sink, prep, h0, *params):
        lib = L.load()
        ctx.sink = sink
        event = _f32c(event)
        h0 = _f32c(h0) if h0 is not None else None          # [N, 3H] initial state (OldModel.init_hidden, CG_init_feats_type); None = zeros
        S, N = tokens.shape
        if prep is not None:          # decoder_prepare() already ran the event-independent part on this workspace
            video, c3d, ps, logp, ws = prep['video'], prep['c3d'], prep['ps'], prep['logp'], prep['ws']
            train = prep['train']
            a = _dec_args(ps, c3d, ev_start, ev_len, event, video, tokens, A, S, ws, logp, disjoint, prepared=1, train=train, h0=h0)
        else:
            video, c3d = _f32c(video), _f32c(c3d)
            ps = [_f32c(p) for p in params]
            V1 = ps[0].shape[0]
            logp = torch.empty(N, S, V1, device=event.device, dtype=torch.float32)
            train = 1 if any(ctx.needs_input_grad) else 0
            a = _dec_args(ps, c3d, ev_start, ev_len, event, video, tokens, A, S, None, logp, disjoint, train=train, h0=h0)
            ws = torch.empty(lib.echr_decoder_ws_floats(C.byref(a)), device=event.device, dtype=torch.float32)
            a.ws = L.ptr(ws)
        d = drop.c()
        L.check(lib.echr_decoder_fwd(C.byref(a), C.byref(d), L.stream_ptr()), 'decoder_fwd')
        ctx.save_for_backward(video, event, c3d, ev_start, ev_len, tokens, ws, logp, *ps)
        ctx.h0 = h0
        ctx.meta = (A, S, drop, disjoint, train)
        return logp

    @staticmethod
    def backward(ctx, g_logp):
        lib = L.load()
        video, event, c3d, ev_start, ev_len, tokens, ws, logp, *ps = ctx.saved_tensors
        A, S, drop, disjoint, train = ctx.meta
        # criterion gradient left here in sparse form by MaskedNLL.backward (LanguageModelCriterion on this node's output)
        pend = ctx.__dict__.pop('_echr_pending_nll', None)     # one entry per LanguageModelCriterion applied to this node's output
        fused = None
        if pend:
            if len(pend) == 1 and getattr(g_logp, '_echr_nll_placeholder', False) and g_logp.stride() == (0, 0, 0):
                fused, g_logp = pend[0], None
            else:          # other consumers of the log-probs (or several criteria): their accumulated gradient plus every criterion's dense one
                for pe in pend:
                    g_logp = g_logp + MaskedNLL.dense_grad(pe[0], pe[1], pe[2], pe[3], *logp.shape)
        if g_logp is not None:
            g_logp = _f32c(g_logp)
        zeroed = 1 if (ctx.sink is not None and ctx.sink.usable()) else 0
        zero_span = None
        if zeroed:
            grads, zero_span = ctx.sink.take(defer_zero=True)      # zero-filled inside echr_decoder_bwd's first stage (one fill launch less)
        else:
            red = getattr(ctx.sink.arena, 'early_reducer', None) if ctx.sink is not None else None
            if red is not None:
                red.check_no_backward_while_in_flight()      # accumulating into ranges whose all-reduce already started
            grads = [torch.empty_like(p) for p in ps]
            grads[0].zero_()                                 # embedding table gradient is scatter-added
        g_event = torch.empty_like(event)
        g_video = torch.empty_like(video) if ctx.needs_input_grad[0] else None
        h0 = ctx.h0
        g_h0 = torch.empty_like(h0) if (h0 is not None and ctx.needs_input_grad[11]) else None
        a = _dec_args(ps, c3d, ev_start, ev_len, event, video, tokens, A, S, ws, logp, disjoint, train=train, h0=h0)
        wsb = torch.empty(lib.echr_decoder_ws_bwd_floats(C.byref(a)), device=event.device, dtype=torch.float32)
        gp = [L.ptr(x) for x in grads]
        g = L.DecGrads(gp[0], gp[1], gp[2], (L.c_f * 3)(*gp[3:6]), (L.c_f * 3)(*gp[6:9]), (L.c_f * 3)(*gp[9:12]),
                       (L.c_f * 3)(*gp[12:15]), gp[15], gp[16], gp[17], gp[18], gp[19], gp[20],
                       L.ptr(g_event), L.ptr(g_video), L.ptr(g_logp) if g_logp is not None else None,
                       L.ptr(fused[0], fused[0].dtype) if fused else None, L.ptr(fused[1]) if fused else None, L.ptr(fused[3]) if fused else None,
                       L.ptr(wsb), zeroed, 0, 0, L.ptr(fused[2][1:2]) if fused else None,
                       L.ptr(zero_span) if zero_span is not None else None, zero_span.numel() if zero_span is not None else 0,
                       1 if (fused and fused[0].dtype == torch.int64) else 0, 0, None, 0, L.ptr(g_h0) if g_h0 is not None else None)
        d = drop.c()
        hook = getattr(ctx.sink.arena, 'early_grad_hook', None) if zeroed else None
        staged = hook is not None and getattr(ctx.sink.arena, 'early_staged', True)
        if staged:
            # data parallel: hand gradients to the reducer as soon as they are final; it starts their all-reduce on the collective
            # stream while the next stage runs on this one.  params order = OldModel.native_params():
            #   [0] embed, [1] logit.weight, [2] logit.bias, [3:6] weight_ih, [6:9] weight_hh, [9:12] bias_ih, [12:15] bias_hh, ...
            sp = ctx.sink.params
            for phase, ready in ((1, [sp[1], sp[2]]), (3, list(sp[3:15])), (4, None)):
                g.phase = phase
                L.check(lib.echr_decoder_bwd(C.byref(a), C.byref(g), C.byref(d), L.stream_ptr()), 'decoder_bwd')
                if ready is not None:
                    hook(ready)
        else:
            # arena path: the returned parameter gradients are views that autograd adopts without touching them, so the last stage of
            # the backward (attention-parameter / embedding gradients) may still be running on the library's second stream while
            # autograd goes on with the event encoder's backward; an end-of-backward callback joins the streams and only then lets go
            # of the workspaces that stage reads
            # (not when a parameter hook would read those gradients inside the backward pass, nor under create_graph, where autograd may
            # clone them: the gradients must then be final when this Function returns)
            g.async_tail = 1 if (zeroed and ASYNC_TAIL[0] and not ctx.sink.has_hooks() and not torch.is_grad_enabled()) else 0
            if g.async_tail and hook is None and ASYNC_LEVEL[0] == 2:
                # no data-parallel hand-over waits for the LSTM-layer gradients: only d event is formed on this stream, the rest of that stage
                # runs on the library's second helper stream and is joined with the tail by the end-of-backward callback
                g.async_tail = 2
            L.check(lib.echr_decoder_bwd(C.byref(a), C.byref(g), C.byref(d), L.stream_ptr()), 'decoder_bwd')
            if hook is not None:
                # data parallel, one-call form: the LSTM-layer gradients (and everything else part A of the backward produced) are final in
                # stream order now; the reducer starts their all-reduce, which overlaps the asynchronous tail and the event encoder's backward
                hook(list(ctx.sink.params[3:15]), after_recurrence=True)
            if g.async_tail:
                keep = [ws, wsb, logp, c3d, tokens, ev_start, ev_len, g_logp, fused]
                sp = L.stream_ptr()

                def _join(keep=keep, sp=sp):
                    L.check(lib.echr_stream_join(sp), 'stream_join')
                    del keep[:]
                torch.autograd.Variable._execution_engine.queue_callback(_join)
        return (g_video, g_event, None, None, None, None, None, None, None, None, None, g_h0) + tuple(grads)


def greedy_sample(video, event, c3d, ev_start, ev_len, A, seq_length, params, debug=None, multinomial=False, temperature=1.0, seed=0,
                  table_cache=None, h0=None):
    """OldModel.sample (OldModel_NEW.py:139-187) with every step on device; one host sync at the end.  Greedy arg-max by default
    (sample_max = 1); multinomial=True draws each token from softmax(logp / temperature) (:160-168) with the library's Philox stream
    keyed by `seed`.

    `table_cache`: a dict owned by the caller (one per model); the persistent decoder's parameter-only operands (token-side gate tables,
    logit-weight image) are kept in it and reused while the parameters are unchanged (data pointers, torch version counters and the
    library's own PARAM_EPOCH).

    Returns (seq int64 [N,T], logp fp32 [N,T]) with T <= seq_length, or ([], []) when nothing was generated."""
    lib = L.load()
    video, event, c3d = _f32c(video), _f32c(event), _f32c(c3d)
    ps = [_f32c(p) for p in params]
    N = event.shape[0]
    dev = event.device
    h0 = _f32c(h0) if h0 is not None else None
    a = _dec_args(ps, c3d, ev_start, ev_len, event, video, None, A, seq_length, None, None, h0=h0)
    ws = torch.empty(lib.echr_decoder_ws_floats(C.byref(a)), device=dev, dtype=torch.float32)
    a.ws = L.ptr(ws)
    wss = torch.empty(lib.echr_sampler_ws_floats(C.byref(a)), device=dev, dtype=torch.float32)
    seq = torch.empty(N, seq_length, device=dev, dtype=torch.int64)
    slp = torch.empty(N, seq_length, device=dev, dtype=torch.float32)
    nun = torch.empty(seq_length + 1, device=dev, dtype=torch.int32)
    tables, valid = None, 0
    if table_cache is not None and not multinomial:
        nt = lib.echr_sampler_table_floats(C.byref(a))
        if nt > 0:
            key = (PARAM_EPOCH[0], nt, str(dev)) + tuple((p.data_ptr(), p._version) for p in params)
            tables = table_cache.get('tables')
            if tables is None or tables.numel() != nt or tables.device != dev:
                tables = table_cache['tables'] = torch.empty(nt, device=dev, dtype=torch.float32)
                table_cache['key'] = None
            valid = 1 if table_cache.get('key') == key else 0
            table_cache['key'] = None      # marked valid again only once this call has completed without an error
    sa = L.SampleArgs(a, seq_length, L.ptr(seq, torch.int64), L.ptr(slp), L.ptr(nun, torch.int32), L.ptr(wss),
                      1 if multinomial else 0, float(temperature), int(seed) & 0xFFFFFFFFFFFFFFFF,
                      L.ptr(tables) if tables is not None else None, valid)
    L.check(lib.echr_decoder_sample(C.byref(sa), L.stream_ptr()), 'decoder_sample')
    counts = nun.cpu().numpy()                 # the only device->host sync of the whole decode
    L.check(lib.echr_check_async(), 'decoder_sample')
    if tables is not None:
        table_cache['key'] = key
    if debug is not None:                      # tests: the raw logits [N,V1] of the last decoder step (sampler workspace: XT | LOGITS | ...)
        E, V1 = ps[0].shape[1], ps[0].shape[0]
        o = (N * E + 63) // 64 * 64
        debug['last_logits'] = wss[o:o + N * V1].view(N, V1).clone()
        debug['seq_full'], debug['logp_full'] = seq.clone(), slp.clone()
        debug['stopped_early'] = int(counts[0])      # persistent decoder: 1 = every event had emitted <eos> before seq_length and the launch stopped there
    T = seq_length
    for t in range(1, seq_length + 1):        # OldModel_NEW.py:179-180: stop at the first step with nobody unfinished
        if counts[t] == 0:
            T = t - 1
            break
    if T == 0:
        return [], []
    return seq[:, :T].contiguous(), slp[:, :T].contiguous()


def decoder_step(it, video, event, c3d, ev_start, ev_len, A, state, params, drop=None):
    """OldModel.get_logprobs_state (OldModel_NEW.py:133-137): ONE timestep, state in / state out.

    it int [N] tokens; state = (h [3,N,H] dropped outputs, c [3,N,H]).  Returns (log-probs [N,V1], (h', c')).  Forward only."""
    lib = L.load()
    video, event, c3d = _f32c(video), _f32c(event), _f32c(c3d)
    ps = [_f32c(p) for p in params]
    N = event.shape[0]
    dev = event.device
    V1 = ps[0].shape[0]
    tok = it.to(device=dev, dtype=torch.int32).contiguous()
    h_in, c_in = _f32c(state[0]), _f32c(state[1])
    if tuple(h_in.shape) != tuple(c_in.shape) or h_in.shape[0] != 3 or h_in.shape[1] != N:
        raise ValueError('state must be a pair of [3,N,H] tensors (got %s, %s)' % (tuple(h_in.shape), tuple(c_in.shape)))
    logp = torch.empty(N, V1, device=dev, dtype=torch.float32)
    a = _dec_args(ps, c3d, ev_start, ev_len, event, video, tok, A, 1, None, logp)
    ws = torch.empty(lib.echr_decoder_ws_floats(C.byref(a)), device=dev, dtype=torch.float32)
    a.ws = L.ptr(ws)
    h_out, c_out = torch.empty_like(h_in), torch.empty_like(c_in)
    d = (drop if drop is not None else DropState(training=False)).c()
    L.check(lib.echr_decoder_step(C.byref(a), L.ptr(h_in), L.ptr(c_in), L.ptr(h_out), L.ptr(c_out), C.byref(d), L.stream_ptr()), 'decoder_step')
    return logp, (h_out, c_out)


def tsrm_attention(roi_feat, position_embedding, n_head, params, d_o, drop=None, fst_mode=0):
    """attention_module_multi_head.forward (MA_attention_8_NEW.py:101-177) on the embedded events roi_feat [N,Df] and the pairwise
    position embedding [N,N,Df].  params = (fc1.w, fc1.b, fc2.w, fc2.b, q.w, q.b, k.w, k.b, out.w [Do,Df], out.b).  Forward only."""
    lib = L.load()
    x, pos = _f32c(roi_feat), _f32c(position_embedding)
    ps = [_f32c(p) for p in params]
    N, Df = x.shape
    if tuple(pos.shape) != (N, N, Df):
        raise ValueError('position_embedding must be [N,N,%d] (got %s)' % (Df, tuple(pos.shape)))
    Din = Df
    ws = torch.empty(lib.echr_tsrm_ws_floats(N, Din, Df, d_o, n_head), device=x.device, dtype=torch.float32)
    out = torch.empty(N, d_o, device=x.device, dtype=torch.float32)
    a = L.TsrmArgs(N, Din, Df, d_o, n_head, None, None, *[L.ptr(p) for p in ps], None, None, None, L.ptr(ws), L.ptr(out), 0, 0, 0, fst_mode)
    d = (drop if drop is not None else DropState(training=False)).c()
    L.check(lib.echr_tsrm_attn_fwd(C.byref(a), L.ptr(x), L.ptr(pos), C.byref(d), L.stream_ptr()), 'tsrm_attn_fwd')
    return out


# --------------------------------------------------------------------------------------------------
_ZERO_PLACEHOLDER = {}


def _nll_target(target, S):
    """Targets for the criterion kernels: int64 (the reference's LongTensor labels) and int32 are both read in place."""
    t = target[:, :S]
    if t.dtype not in (torch.int64, torch.int32):
        t = t.to(torch.int64)
    return t.contiguous()


class MaskedNLL(torch.autograd.Function):
    """LanguageModelCriterion.forward (misc/utils.py:66-75) on device."""

    @staticmethod
    def forward(ctx, logp, target, mask, node=None):
        lib = L.load()
        ctx.node = node
        N, S, V1 = logp.shape
        logp = logp.contiguous()
        tgt = _nll_target(target, S)
        msk = mask[:, :S].to(torch.float32).contiguous()
        out = torch.empty(2, device=logp.device, dtype=torch.float32)
        fn = lib.echr_nll_loss_fwd_i64 if tgt.dtype == torch.int64 else lib.echr_nll_loss_fwd
        L.check(fn(L.ptr(logp), L.ptr(tgt, tgt.dtype), L.ptr(msk), L.ptr(out), N, S, V1, L.stream_ptr()), 'nll_loss_fwd')
        ctx.save_for_backward(tgt, msk, out)
        ctx.shape = (N, S, V1)
        return out[0]

    @staticmethod
    def backward(ctx, g):
        tgt, msk, out = ctx.saved_tensors
        N, S, V1 = ctx.shape
        if ctx.node is not None:
            # the log-probs came from DecoderFunction: leave the criterion's gradient with that node in its sparse form (targets, mask,
            # upstream scalar) and hand autograd a stride-0 all-zero placeholder.  DecoderFunction.backward takes the fused path when the
            # placeholder arrives untouched, and adds the dense form when other consumers of the log-probs contributed gradients too.
            ctx.node.__dict__.setdefault('_echr_pending_nll', []).append((tgt, msk, out, _f32c(g).reshape(1)))
            z = _ZERO_PLACEHOLDER.get(msk.device)
            if z is None:                         # one zero scalar per device, filled once (never written: only ever expanded)
                z = _ZERO_PLACEHOLDER[msk.device] = torch.zeros((), device=msk.device, dtype=torch.float32)
            ph = z.expand(N, S, V1)
            ph._echr_nll_placeholder = True
            return ph, None, None, None
        return MaskedNLL.dense_grad(tgt, msk, out, g, N, S, V1), None, None, None

    @staticmethod
    def dense_grad(tgt, msk, out, g, N, S, V1):
        lib = L.load()
        g_logp = torch.empty(N, S, V1, device=msk.device, dtype=torch.float32)
        fn = lib.echr_nll_loss_bwd_i64 if tgt.dtype == torch.int64 else lib.echr_nll_loss_bwd
        L.check(fn(L.ptr(tgt, tgt.dtype), L.ptr(msk), L.ptr(out), L.ptr(_f32c(g).reshape(1)), L.ptr(g_logp), N, S, V1,
                   L.stream_ptr()), 'nll_loss_bwd')
        return g_logp


# Bumped by every parameter update the library performs through raw pointers (clamp_adam_: torch's per-tensor version counters do not see
# those writes); part of the key of caches of parameter-derived operands (OldModel.sample's decoding tables).
PARAM_EPOCH = [0]


def clamp_adam_(p, g, m, v, step, lr, beta1=0.9, beta2=0.999, eps=1e-8, clip=100.0, applied=None):
    """In-place fused clamp(+-clip) + Adam over flat fp32 buffers (misc/utils.py:107-111 + optim.Adam).  `applied`: optional int32 device
    tensor [1] the launch increments iff the update was applied (echr_clamp_adam_counted)."""
    lib = L.load()
    PARAM_EPOCH[0] += 1
    L.check(lib.echr_clamp_adam_counted(L.ptr(p), L.ptr(g), L.ptr(m), L.ptr(v), p.numel(), int(step), float(lr), float(beta1),
                                        float(beta2), float(eps), float(clip), None if applied is None else L.ptr(applied, torch.int32),
                                        L.stream_ptr()), 'clamp_adam')


def clamp_(g, clip):
    lib = L.load()
    L.check(lib.echr_clamp(L.ptr(g), g.numel(), float(clip), L.stream_ptr()), 'clamp')
    return g


def gemm(A, B, trans_b=True, bias=None, algo=0):
    """C = A @ B^T (trans_b) or A @ B on the fp32 MFMA path; exposed for kernel-level parity tests."""
    lib = L.load()
    M, K = A.shape
    Nn = B.shape[0] if trans_b else B.shape[1]
    Cc = torch.empty(M, Nn, device=A.device, dtype=torch.float32)
    d = L.GemmDesc()
    d.A, d.B, d.C = L.ptr(A), L.ptr(B), L.ptr(Cc)
    d.M, d.N, d.K = M, Nn, K
    d.sam, d.sak = K, 1
    d.sbk, d.sbn = (1, K) if trans_b else (Nn, 1)
    d.ldc = Nn
    d.batch, d.alpha, d.beta, d.split_k, d.algo = 1, 1.0, 0.0, -1, algo
    d.bias = L.ptr(bias) if bias is not None else None
    L.check(lib.echr_gemm_f32(C.byref(d), L.stream_ptr()), 'gemm_f32')
    return Cc


# --------------------------------------------------------------------------------------------------
class SSTFunction(torch.autograd.Function):
    """SST.forward (models/sst_model.py:31-40): 2-layer LSTM over one video + sigmoid proposal head, native."""

    @staticmethod
    def forward(ctx, x, p_drop, drop, sink, *params):
        lib = L.load()
        ctx.sink = sink
        x = _f32c(x)
        ps = [_f32c(p) for p in params]           # w_ih0, w_hh0, b_ih0, b_hh0, w_ih1, w_hh1, b_ih1, b_hh1, w_sc, b_sc
        T, D = x.shape
        H, K = ps[1].shape[1], ps[8].shape[0]
        dev = x.device
        ws = torch.empty(lib.echr_sst_ws_floats(T, D, H, K), device=dev, dtype=torch.float32)
        tap = torch.empty(T, H, device=dev, dtype=torch.float32)
        scores = torch.empty(T, K, device=dev, dtype=torch.float32)
        a = SSTFunction._args(ps, x, p_drop, ws, tap, scores)
        d = drop.c()
        L.check(lib.echr_sst_fwd(C.byref(a), C.byref(d), L.stream_ptr()), 'sst_fwd')
        ctx.save_for_backward(x, ws, tap, scores, *ps)
        ctx.meta = (p_drop, drop)
        return tap, scores

    @staticmethod
    def _args(ps, x, p_drop, ws, tap, scores):
        T, D = x.shape
        H, K = ps[1].shape[1], ps[8].shape[0]
        two = lambda a, b: (L.c_f * 2)(L.ptr(a), L.ptr(b))
        return L.SstArgs(T, D, H, K, float(p_drop), two(ps[0], ps[4]), two(ps[1], ps[5]), two(ps[2], ps[6]), two(ps[3], ps[7]),
                         L.ptr(ps[8]), L.ptr(ps[9]), L.ptr(x), L.ptr(ws), L.ptr(tap), L.ptr(scores))

    @staticmethod
    def backward(ctx, g_tap, g_scores):
        lib = L.load()
        x, ws, tap, scores, *ps = ctx.saved_tensors
        p_drop, drop = ctx.meta
        T, D = x.shape
        H, K = ps[1].shape[1], ps[8].shape[0]
        # flat arena (no gradient live yet): ONE zero fill of the span, then every product accumulates (`zeroed`): the split-K weight-gradient
        # products would otherwise zero-fill their own outputs, one launch each
        use_arena = ctx.sink is not None and ctx.sink.usable()
        grads = ctx.sink.take() if use_arena else [torch.empty_like(p) for p in ps]
        wsb = torch.empty(lib.echr_sst_ws_bwd_floats(T, D, H, K), device=x.device, dtype=torch.float32)
        a = SSTFunction._args(ps, x, p_drop, ws, tap, scores)
        two = lambda a_, b_: (L.c_f * 2)(L.ptr(a_), L.ptr(b_))
        g = L.SstGrads(two(grads[0], grads[4]), two(grads[1], grads[5]), two(grads[2], grads[6]), two(grads[3], grads[7]),
                       L.ptr(grads[8]), L.ptr(grads[9]), L.ptr(_f32c(g_tap)) if g_tap is not None else None,
                       L.ptr(_f32c(g_scores)) if g_scores is not None else None, L.ptr(wsb), 1 if use_arena else 0)
        d = drop.c()
        L.check(lib.echr_sst_bwd(C.byref(a), C.byref(g), C.byref(d), L.stream_ptr()), 'sst_bwd')
        return (None, None, None, None) + tuple(grads)


class TapBCE(torch.autograd.Function):
    """TAPModelCriterion.forward (misc/utils.py:78-99) on device."""

    @staticmethod
    def forward(ctx, scores, masks, labels, w1):
        lib = L.load()
        scores, masks, labels, w1 = _f32c(scores), _f32c(masks), _f32c(labels), _f32c(w1).reshape(-1)
        T, K = scores.shape
        buf = torch.empty(65, device=scores.device, dtype=torch.float32)          # loss | 64 partial sums
        loss = buf[:1]
        L.check(lib.echr_tap_bce_fwd_ws(L.ptr(scores), L.ptr(masks), L.ptr(labels), L.ptr(w1), L.ptr(loss), L.ptr(buf[1:]), T, K, L.stream_ptr()),
                'tap_bce_fwd')
        ctx.save_for_backward(scores, masks, labels, w1)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        lib = L.load()
        scores, masks, labels, w1 = ctx.saved_tensors
        T, K = scores.shape
        gs = torch.empty_like(scores)
        L.check(lib.echr_tap_bce_bwd(L.ptr(scores), L.ptr(masks), L.ptr(labels), L.ptr(w1), L.ptr(_f32c(g).reshape(1)), L.ptr(gs), T, K,
                                     L.stream_ptr()), 'tap_bce_bwd')
        return gs, None, None, None


def h2_pack(x, transposed=False):
    """h2-packed image (two block-scaled fp16 planes, include/echr_hip.h) of the operand x [R,K] -- or, with transposed=True,
    of x^T for x stored [K,R] (the pack transposes on the fly).  Returns (uint8 buffer, R, K)."""
    lib = L.load()
    x = _f32c(x)
    if transposed:
        K, R = x.shape
        s_row, s_col = 1, x.stride(0)
    else:
        R, K = x.shape
        s_row, s_col = x.stride(0), 1
    buf = torch.empty(int(lib.echr_h2_bytes(R, K)), device=x.device, dtype=torch.uint8)
    L.check(lib.echr_h2_pack(L.ptr(x), R, K, s_row, s_col, buf.data_ptr(), L.stream_ptr()), 'h2_pack')
    return buf, R, K


def gemm_h2(a_packed, b_packed, bias=None):
    """C[M,N] = A . B^T (+ bias) from two h2-packed operands."""
    lib = L.load()
    (ab, M, K), (bb, N, K2) = a_packed, b_packed
    if K != K2:
        raise ValueError('contraction lengths differ (%d vs %d)' % (K, K2))
    out = torch.empty(M, N, device=ab.device, dtype=torch.float32)
    d = L.GemmDesc()
    d.A, d.B, d.C = ab.data_ptr(), bb.data_ptr(), out.data_ptr()
    d.M, d.N, d.K = M, N, K
    d.sam, d.sak, d.sbk, d.sbn = K, 1, 1, K
    d.ldc, d.batch, d.alpha, d.beta, d.split_k, d.algo = N, 1, 1.0, 0.0, -1, 2
    if bias is not None:
        d.bias = L.ptr(_f32c(bias))
    L.check(lib.echr_gemm_f32(C.byref(d), L.stream_ptr()), 'gemm_h2')
    return out
