"""One training iteration of the caption path as ONE library call (echr_train_step, include/echr_hip.h).

Reference protocol (train.py:281-317, m_batch = 1):
    optimizer.zero_grad(); pred = cg_model(tap_feats, c3d_feats, lda_feats, labels, ind, soi, mode='train')
    loss = crit(pred, labels[:, 1:], masks[:, 1:]); loss.backward(); clip_gradient(optimizer, c); optimizer.step()
`FusedTrainStep(model, optimizer)(...)` is that sequence without autograd: Python fills one argument struct (pointers are stable: flat
parameter / gradient arena, one persistent workspace), packs the index vectors, and makes ONE ctypes call; the library sequences the
same entry points the autograd Functions of echr_amd/functional.py call.  The autograd path stays the general one (gradient
accumulation, other consumers of the log-probs, joint training of the proposal encoder, hooks); both produce the same update
(tests/test_gpu_parity.py::test_fused_train_step_equals_autograd_path).
"""
import ctypes as C
import math

import numpy as np
import torch

from . import _lib as L
from . import functional as EF
from .models.OldModel_NEW import n_decoder_steps
from .optim import ClampAdam


import os
MASKED_ROWS = [os.environ.get('ECHR_MASKED_ROWS', '1') != '0']          # late-fusion stage on the rows with a non-zero criterion mask only


class FusedTrainStep(object):
    LOSS_SLOTS = 16

    def __init__(self, model, optimizer, grad_clip=None):
        if not isinstance(optimizer, ClampAdam) or optimizer.arena is None:
            raise ValueError('FusedTrainStep needs echr_amd.optim.ClampAdam built with the flat arena (model.build_arena())')
        arena = optimizer.arena
        if getattr(model, '_echr_arena', None) is not arena or not arena.params_in_arena():
            raise ValueError('the optimiser\'s arena is not the model\'s (call model.build_arena() after .cuda(), pass it to ClampAdam)')
        if not hasattr(model, 'fusion_model') or model.opt.event_context_type not in ('ER1', 'ER2', 'ER3'):
            raise NotImplementedError('the fused step needs the TSRM event encoder (event_context_type ER1 / ER2 / ER3)')
        if len(optimizer.param_groups) != 1 or {id(p) for p in optimizer.param_groups[0]['params']} != {id(p) for p in arena.params}:
            raise ValueError('the optimiser must hold exactly the model\'s parameters in one group')
        self.model, self.optim, self.arena = model, optimizer, arena
        self.grad_clip = grad_clip if grad_clip is not None else optimizer.grad_clip
        self.lib = L.load()
        # (the helper streams now rather than at the first iteration: their hardware queues then do not depend on what creates streams in between.
        # A process that brings up an RCCL communicator should call echr_streams_init() BEFORE init_process_group -- INTEGRATION.md)
        with torch.cuda.device(arena.flat_p.device):
            L.check(self.lib.echr_streams_init(), 'streams_init')
        self.dev = arena.flat_p.device
        self.a = L.TrainStepArgs()
        self.ws = None
        self.one = torch.ones(1, device=self.dev, dtype=torch.float32)
        self.loss_ring = torch.zeros(self.LOSS_SLOTS, 2, device=self.dev, dtype=torch.float32)
        self.calls = 0
        self._fill_static()

    def join(self):
        """Make the current stream wait for a deferred update (defer_update=True); no-op otherwise."""
        L.check(self.lib.echr_stream_join(L.stream_ptr()), 'stream_join')

    # ---- pointers that never change: parameters and their gradient slots ------------------------------------------------------
    def _fill_static(self):
        a, m, ar = self.a, self.model, self.arena
        gp = lambda p: ar.flat_g.data_ptr() + 4 * ar.offsets[ar.slot(p)]
        lm, fm = m.lm_model, m.fusion_model
        tp = fm.native_params()
        tsrm_params = (fm.event_emb.weight, fm.event_emb.bias, fm.enc_attn.pair_pos_fc1.weight, fm.enc_attn.pair_pos_fc1.bias,
                       fm.enc_attn.pair_pos_fc2.weight, fm.enc_attn.pair_pos_fc2.bias, fm.enc_attn.query_1.weight, fm.enc_attn.query_1.bias,
                       fm.enc_attn.key_1.weight, fm.enc_attn.key_1.bias, fm.enc_attn.linear_out_1.weight, fm.enc_attn.linear_out_1.bias)
        for name, p, v in zip(EF.TSRM_PARAMS, tsrm_params, tp):
            setattr(a.tsrm, name, L.ptr(v))
            setattr(a.tsrm_g, 'g_' + name, gp(p))
        a.tsrm.Din, a.tsrm.Df, a.tsrm.Do, a.tsrm.G = tp[0].shape[1], tp[0].shape[0], tp[10].shape[0], fm.enc_attn.group
        a.tsrm.fst_mode = fm.fst_mode()
        ps = lm.native_params()
        (embed, w_logit, b_logit, wi0, wi1, wi2, wh0, wh1, wh2, bi0, bi1, bi2, bh0, bh1, bh2, w_c2a, b_c2a, w_h2a, b_h2a, w_alpha, b_alpha) = ps
        d, g = a.dec, a.dec_g
        d.embed, d.w_logit, d.b_logit = L.ptr(embed), L.ptr(w_logit), L.ptr(b_logit)
        d.w_ih, d.w_hh = L.ptr3((wi0, wi1, wi2), 'w_ih'), L.ptr3((wh0, wh1, wh2), 'w_hh')
        d.b_ih, d.b_hh = L.ptr3((bi0, bi1, bi2), 'b_ih'), L.ptr3((bh0, bh1, bh2), 'b_hh')
        d.w_c2a, d.b_c2a, d.w_h2a, d.b_h2a, d.w_alpha, d.b_alpha = (L.ptr(x) for x in (w_c2a, b_c2a, w_h2a, b_h2a, w_alpha, b_alpha))
        g.g_embed, g.g_w_logit, g.g_b_logit = gp(embed), gp(w_logit), gp(b_logit)
        g.g_w_ih = (L.c_f * 3)(gp(wi0), gp(wi1), gp(wi2))
        g.g_w_hh = (L.c_f * 3)(gp(wh0), gp(wh1), gp(wh2))
        g.g_b_ih = (L.c_f * 3)(gp(bi0), gp(bi1), gp(bi2))
        g.g_b_hh = (L.c_f * 3)(gp(bh0), gp(bh1), gp(bh2))
        g.g_w_c2a, g.g_b_c2a, g.g_w_h2a, g.g_b_h2a, g.g_w_alpha, g.g_b_alpha = (gp(x) for x in (w_c2a, b_c2a, w_h2a, b_h2a, w_alpha, b_alpha))
        d.H, d.E, d.Ha, d.V1 = wh0.shape[1], embed.shape[1], w_c2a.shape[0], embed.shape[0]
        d.D = w_c2a.shape[1]
        d.De, d.Dv = wi0.shape[1] - d.E, wi2.shape[1] - d.E
        if wi1.shape[1] != d.E + d.D or a.tsrm.Do != d.De:
            raise ValueError('LSTM input widths do not match the contexts')
        a.flat_g, a.n_flat, a.flat_p = ar.flat_g.data_ptr(), ar.total, ar.flat_p.data_ptr()
        a.g_loss = self.one.data_ptr()
        # the reference's non-recipe options (CaptionGenerator.py:106-130 event_context_type; OldModel_NEW.py:72-96 CG_init_feats_type)
        a.event_parts = {'ER1': 1, 'ER2': 2, 'ER3': 3}[m.opt.event_context_type]
        if getattr(lm, 'CG_init_feats_dim', 0):
            t = lm.CG_init_feats_type
            a.w_init, a.b_init = L.ptr(lm.init_linear.weight), L.ptr(lm.init_linear.bias)
            a.g_w_init, a.g_b_init = gp(lm.init_linear.weight), gp(lm.init_linear.bias)
            a.init_use_v, a.init_use_e, a.init_use_c = int('V' in t), int('E' in t), int('C' in t)
        else:
            a.w_init = a.b_init = a.g_w_init = a.g_b_init = None
            a.init_use_v = a.init_use_e = a.init_use_c = 0
        a.vh_offset, a.tap_rows = -1, 0
        self._keep = (tp, ps)          # (fusion_model.native_params() builds a view of linear_out_1.weight: keep it alive)
        self._epoch_ptrs = (ar.flat_p.data_ptr(), ar.flat_g.data_ptr())

    def _flat_state(self):
        o = self.optim
        if o._flat is None:
            if any(o.state[p] for p in self.arena.params):
                raise RuntimeError('per-tensor optimiser state exists (resumed run on the per-tensor path): load it with ClampAdam.load_state_dict '
                                   'on an arena optimiser, or use the autograd path')
            o._flat = dict(step=0, m=torch.zeros_like(self.arena.flat_p), v=torch.zeros_like(self.arena.flat_p))
        return o._flat

    def prepare(self, c3d_feats, lda_feats, lm_labels, ind_select_list, soi_select_list, targets, masks):
        """Joint 'tap_cg' iteration (train.py:300-313), optional first half: everything of the iteration that does not read tap_feats (index
        staging, the decoder's event-independent part, the gradient-arena fill) starts on the library's prepare stream and runs beside the
        proposal encoder's forward queued next.  Follow with `self(tap_feats, <the same arguments>, prepared=True, ...)`."""
        self._setup(None, c3d_feats, lda_feats, lm_labels, ind_select_list, soi_select_list, targets, masks, True, False, None, False)
        L.check(self.lib.echr_train_step_prepare(C.byref(self.a), L.stream_ptr()), 'train_step_prepare')
        self._prepared = True

    def cancel_prepare(self):
        """Abandon a prepare() whose second half will not follow (the caller's code between the two raised): the prepare stream's work is
        ordered before the workspace can be reused and the object accepts ordinary calls again."""
        if getattr(self, '_prepared', False):
            self._prepared = False
            L.check(self.lib.echr_decoder_fwd_prepare_cancel(L.stream_ptr()), 'decoder_fwd_prepare_cancel')

    def __call__(self, tap_feats, c3d_feats, lda_feats, lm_labels, ind_select_list, soi_select_list, targets, masks, step=True, forward_only=False,
                 tap_grad=None, defer_update=False, prepared=False, handover=False, handover_cb=None, mid_cb=None):
        """One iteration; returns the loss as a 0-d device tensor (no host sync).  `targets` / `masks`: what the reference hands its
        criterion (labels[:, 1:], masks[:, 1:]), host or device tensors.  step=False stops after the backward pass and exposes the
        gradients as `.grad` views of the arena (data-parallel reduce, inspection); the caller then steps the optimiser itself.
        `tap_grad`: a zero-filled float32 device tensor shaped like `tap_feats` that receives d loss / d tap_feats (added in place) -- the
        joint 'tap_cg' iteration of train.py:300-313 backpropagates it into the proposal encoder together with its own loss:
        `torch.autograd.backward([tap_loss, tap_feats], [None, tap_grad])`.  `defer_update=True` (with tap_grad and step): the call returns
        once tap_grad and the loss are final in stream order; the parameter gradients and the Adam update finish on the library's helper
        streams beside the proposal encoder's backward.  The next call joins by itself; call `join()` before touching the model's parameters
        in any other way (saving, evaluating, the autograd path).  `prepared=True`: `prepare()` ran with the same arguments.
        `handover=True` (with step=False): the backward pass records the data-parallel hand-over points (echr_handover_wait; DataParallelStep);
        `handover_cb(which, stream_ptr)`: called on the host from inside the call at each point (echr_train_step_args.handover_cb).
        `mid_cb()` (joint form): called on the host from inside the call right behind the work that leads to tap_grad, before the
        parameter-gradient tail is forked (echr_train_step_args.mid_cb; JointTrainStep queues the proposal encoder's backward there);
        `self.mid_called` says whether the library took that form."""
        a, lib = self.a, self.lib
        # a persistent launch of an EARLIER iteration gave up: the optimiser kernels queued behind it skipped their updates (parameters and
        # moments untouched) while the step was already counted -- L.check lets every optimiser wind its count back to the updates its own
        # device word says were applied (ClampAdam._on_async_abort), here and at every other site that can surface the -62
        L.check(lib.echr_check_async(), 'train_step (asynchronous failure of an earlier call)')
        if prepared:
            if not getattr(self, '_prepared', False) or not step or forward_only:
                raise RuntimeError('prepared=True needs a preceding prepare() and a full training step')
            self._prepared = False
            self._set_tap(EF._f32c(tap_feats), tap_grad, defer_update, True, False)
            a.prepared = 1
            slot, st = self._slot, self._state
        else:
            if getattr(self, '_prepared', False):
                raise RuntimeError('prepare() must be followed by a call with prepared=True')
            slot, st = self._setup(tap_feats, c3d_feats, lda_feats, lm_labels, ind_select_list, soi_select_list, targets, masks, step, forward_only,
                                   tap_grad, defer_update)
            a.prepared = 0
        a.handover = 1 if (handover and not step and not forward_only) else 0
        if a.handover and handover_cb is not None:
            # ONE ctypes trampoline per object (building a CFUNCTYPE per call is host time inside the timed multi-rank loop); it forwards to
            # the callback of the current call.  An exception inside a ctypes callback cannot propagate: kept and re-raised behind the call
            self._cb_error, self._cb_cur = None, handover_cb
            if getattr(self, '_cb_keep', None) is None:
                def _tramp(which, stream, _user):
                    try:
                        self._cb_cur(int(which), int(stream))
                    except BaseException as e:          # noqa: BLE001
                        self._cb_error = e
                self._cb_keep = L.HANDOVER_FN(_tramp)
                self._cb_ptr = C.cast(self._cb_keep, C.c_void_p)
            a.handover_cb = self._cb_ptr
        else:
            a.handover_cb = None
        a.handover_user = None
        self.mid_called = False
        if mid_cb is not None:
            self._mid_cur = mid_cb
            if getattr(self, '_mid_keep', None) is None:
                def _mid(_stream, _user):
                    self.mid_called = True
                    try:
                        self._mid_cur()
                    except BaseException as e:          # noqa: BLE001  (cannot propagate through ctypes: re-raised behind the call)
                        self._cb_error = e
                self._mid_keep = L.MID_FN(_mid)
                self._mid_ptr = C.cast(self._mid_keep, C.c_void_p)
            self._cb_error = None
            a.mid_cb = self._mid_ptr
        else:
            a.mid_cb = None
        a.mid_user = None
        # (set BEFORE the call: if it fails half-way, helper-stream work may already be queued, and the next _setup must join it before it
        # drops the references to this call's inputs)
        self._pending_deferred = bool(a.defer_update)
        L.check(lib.echr_train_step(C.byref(a), L.stream_ptr()), 'train_step')
        if (a.handover_cb or a.mid_cb) and getattr(self, '_cb_error', None) is not None:
            e, self._cb_error = self._cb_error, None
            raise e
        return self._finish(slot, st, forward_only)

    def _set_tap(self, tap, tap_grad, defer_update, step, forward_only):
        a, d = self.a, self.a.dec
        if (tap.shape[1] if a.event_parts & 2 else 0) + (d.D if a.event_parts & 1 else 0) != a.tsrm.Din or tap.shape[0] < self._tv_needed:
            raise L.EchrHipError('tap_feats %s do not match the model / the event anchors' % (tuple(tap.shape),))
        a.tap, a.Ht = tap.data_ptr(), tap.shape[1]
        self._tap_keep = tap
        if tap_grad is not None and not forward_only:
            if not (tap_grad.is_cuda and tap_grad.dtype == torch.float32 and tap_grad.is_contiguous() and tuple(tap_grad.shape) == tuple(tap.shape)):
                raise ValueError('tap_grad must be a contiguous float32 device tensor shaped like tap_feats %s' % (tuple(tap.shape),))
            a.g_tap = tap_grad.data_ptr()
        else:
            a.g_tap = None
        a.defer_update = 1 if (defer_update and tap_grad is not None and step and not forward_only) else 0
        # 'VH' (scene context = tap_feats.mean(0), CaptionGenerator.py:95-99) with tap_grad: the library spreads d video's span back over g_tap
        vt = self.model.opt.video_context_type
        if 'VH' in vt and a.g_tap:
            o = self.model.opt
            a.vh_offset = (o.lda_dim if 'VL' in vt else 0) + (o.video_dim if 'VC' in vt else 0)
            a.tap_rows = tap.shape[0]
        else:
            a.vh_offset, a.tap_rows = -1, 0

    def _setup(self, tap_feats, c3d_feats, lda_feats, lm_labels, ind_select_list, soi_select_list, targets, masks, step, forward_only,
               tap_grad, defer_update):
        a, m, ar, lib = self.a, self.model, self.arena, self.lib
        if getattr(self, '_pending_deferred', False):
            # a deferred update may still be reading the previous call's inputs (c3d, the staged indices) on the library's streams: order this
            # stream behind it BEFORE the references below are dropped and torch's allocator may hand that memory to someone else
            self.join()
            self._pending_deferred = False
        if not c3d_feats.is_cuda:
            raise L.EchrHipError('FusedTrainStep runs on the GPU only')
        if (ar.flat_p.data_ptr(), ar.flat_g.data_ptr()) != self._epoch_ptrs or not ar.params_in_arena():
            raise RuntimeError('the parameter arena moved since this FusedTrainStep was built')
        soi = np.asarray(soi_select_list, dtype=np.int64).reshape(-1, 2)
        ind = np.asarray(ind_select_list, dtype=np.int64).reshape(-1)
        lens = soi[:, 1] - soi[:, 0]
        N = len(soi)
        Tv = c3d_feats.shape[0] if tap_feats is None else min(c3d_feats.shape[0], tap_feats.shape[0])
        if N == 0 or lens.min() <= 0:
            raise ValueError('every event needs at least one segment (soi=%s)' % (soi.tolist(),))
        if len(ind) != N:
            raise ValueError('ind_select_list and soi_select_list differ in length (%d vs %d)' % (len(ind), N))
        if soi.min() < 0 or soi[:, 1].max() > Tv or ind.min() < 0 or ind.max() >= Tv:
            raise ValueError('event intervals / anchors fall outside the %d feature rows' % Tv)
        labels = lm_labels.numpy() if isinstance(lm_labels, torch.Tensor) and not lm_labels.is_cuda else np.asarray(lm_labels.cpu() if isinstance(lm_labels, torch.Tensor) else lm_labels)
        S = n_decoder_steps(labels)
        if S == 0:
            raise ValueError('label tensor needs at least two columns')
        if labels.shape[0] != N:
            raise ValueError('labels have %d rows for %d events' % (labels.shape[0], N))
        c3d, lda = EF._f32c(c3d_feats), EF._f32c(lda_feats)
        tap = None if tap_feats is None else EF._f32c(tap_feats)
        vt = m.opt.video_context_type
        if vt != 'VL':
            # scene context 'VC' / 'VH' (CaptionGenerator.py:87-104): the mean rows are formed ahead of the call (two small launches); what the
            # library sees as its `video` vector is the concatenation.  'VH' makes the scene vector a function of tap_feats: with tap_grad the
            # library routes d video's span back into it (echr_train_step_args.vh_offset, _set_tap).  prepare() runs before tap_feats exist
            if 'VH' in vt and tap is None:
                raise NotImplementedError("video_context_type with 'VH': the scene vector needs tap_feats, prepare() runs ahead of them "
                                          "(call without prepare())")
            with torch.no_grad():
                lda = EF._f32c(m.get_video_context(tap, c3d, lda, ind_select_list, soi_select_list))
        self._tv_needed = int(max(soi[:, 1].max(), ind.max() + 1))
        # Criterion inputs.  On the host (numpy / CPU tensors, as the reference's loader hands them over, train.py:273-279): they travel with
        # the index vectors, and the rows whose mask is non-zero are listed -- the masked-out label positions behind a caption's end cannot
        # reach the loss (misc/utils.py:66-75 multiplies by the mask), so training forms logits, d logits and the logit-layer products on the
        # active rows only (ECHR_MASKED_ROWS=0: all rows).  On the device: used in place, all rows.
        host_nll = not (isinstance(targets, torch.Tensor) and targets.is_cuda) and not (isinstance(masks, torch.Tensor) and masks.is_cuda)
        act = None
        if host_nll:
            tg_h = np.ascontiguousarray(np.asarray(targets)[:, :S], dtype=np.int32)
            mk_h = np.ascontiguousarray(np.asarray(masks)[:, :S], dtype=np.float32)
            if tg_h.shape != (N, S) or mk_h.shape != (N, S):
                raise ValueError('targets / masks must be [N, >= S] (got %s, %s)' % (tuple(np.asarray(targets).shape), tuple(np.asarray(masks).shape)))
            if MASKED_ROWS[0] and not forward_only:
                # active = up to the LAST non-zero mask entry of each caption: a position behind it reaches neither the loss nor, through
                # the recurrence, any earlier gradient, so every gradient of the reverse recurrence is exactly zero there and the
                # weight-gradient products skip those rows as well (a zero inside a caption stays listed: later steps feed back into it)
                live = np.flip(np.logical_or.accumulate(np.flip(mk_h != 0, 1), 1), 1)
                act = np.flatnonzero(live.T.reshape(-1)).astype(np.int32)               # time-major rows t*N + n, ascending
                if act.size == 0 or act.size == N * S:
                    act = None
        n_act = 0 if act is None else int(act.size)
        host = np.empty((3 + S) * N + n_act + (2 * N * S if host_nll else 0), dtype=np.int32)
        host[:N], host[N:2 * N], host[2 * N:3 * N] = soi[:, 0], lens, ind
        host[3 * N:(3 + S) * N] = labels[:, :S].T.reshape(-1)
        o = (3 + S) * N
        if n_act:
            host[o:o + n_act] = act
        if host_nll:
            host[o + n_act:o + n_act + N * S] = tg_h.reshape(-1)
            host[o + n_act + N * S:] = mk_h.reshape(-1).view(np.int32)
            tgt = msk = None
        else:
            tgt = EF._nll_target(targets if targets.is_cuda else EF.upload(targets, self.dev), S)
            msk = (masks if masks.is_cuda else EF.upload(masks, self.dev))[:, :S].to(torch.float32).contiguous()
        d = a.dec
        a.tsrm.N = d.N = N
        d.A, d.Tv, d.S, d.rows_disjoint = int(lens.max()), c3d.shape[0], S, 1 if EF.rows_disjoint(soi) else 0
        if c3d.shape[1] != d.D or lda.numel() != d.Dv:
            raise L.EchrHipError('feature widths do not match the model (c3d %d, lda %d)' % (c3d.shape[1], lda.numel()))
        d.c3d, d.video = c3d.data_ptr(), lda.data_ptr()
        # (tgt / msk: converted copies that only the argument struct's raw pointers reference -- after prepare() the caller allocates
        # before the second half reads them, so they must stay alive until the next _setup)
        self._keep = (c3d, lda, host, tgt, msk)
        if tap is None:                        # prepare(): tap_feats arrive with the second half
            a.tap, a.Ht, a.g_tap, a.defer_update = None, m.opt.hidden_dim, None, 0
        else:
            self._set_tap(tap, tap_grad, defer_update, step, forward_only)
        a.host_index = host.ctypes.data
        a.n_active, a.host_nll = n_act, 1 if host_nll else 0
        if host_nll:
            a.nll_target, a.nll_target_i64, a.nll_mask = None, 0, None
        else:
            a.nll_target, a.nll_target_i64, a.nll_mask = tgt.data_ptr(), 1 if tgt.dtype == torch.int64 else 0, msk.data_ptr()
        self.last_active_rows = n_act
        drop = m.lm_model.next_drop_state(m.fusion_model.enc_attn.dropout.p)
        drop.training = m.training
        a.drop = drop.c()
        need = lib.echr_train_step_ws_floats(C.byref(a))
        if self.ws is None or self.ws.numel() < need:
            self.join()                        # (a deferred update may still be reading the old workspace on the helper streams)
            self.ws = None                     # (released in stream order by the caching allocator)
            self.ws = torch.empty(need, device=self.dev, dtype=torch.float32)
        a.ws, a.ws_floats = self.ws.data_ptr(), self.ws.numel()
        slot = self.loss_ring[self.calls % self.LOSS_SLOTS]
        self.calls += 1
        a.loss = slot.data_ptr()
        a.overlap_encoder = 1 if m.overlap_encoder else 0
        a.forward_only = 1 if forward_only else 0
        a.do_step = 1 if (step and not forward_only) else 0
        o = self.optim
        if not forward_only:
            for p in ar.params:                # the arena is rewritten from scratch: stale .grad views must not survive as "accumulated" gradients
                if p.grad is not None:
                    p.grad = None
            ar.deferred_clamp = None
            ar.end_backward_pass()
        if a.do_step:
            st = self._flat_state()
            group = o.param_groups[0]
            clip = o.pending_clip if o.pending_clip is not None else self.grad_clip
            o.pending_clip = None
            a.adam_m, a.adam_v, a.adam_step = st['m'].data_ptr(), st['v'].data_ptr(), st['step'] + 1
            a.adam_applied = o.applied_counter(self.dev).data_ptr()
            a.lr, (a.beta1, a.beta2), a.eps = group['lr'], group['betas'], group['eps']
            a.clip = float('inf') if clip is None else float(clip)
        else:
            st = None
        self._slot, self._state = slot, st
        return slot, st

    def _finish(self, slot, st, forward_only):
        a, m, ar = self.a, self.model, self.arena
        if a.do_step:
            st['step'] += 1
            self.optim._count_step([st])
            EF.PARAM_EPOCH[0] += 1
            ar._zeroed = []
        elif not forward_only:
            # gradients are final in stream order: expose them the way the autograd path does (views of the arena); never-used
            # parameters keep .grad None (their slots are zero)
            unused = getattr(self, '_unused', None)
            if unused is None:
                unused = self._unused = {id(p) for p in list(m.lm_model.core.fusion_layer.parameters()) + list(m.fusion_model.h2a_layer.parameters())}
            for i, p in enumerate(ar.params):
                if id(p) not in unused:
                    p.grad = ar.grad_view(i)
            ar._zeroed = [(0, ar.total)]
        return slot[0]


class JointTrainStep(object):
    """The joint 'tap_cg' iteration of the reference (train.py:292-329: tap_feats, props = tap_model(c3d); cg forward; loss = lambda1 *
    tap_loss + lambda2 * cg_loss; backward; clip + step of both optimisers) around `FusedTrainStep`, without an autograd graph:

        [prepare: the caption side's tap-independent half starts]  ->  proposal encoder forward + weighted BCE (echr_sst_fwd,
        echr_tap_bce_fwd)  ->  echr_train_step(tap_grad, defer_update) -- and from INSIDE that call, right behind the work that leads to
        d loss / d tap_feats (echr_train_step_args.mid_cb): the proposal encoder's backward (echr_tap_bce_bwd, echr_sst_bwd) and its clamp +
        Adam; the caption side's parameter-gradient tail and update are forked behind it.

    Why the hook: the proposal encoder's reverse recurrence is one 64-workgroup launch that waits for d tap_feats alone.  Issued from
    Python after the call returned (torch.autograd.backward walking the SST graph) it started ~0.57 ms after d tap_feats was final, behind
    the host's issue of the caption tail; issued from the hook it starts at once and the chip-filling tail follows it instead of
    preceding it.  Same sums as loss.backward() on the joint loss: gradients of both models, both updates."""

    def __init__(self, fused, tap_model, tap_optim, lambda1=1.0, tap_grad_clip=None, early_prepare=True, order=None):
        ar = getattr(tap_model, '_echr_arena', None)
        if not isinstance(tap_optim, ClampAdam) or tap_optim.arena is None or tap_optim.arena is not ar or not ar.params_in_arena():
            raise ValueError('JointTrainStep needs the proposal encoder on a flat arena (tap_model.build_arena()) and a ClampAdam built with it')
        if len(tap_optim.param_groups) != 1 or {id(p) for p in tap_optim.param_groups[0]['params']} != {id(p) for p in ar.params}:
            raise ValueError('the proposal encoder\'s optimiser must hold exactly its parameters in one group')
        self.fused, self.tap_model, self.tap_optim, self.tap_arena = fused, tap_model, tap_optim, ar
        self.lambda1, self.tap_grad_clip, self.early = float(lambda1), tap_grad_clip, bool(early_prepare)
        # 'after' (default): the proposal encoder's backward is queued when the caption call has returned -- on the caller's stream right behind
        # d tap_feats, BESIDE the caption side's tail on the library's streams; 'hook': from inside the call, the tail forked behind it
        # (serialised: measured 3.19 vs 3.00 ms per config-5 iteration)
        self.order = order or os.environ.get('ECHR_JOINT_ORDER', 'after')
        self.lib, self.dev = fused.lib, fused.dev
        self.lam = torch.full((1,), self.lambda1, device=self.dev, dtype=torch.float32)
        self._buf_key, self._bufs = None, None
        self.tap_loss = None

    def _buffers(self, T, D, H, K):
        key = (T, D, H, K)
        if self._buf_key != key:
            self.fused.join()          # (a deferred update may still read the old tap_feats)
            lib, dev, f32 = self.lib, self.dev, torch.float32
            self._bufs = dict(ws=torch.empty(lib.echr_sst_ws_floats(T, D, H, K), device=dev, dtype=f32),
                              wsb=torch.empty(lib.echr_sst_ws_bwd_floats(T, D, H, K), device=dev, dtype=f32),
                              tap=torch.empty(T, H, device=dev, dtype=f32), scores=torch.empty(T, K, device=dev, dtype=f32),
                              g_tap=torch.empty(T, H, device=dev, dtype=f32), g_scores=torch.empty(T, K, device=dev, dtype=f32),
                              loss=torch.zeros(65, device=dev, dtype=f32))
            self._buf_key = key
        return self._bufs

    def __call__(self, c3d_feats, lda_feats, lm_labels, ind_select_list, soi_select_list, targets, masks, tap_masks, tap_labels, w1):
        """Returns (lambda1 * tap_loss + cg_loss) as a 0-d device tensor; `self.tap_loss` / `self.cg_loss` hold the two terms."""
        lib, f, tm = self.lib, self.fused, self.tap_model
        c3d = EF._f32c(c3d_feats)
        ps = [EF._f32c(p) for p in tm.native_params()]
        T, D = c3d.shape
        H, K = ps[1].shape[1], ps[8].shape[0]
        B = self._buffers(T, D, H, K)
        ar = self.tap_arena
        for p in ar.params:
            p.grad = None
        ar.deferred_clamp = None
        ar.end_backward_pass()
        if self.early:          # the caption side's tap-independent half beside the proposal encoder's forward (which leaves 192 CUs idle)
            f.prepare(c3d, lda_feats, lm_labels, ind_select_list, soi_select_list, targets, masks)
        try:
            # the gradient span of the proposal encoder and d tap_feats: zero-filled HERE, off the chain that later leads to its backward
            ar.flat_g.zero_()
            B['g_tap'].zero_()
            p_drop = float(tm.rnn.dropout)
            if tm._drop_seed is None:
                tm._drop_seed = (int(torch.initial_seed()) ^ 0x55AA) & 0xFFFFFFFFFFFFFFFF
            drop = EF.DropState(tm._drop_seed, tm._drop_calls, p_drop > 0.0)
            if p_drop > 0.0:
                tm._drop_calls += 1
            sa = EF.SSTFunction._args(ps, c3d, p_drop, B['ws'], B['tap'], B['scores'])
            dc = drop.c()
            L.check(lib.echr_sst_fwd_states(C.byref(sa), C.byref(dc), L.stream_ptr()), 'sst_fwd_states')           # models/sst_model.py:31-37
            mk, lb, ww = (EF._f32c(x if x.is_cuda else x.to(self.dev)) for x in (tap_masks, tap_labels, w1))
            ww = ww.reshape(-1)

            def tap_criterion():
                # (the proposal head and its criterion, queued BEHIND the caption call: the caption side waits for tap_feats alone, and these kernels would sit between the
                # proposal encoder's forward and the event encoder on the one stream that is the iteration's critical chain there)
                L.check(lib.echr_sst_head_fwd(C.byref(sa), L.stream_ptr()), 'sst_head_fwd')                          # models/sst_model.py:38-39
                L.check(lib.echr_tap_bce_fwd_ws(L.ptr(B['scores']), L.ptr(mk), L.ptr(lb), L.ptr(ww), L.ptr(B['loss'][:1]), L.ptr(B['loss'][1:]), T, K,
                                                L.stream_ptr()), 'tap_bce_fwd')                                        # misc/utils.py:78-99

            def sst_backward():
                # d (lambda1 * tap_loss) / d scores, then the proposal encoder's backward with d loss / d tap_feats from the caption side
                tap_criterion()
                L.check(lib.echr_tap_bce_bwd(L.ptr(B['scores']), L.ptr(mk), L.ptr(lb), L.ptr(ww), L.ptr(self.lam), L.ptr(B['g_scores']), T, K,
                                             L.stream_ptr()), 'tap_bce_bwd')
                two = lambda i, j: (L.c_f * 2)(ar.flat_g.data_ptr() + 4 * ar.offsets[ar.slot(tm.native_params()[i])],
                                               ar.flat_g.data_ptr() + 4 * ar.offsets[ar.slot(tm.native_params()[j])])
                one = lambda i: ar.flat_g.data_ptr() + 4 * ar.offsets[ar.slot(tm.native_params()[i])]
                sg = L.SstGrads(two(0, 4), two(1, 5), two(2, 6), two(3, 7), one(8), one(9), L.ptr(B['g_tap']), L.ptr(B['g_scores']), L.ptr(B['wsb']), 1)
                L.check(lib.echr_sst_bwd(C.byref(sa), C.byref(sg), C.byref(dc), L.stream_ptr()), 'sst_bwd')
                self.tap_optim.step_flat_raw(self.tap_grad_clip)                                                         # train.py:315-317 for tap_optimizer

            cg = f(B['tap'], c3d, lda_feats, lm_labels, ind_select_list, soi_select_list, targets, masks, tap_grad=B['g_tap'], defer_update=True,
                   prepared=self.early, mid_cb=sst_backward if self.order == 'hook' else None)
        except BaseException:
            f.cancel_prepare()
            raise
        if not f.mid_called:          # the library ran the plain form (options the deferred form declines): the same work, behind the call
            sst_backward()
        self._keep = (c3d, mk, lb, ww, ps)
        self.tap_loss, self.cg_loss = B['loss'][0], cg
        return self.lam[0] * B['loss'][0] + cg


class DataParallelStep(object):
    """One data-parallel iteration on the one-call path: the SAME host path for every world size.

    rank r:  echr_train_step(step=False, handover) on its own video  ->  SUM over ranks of the flat gradient arena  ->  clip_gradient + Adam,
    identically on every rank -- the reference's m_batch accumulation (train.py:281-283 sums the per-video gradients, :313-317 clamps the sum
    and steps once) with the m_batch videos on R ranks: SUM without 1/R, clamp AFTER the reduce.

    The exchange is staged: the backward pass inside the call hands over two contiguous arena ranges long before its last kernel -- the
    logit layer (35 % of the gradient bytes, final ~0.1 ms behind the reverse recurrence, on the library's tail stream) and the three LSTM
    layers (39 %, final behind the grouped weight-gradient product on its prepare stream).  For each, the library calls back on the host
    right behind the last launch that writes the range (echr_train_step_args.handover_cb) and the range's collective is queued with that
    library stream as the CURRENT stream (torch.distributed orders its collective stream behind the current stream), so it runs beside the
    rest of the tail and the event encoder's backward without any further stream or event (`via='event'`: the older form -- one side
    stream per range waits for the hand-over event, echr_handover_wait); the remaining ranges (event encoder + embedding, attention: 26 %)
    follow from the caller's stream, asynchronously too, and the caller's stream waits ONCE, for the last collective queued.  Every collective starts behind the reverse recurrence and is
    waited for before clamp + Adam, i.e. before the next iteration's forward recurrence: no collective kernel is ever resident beside a
    persistent pair.  `overlap=False`: ONE collective on the whole arena behind the call."""

    def exchange_report(self):
        """After a measured pass (`self.measure = True`, then a device synchronisation): median / max of the time the caller's stream
        waited for the collectives per step [ms] -- the EXPOSED part of the exchange -- and the ranges of the last step."""
        ms = sorted(a.elapsed_time(b) for a, b in self.exposed_ms)
        self.exposed_ms = []
        names = {0: 'logit layer', 1: 'LSTM layers'}
        rmap = {(lo, hi): names[w] for w, (lo, hi) in self._range.items()}
        return dict(exposed_ms_median=round(ms[len(ms) // 2], 4) if ms else None, exposed_ms_max=round(ms[-1], 4) if ms else None, steps=len(ms),
                    ranges=[dict(name=rmap.get((r['lo'], r['hi']), 'remainder'), bytes=r['bytes'], early=r['early']) for r in self.last_ranges],
                    n_collectives=self.n_collectives, n_early=self.n_early)

    def __init__(self, fused, group=None, overlap=True, algo=None, via=None):
        from . import parallel
        self.P, self.fused, self.group, self.overlap, self.algo = parallel, fused, group, bool(overlap), algo
        # how an early range reaches the collective stream: 'callback' (default) = queued from inside the call with the library's stream current
        # (echr_train_step_args.handover_cb), 'event' = a side stream per range that waits for the hand-over event (echr_handover_wait)
        self.via = via or os.environ.get('ECHR_DP_VIA', 'callback')
        self._ext, self._keep = {}, None
        ar, lm = fused.arena, fused.model.lm_model
        core = lm.core
        lstm = [p for k in range(3) for p in getattr(core, 'layer%d' % k).parameters()]
        self.ranges = []
        for which, params in ((0, [lm.logit.weight, lm.logit.bias]), (1, lstm)):
            slots = sorted(ar.slot(p) for p in params)
            if slots == list(range(slots[0], slots[-1] + 1)):          # one contiguous arena range (it is, for the reference's module order)
                self.ranges.append((which,) + tuple(ar.span(slots)))
        self._range = {which: (lo, hi) for which, lo, hi in self.ranges}
        self.side = [torch.cuda.Stream(device=fused.dev) for _ in self.ranges] if (self.overlap and self.via != 'callback') else []
        self.n_collectives = 0
        self.n_early = 0
        # opt-in instrumentation (bench.py's exchange pass): HIP events around the caller's stream's wait for the collectives -- what of the
        # exchange the backward tail did NOT hide -- and the ranges of the last step (name, bytes, early or not)
        self.measure = False
        self.exposed_ms = []
        self.last_ranges = []

    def _in_order(self):
        """True when every collective of this step ran on ONE in-order device stream (the nccl = RCCL backend: one stream per process group
        and device), so that waiting for the last one queued waits for all of them.  ECHR_DP_WAIT_ALL=1: wait for each."""
        import torch.distributed as dist
        if os.environ.get('ECHR_DP_WAIT_ALL') == '1':
            return False
        try:
            return dist.get_backend(self.group) == 'nccl'
        except Exception:
            return False

    def __call__(self, *args, **kw):
        P, f, ar = self.P, self.fused, self.fused.arena
        import torch.distributed as dist
        active = dist.is_available() and dist.is_initialized()
        pend = []

        def at_handover(which, stream_ptr):
            # host callback from inside echr_train_step, right behind the last launch that writes the range: the collective is queued with the
            # library's own stream as the CURRENT stream, so torch.distributed orders its collective stream behind exactly this point -- no side
            # stream, no further event (five or more streams on this runtime's four hardware queues share queues, and a wait queued on a shared
            # queue stalls the unrelated stream behind it: measured +0.2 ms per iteration with two side streams)
            rng = self._range.get(which)
            if rng is None:
                return
            ext = self._ext.get(stream_ptr)
            if ext is None:
                ext = self._ext[stream_ptr] = torch.cuda.ExternalStream(stream_ptr, device=f.dev)
            with torch.cuda.stream(ext):
                pend.append((rng[0], rng[1], P.reduce_sum_(ar.flat_g[rng[0]:rng[1]], self.group, self.algo, async_op=True)))

        use_cb = self.overlap and active and self.via == 'callback'
        loss = f(*args, step=False, handover=self.overlap and active, handover_cb=at_handover if use_cb else None, **kw)
        n = 0
        if active:
            if self.overlap and not use_cb:          # the event form (echr_handover_wait): one side stream per range
                for (which, lo, hi), s in zip(self.ranges, self.side):
                    with torch.cuda.stream(s):
                        rc = f.lib.echr_handover_wait(which, L.stream_ptr())
                        if rc < 0:
                            L.check(rc, 'handover_wait')
                        if rc == 0:          # (1: this configuration recorded no hand-over point -- the range joins the remainder below)
                            pend.append((lo, hi, P.reduce_sum_(ar.flat_g[lo:hi], self.group, self.algo, async_op=True)))
            self.n_early = len(pend)
            # the remaining ranges, asynchronously as well: queued back to back on the collective stream behind the caller's stream's position
            # (= the end of the backward pass), waited for ONCE -- a blocking collective costs two cross-stream edges (10-20 us each on this
            # runtime) before the next one may even be queued
            works = [w for _, _, w in pend]
            pos = 0
            rng_log = [dict(lo=lo, hi=hi, bytes=4 * (hi - lo), early=True) for lo, hi, _ in pend]
            for lo, hi, _ in sorted(pend, key=lambda t: t[0]) + [(ar.total, ar.total, None)]:
                if lo > pos:
                    works.append(P.reduce_sum_(ar.flat_g[pos:lo], self.group, self.algo, async_op=True))
                    rng_log.append(dict(lo=pos, hi=lo, bytes=4 * (lo - pos), early=False))
                    n += 1
                pos = max(pos, hi)
            n += len(pend)
            self.last_ranges = rng_log
            ev = None
            if self.measure:
                ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                ev[0].record()          # the caller's stream is behind the whole backward pass here: what follows is waiting for the wire
            if works and self._in_order():
                # one process group = one collective stream, in order: the caller's stream waits for the LAST collective queued (the early ones
                # were queued first); the other handles only have to stay alive until then
                works[-1].wait()
                self._keep = works
            else:
                for w in works:
                    w.wait()          # the caller's stream continues behind the collectives
            if ev is not None:
                ev[1].record()
                self.exposed_ms.append(ev)
        self.n_collectives = n
        o = f.optim
        if f.grad_clip is not None:
            from .misc.utils import clip_gradient
            clip_gradient(o, f.grad_clip)          # (recorded; the clamp itself is fused into the step kernel)
        o.step()
        return loss
