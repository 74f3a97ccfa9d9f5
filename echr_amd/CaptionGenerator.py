"""CaptionGenerator -- drop-in for the reference's CaptionGenerator.py (:7-167) on MI355X.

Same constructor side effects on `opt` (video/event/clip_context_dim, :56-84), same `forward`
signature and live modes ('train' -> log-probs [N,S,V+1]; 'eval' -> (seq, logp)), same sub-module
names (`fusion_model`, `lm_model`) and state_dict keys.  The three context levels are built without
python loops over events: index lists are uploaded once as int32 (start, length, anchor) vectors and
every kernel addresses the video features through them.
"""
import os

import torch
from torch import nn

from . import functional as EF
from . import models
from .models.OldModel_NEW import ClipView


class CaptionGenerator(nn.Module):
    def __init__(self, opt):
        super(CaptionGenerator, self).__init__()
        self.opt = opt
        self.change_context_dim()
        if 'TSRM' in opt.fusion_model and 'ER' in opt.event_context_type:
            self.fusion_model = models.setup_fusion(opt)
        self.lm_model = models.setup_lm(opt)
        self.overlap_encoder = os.environ.get('ECHR_OVERLAP_ENCODER', '1') != '0'        # 'train' mode: run the decoder's event-independent precompute concurrently with the event encoder
        if not any(k in opt.video_context_type for k in ('VL', 'VC', 'VH')) or opt.event_context_type not in ('ER1', 'ER2', 'ER3') or opt.clip_context_type != 'CC':
            raise NotImplementedError('the HIP path implements the ECHR recipe: video_context_type from VL / VC / VH (any combination), '
                                      'event_context_type ER1 / ER2 / ER3, clip_context_type=CC (experiments/train_ECHR.sh)')

    def _require_live_decoder(self):
        if type(self.lm_model).__name__ != 'ThreestreamModel':
            self.lm_model.forward()          # raises: parameter-container decoders (show_attend_tell) never run

    def build_arena(self):
        """Pack parameters and gradients into flat device buffers (echr_amd/arena.py).  Call after .cuda(); enables the
        single-launch fused optimiser step and the single-bucket gradient all-reduce."""
        from .arena import ParamArena
        arena = ParamArena(self)
        self.lm_model._echr_arena_ref = arena
        if hasattr(self, 'fusion_model'):
            self.fusion_model._echr_arena_ref = arena
        return arena

    def set_dropout_state(self, seed, calls=0):
        """Pin the counter-based dropout stream (tests / reproducible runs)."""
        self.lm_model._drop_seed = int(seed)
        self.lm_model._drop_calls = int(calls)

    def forward(self, tap_feats, c3d_feats, lda_feats, lm_labels, ind_select_list, soi_select_list, mode='train'):
        if mode not in ('train', 'eval'):
            raise NotImplementedError("mode=%r: only 'train' and 'eval' are live in the reference as shipped (SURVEY section 2 row 12)" % (mode,))
        self._require_live_decoder()
        if not c3d_feats.is_cuda:
            raise EF.L.EchrHipError('CaptionGenerator runs on the GPU only: move the module and its inputs with .cuda()')
        ev = EF.event_index_tensors(soi_select_list, ind_select_list, c3d_feats.device, min(c3d_feats.shape[0], tap_feats.shape[0]))
        drop = self.lm_model.next_drop_state(self.fusion_model.enc_attn.dropout.p if hasattr(self, 'fusion_model') else 0.0)
        drop.training = self.training
        video = self.get_video_context(tap_feats, c3d_feats, lda_feats, ind_select_list, soi_select_list)
        clip, clip_mask = self.get_clip_context(tap_feats, c3d_feats, lda_feats, ind_select_list, soi_select_list, _ev=ev)
        prepared = None
        if mode == 'train' and self.overlap_encoder:
            # the decoder's event-independent precompute starts on a second stream and overlaps the event encoder launched next
            prepared = self.lm_model.prepare(video, clip, clip_mask, lm_labels)
        try:
            event = self.get_event_context(tap_feats, c3d_feats, lda_feats, ind_select_list, soi_select_list, _ev=ev, _drop=drop)
        except Exception:
            if prepared is not None:
                EF.decoder_prepare_cancel()      # the second stream still writes the handle's buffers: order them before they are freed
            raise
        if mode == 'train':
            return self.lm_model(video, event, clip, clip_mask, lm_labels, drop=drop, prepared=prepared)
        return self.lm_model.sample(video, event, clip, clip_mask)

    def change_context_dim(self):
        opt = self.opt
        vt, et, ct = opt.video_context_type, opt.event_context_type, opt.clip_context_type
        opt.video_context_dim = (opt.lda_dim if 'VL' in vt else 0) + (opt.video_dim if 'VC' in vt else 0) + \
                                (opt.hidden_dim if 'VH' in vt else 0)
        if 'ER' in et:
            opt.event_context_dim = opt.d_o
        else:
            opt.event_context_dim = (opt.video_dim if 'EC' in et else 0) + (opt.hidden_dim if 'EH' in et else 0)
        opt.clip_context_dim = (opt.video_dim if 'CC' in ct else 0) + (opt.hidden_dim if 'CH' in ct else 0)

    def get_video_context(self, tap_feats, c3d_feats, lda_feats, ind_select_list, soi_select_list):
        """Scene context (CaptionGenerator.py:87-104): 'VL' the LDA topic vector as is, 'VC' / 'VH' the mean over all T_v rows of the C3D
        features / the proposal encoder's states (echr_col_mean_fwd), concatenated in that order."""
        vt = self.opt.video_context_type
        if vt == 'VL':
            return lda_feats
        parts = []
        if 'VL' in vt:
            parts.append(EF._f32c(lda_feats).reshape(-1))
        if 'VC' in vt:
            parts.append(EF.ColMean.apply(c3d_feats))
        if 'VH' in vt:
            parts.append(EF.ColMean.apply(tap_feats))
        return parts[0] if len(parts) == 1 else torch.cat(parts, 0)

    def get_event_context(self, tap_feats, c3d_feats, lda_feats, ind_select_list, soi_select_list, _ev=None, _drop=None):
        """TSRM over the per-event features (CaptionGenerator.py:106-130): 'ER1' the mean-pooled C3D rows, 'ER2' the SST state at the anchor,
        'ER3' (the recipe) both, concatenated."""
        ev_start, ev_len, ind, _ = _ev if _ev is not None else EF.event_index_tensors(soi_select_list, ind_select_list, c3d_feats.device)
        parts = {'ER1': 1, 'ER2': 2, 'ER3': 3}[self.opt.event_context_type]
        ech = EF.EventPoolGather.apply(c3d_feats, tap_feats, ev_start, ev_len, ind, parts)
        return self.fusion_model(ech, soi_select_list, ev_tensors=(ev_start, ev_len), drop=_drop)

    def get_clip_context(self, tap_feats, c3d_feats, lda_feats, ind_select_list, soi_select_list, _ev=None):
        """'CC' frame-level context.  Internal callers get a zero-copy ClipView (+ None mask); external callers
        (no `_ev`) get the reference's padded [N,A,D] tensor and [N,A] mask (CaptionGenerator.py:140-167)."""
        if _ev is not None:
            ev_start, ev_len, _, A = _ev
            return ClipView(c3d_feats, ev_start, ev_len, A, EF.rows_disjoint(soi_select_list)), None
        ev_start, ev_len, _, A = EF.event_index_tensors(soi_select_list, ind_select_list, c3d_feats.device)
        return ClipView(c3d_feats, ev_start, ev_len, A).materialize()
