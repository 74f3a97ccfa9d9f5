"""Three-stream attention caption decoder -- drop-in for the reference's models/OldModel_NEW.py
(the `three_stream` branch that models.setup_lm can return: OldModel :18-187, Attention :366-401,
ThreeStream_Core :762-823, ThreestreamModel :1040-1043).

Module/parameter names and shapes match the reference so `state_dict()`s are interchangeable.  The
sub-modules (nn.Embedding, nn.Linear, nn.LSTMCell) are parameter containers only: `forward` and
`sample` run the whole sequence through libechr_hip.so (echr_decoder_fwd/bwd, echr_decoder_sample).
"""
import numpy as np
import torch
import torch.nn as nn

from .. import functional as EF


class ClipView(object):
    """The frame-level ('CC') context without the zero-padded copy: event n's slot a is row
    ev_start[n] + a of `feats` [T,D] (CaptionGenerator.py:140-167 materialises [N,A,D] + mask instead)."""

    def __init__(self, feats, ev_start, ev_len, max_len, rows_disjoint=False):
        self.feats, self.ev_start, self.ev_len, self.max_len = feats, ev_start, ev_len, int(max_len)
        self.rows_disjoint = bool(rows_disjoint)          # no two events share a row of `feats`

    @property
    def shape(self):
        return (self.ev_start.numel(), self.max_len, self.feats.shape[1])

    def materialize(self):
        """(clip [N,A,D], mask [N,A]) exactly as the reference builds them (pure data movement)."""
        N, A, D = self.shape
        a = torch.arange(A, device=self.feats.device)
        mask = (a[None, :] < self.ev_len[:, None].long())
        rows = (self.ev_start[:, None].long() + a[None, :]).clamp_(max=self.feats.shape[0] - 1)
        clip = self.feats[rows] * mask[:, :, None].to(self.feats.dtype)
        return clip, mask.to(self.feats.dtype)

    @staticmethod
    def from_padded(clip, clip_mask):
        """Wrap a reference-style padded clip tensor [N,A,D] + prefix mask [N,A] (one small D2H sync for the lengths)."""
        N, A, D = clip.shape
        lens = clip_mask.reshape(N, A).sum(1).round().to(torch.int32)
        if int(lens.min()) <= 0:
            raise ValueError('clip_mask has an empty row')
        starts = (torch.arange(N, device=clip.device, dtype=torch.int32) * A)
        return ClipView(clip.reshape(N * A, D).contiguous(), starts.contiguous(), lens.contiguous(), A, rows_disjoint=True)


def n_decoder_steps(seq):
    """Iterations of the teacher-forced loop incl. its early break on an all-zero label column
    (OldModel_NEW.py:105,122), computed once on the host instead of one device sync per step."""
    if isinstance(seq, torch.Tensor):
        nz = (seq != 0).any(0).cpu().numpy()
    else:
        nz = (np.asarray(seq) != 0).any(0)
    L = len(nz)
    S = 0
    for i in range(L - 1):
        if i >= 1 and not nz[i]:
            break
        S += 1
    return S


class OldModel(nn.Module):
    def __init__(self, opt):
        super(OldModel, self).__init__()
        self.CG_init_feats_type = opt.CG_init_feats_type
        self.opt = opt
        self.vocab_size = opt.CG_vocab_size
        self.input_encoding_size = opt.CG_input_encoding_size
        self.rnn_type = opt.CG_rnn_type
        self.rnn_size = opt.CG_rnn_size
        self.num_layers = opt.CG_num_layers
        self.drop_prob_lm = opt.CG_drop_prob
        self.seq_length = opt.CG_seq_length
        self.CG_init_feats_dim = self.decide_init_feats_dim()
        self.ss_prob = 0.0
        if self.CG_init_feats_dim:
            # non-zero initial state (OldModel_NEW.py:38-39, :79-96): h(-1) = c(-1) = init_linear(cat(selected contexts)); registered ahead of
            # `embed` like the reference does
            self.init_linear = nn.Linear(self.CG_init_feats_dim, self.num_layers * self.rnn_size)
        self.embed = nn.Embedding(self.vocab_size + 1, self.input_encoding_size)
        if 'three_stream' not in opt.caption_model or 'three_stream_2stream' in opt.caption_model:
            raise NotImplementedError('caption_model=%r: only the three_stream decoder is on the HIP path' % (opt.caption_model,))
        self.logit = nn.Linear(3 * self.rnn_size, self.vocab_size + 1)
        self.dropout = nn.Dropout(self.drop_prob_lm)
        self.init_weights()
        self._drop_seed = None
        self._drop_calls = 0

    def decide_init_feats_dim(self):
        t, o = self.CG_init_feats_type, self.opt
        return (o.video_context_dim if 'V' in t else 0) + (o.event_context_dim if 'E' in t else 0) + \
               (o.clip_context_dim if 'C' in t else 0)

    def init_weights(self):
        r = 0.1                                            # OldModel_NEW.py:66-70
        with torch.no_grad():
            self.embed.weight.uniform_(-r, r)
            self.logit.bias.zero_()
            self.logit.weight.uniform_(-r, r)

    # ---- dropout bookkeeping ----------------------------------------------------------------------
    def next_drop_state(self, p_tsrm=0.3):
        """Dropout configuration for the next forward: fresh Philox offset per training-mode call."""
        if self._drop_seed is None:
            seed = int(torch.initial_seed())
            if torch.distributed.is_available() and torch.distributed.is_initialized():
                # data parallel: every rank must draw its OWN masks (replicas see different videos, like m_batch different iterations
                # of the reference), so the rank is mixed into the seed
                seed ^= ((torch.distributed.get_rank() + 1) * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
            self._drop_seed = seed & 0xFFFFFFFFFFFFFFFF
        c = self.core
        st = EF.DropState(self._drop_seed, self._drop_calls, self.training, p_tsrm, c.dropout0.p, self.dropout.p)
        if self.training:
            self._drop_calls += 1
        return st

    def native_params(self):
        cached = self.__dict__.get('_native_params')
        if cached is not None:
            return cached
        self.__dict__['_native_params'] = self._native_params_now()          # Parameter OBJECTS are stable (.data may move into an arena)
        return self.__dict__['_native_params']

    def invalidate_native_caches(self):
        """Forget the cached Parameter tuple and the greedy decoder's parameter-derived tables.  Called by everything of nn.Module that can
        REPLACE Parameter objects or rewrite their storage behind torch's version counters (`_apply`: .to / .cuda / to_empty; load_state_dict,
        also with assign=True; pickling).  Code that re-assigns a sub-module's Parameter by hand (weight tying, pruning) or writes through
        `.data` must call it too."""
        self.__dict__.pop('_native_params', None)
        self.__dict__.pop('_sample_tables', None)

    def _apply(self, fn, *args, **kwargs):
        self.invalidate_native_caches()
        return super(OldModel, self)._apply(fn, *args, **kwargs)

    def _load_from_state_dict(self, *args, **kwargs):
        self.invalidate_native_caches()
        return super(OldModel, self)._load_from_state_dict(*args, **kwargs)

    def __getstate__(self):
        d = dict(self.__dict__)          # the caches hold Parameter references / a large device tensor: neither belongs in a pickle or a deepcopy
        d.pop('_native_params', None)
        d.pop('_sample_tables', None)
        return d

    def _native_params_now(self):
        c = self.core
        a = c.attention
        return (self.embed.weight, self.logit.weight, self.logit.bias,
                c.layer0.weight_ih, c.layer1.weight_ih, c.layer2.weight_ih,
                c.layer0.weight_hh, c.layer1.weight_hh, c.layer2.weight_hh,
                c.layer0.bias_ih, c.layer1.bias_ih, c.layer2.bias_ih,
                c.layer0.bias_hh, c.layer1.bias_hh, c.layer2.bias_hh,
                a.ctx2att.weight, a.ctx2att.bias, a.h2att.weight, a.h2att.bias, a.alpha_net.weight, a.alpha_net.bias)

    @staticmethod
    def _clip_view(clip, clip_mask):
        return clip if isinstance(clip, ClipView) else ClipView.from_padded(clip, clip_mask)

    def _tokens(self, seq, dev):
        """[S,N] time-major int32 input tokens of the teacher-forced loop (S = its iteration count, OldModel_NEW.py:105,122)."""
        S = n_decoder_steps(seq)
        if S == 0:
            raise ValueError('label tensor needs at least two columns')
        seq_t = torch.as_tensor(np.asarray(seq) if not isinstance(seq, torch.Tensor) else seq)
        if seq_t.is_cuda:
            return seq_t[:, :S].t().to(device=dev, dtype=torch.int32).contiguous()
        # host labels: slice / transpose / cast on the host, ONE small H2D copy instead of three device ops
        return EF.upload(seq_t[:, :S].t().to(torch.int32).contiguous(), dev)

    def prepare(self, video, clip, clip_mask, seq):
        """Start the part of forward() that does not need the event context (packs, ctx2att over the video, token-side gate products) on
        the library's second stream; call BEFORE launching the event encoder and pass the result to forward(prepared=...)."""
        if self.training and self.ss_prob > 0.0:
            raise NotImplementedError('scheduled sampling (ss_prob > 0) is never enabled by the reference and is not on the HIP path')
        cv = self._clip_view(clip, clip_mask)
        tokens = self._tokens(seq, cv.feats.device)
        return EF.decoder_prepare(video, cv.feats, cv.ev_start, cv.ev_len, tokens, cv.max_len, cv.rows_disjoint, self.native_params())

    def forward(self, video, event, clip, clip_mask, seq, drop=None, prepared=None):
        """Teacher-forced log-probs [N,S,V+1] (OldModel_NEW.py:98-130)."""
        if self.training and self.ss_prob > 0.0:
            raise NotImplementedError('scheduled sampling (ss_prob > 0) is never enabled by the reference and is not on the HIP path')
        cv = self._clip_view(clip, clip_mask)
        tokens = prepared['tokens'] if prepared is not None else self._tokens(seq, event.device)
        if drop is None:
            drop = self.next_drop_state()
        arena = getattr(self, '_echr_arena_ref', None)
        sink = EF.GradSink(arena, self.native_params()) if arena is not None else None
        return EF.DecoderFunction.apply(video, event, cv.feats, cv.ev_start, cv.ev_len, tokens, cv.max_len, cv.rows_disjoint, drop, sink,
                                        prepared, self._initial_state(video, event, cv), *self.native_params())

    def _initial_state(self, video, event, cv):
        """None (the recipe: zero state) or h0 [N, 3H] = init_linear(cat([video | event | clip.mean(1)])) (OldModel_NEW.py:79-92) -- the decoder
        entry points take the map in this layout; the reference's view(N, 3, H).transpose(0, 1) is `init_hidden`'s."""
        if not self.CG_init_feats_dim:
            return None
        t = self.CG_init_feats_type
        arena = getattr(self, '_echr_arena_ref', None)
        sink = EF.GradSink(arena, (self.init_linear.weight, self.init_linear.bias)) if arena is not None else None
        return EF.InitState.apply(video, event, cv.feats, cv.ev_start, cv.ev_len, cv.max_len, ('V' in t, 'E' in t, 'C' in t), sink,
                                  self.init_linear.weight, self.init_linear.bias)

    def init_hidden(self, video, event, clip, clip_mask=None):
        """Initial state (h, c), each [3,N,H] (OldModel_NEW.py:72-96): zeros, or -- with CG_init_feats_type -- the init_linear map for both."""
        n = event.shape[0] if event is not None else clip.shape[0]
        w = self.logit.weight
        if not self.CG_init_feats_dim:
            return (w.new_zeros(self.num_layers, n, self.rnn_size), w.new_zeros(self.num_layers, n, self.rnn_size))
        h0 = self._initial_state(video, event, self._clip_view(clip, clip_mask))
        m = h0.view(n, self.num_layers, self.rnn_size).transpose(0, 1)
        return (m, m)

    def get_logprobs_state(self, it, video, event, clip, clip_mask, state):
        """One decoder timestep, state in / state out (OldModel_NEW.py:133-137): (log-probs [N,V+1], (h', c')).

        Runs through echr_decoder_step; forward only (the reference's training loop is forward(), which keeps the whole
        sequence inside the library).  In training mode each call draws a fresh dropout stream."""
        cv = self._clip_view(clip, clip_mask)
        drop = self.next_drop_state()
        with torch.no_grad():
            return EF.decoder_step(it, video, event, cv.feats, cv.ev_start, cv.ev_len, cv.max_len, state, self.native_params(), drop)

    def sample(self, video, event, clip, clip_mask, opt={}):
        """Decoding without beam search (OldModel_NEW.py:139-187, beam_size = 1): greedy arg-max (sample_max = 1, the recipe's setting) or,
        with sample_max = 0, a draw from softmax(logp / temperature) per step (:160-168).  The draws come from the library's
        counter-based Philox stream (seed: set_dropout_state / torch.initial_seed, advanced once per call), not from
        torch.multinomial's generator: same distribution, different random numbers."""
        if opt.get('beam_size', 1) != 1:
            raise NotImplementedError('beam search (beam_size > 1) is not on the HIP path')
        cv = self._clip_view(clip, clip_mask)
        multinomial = opt.get('sample_max', 1) != 1
        seed = 0
        if multinomial:
            if self._drop_seed is None:
                self.next_drop_state()                     # derives the (rank-mixed) seed once
            self._sample_calls = getattr(self, '_sample_calls', 0) + 1
            seed = (self._drop_seed * 0x9E3779B97F4A7C15 + self._sample_calls) & 0xFFFFFFFFFFFFFFFF
        with torch.no_grad():
            if '_sample_tables' not in self.__dict__:
                self._sample_tables = {}          # decoding operands derived from the parameters alone, reused across calls (EF.greedy_sample)
            return EF.greedy_sample(video, event, cv.feats, cv.ev_start, cv.ev_len, cv.max_len, self.seq_length,
                                    self.native_params(), multinomial=multinomial, temperature=float(opt.get('temperature', 1.0)),
                                    seed=seed, table_cache=self._sample_tables, h0=self._initial_state(video, event, cv))


class Attention(nn.Module):
    """Additive attention parameters (OldModel_NEW.py:366-375); arithmetic fused into the decoder kernels."""

    def __init__(self, opt):
        super(Attention, self).__init__()
        self.rnn_size = opt.CG_rnn_size
        self.att_hid_size = opt.CG_att_hid_size
        self.att_feat_size = opt.clip_context_dim
        self.ctx2att = nn.Linear(self.att_feat_size, self.att_hid_size)
        self.h2att = nn.Linear(self.rnn_size, self.att_hid_size)
        self.alpha_net = nn.Linear(self.att_hid_size, 1)


class ThreeStream_Core(nn.Module):
    """Three independent LSTM-cell streams (event / attended clip / scene) + late fusion (OldModel_NEW.py:762-799)."""

    def __init__(self, opt):
        super(ThreeStream_Core, self).__init__()
        self.opt = opt
        self.input_encoding_size = opt.CG_input_encoding_size
        self.rnn_type = opt.CG_rnn_type
        self.rnn_size = opt.CG_rnn_size
        self.drop_prob_lm = opt.CG_drop_prob
        self.fc_feat_size = opt.CG_fc_feat_size
        self.att_feat_size = opt.clip_context_dim
        self.att_hid_size = opt.CG_att_hid_size
        # accepted and ignored, exactly as the reference's ThreeStream_Core does: CG_input_dim is computed (:775-776,:790-799) and never used --
        # no layer is sized by it and forward (:801-823) never reads it
        self.CG_input_feats_type = opt.CG_input_feats_type
        self.CG_input_dim = sum(d for c, d in (('V', opt.video_context_dim), ('E', opt.event_context_dim), ('C', opt.clip_context_dim))
                                if c in self.CG_input_feats_type)
        E = self.input_encoding_size
        self.layer0 = nn.LSTMCell(opt.event_context_dim + E, self.rnn_size)
        self.layer1 = nn.LSTMCell(opt.clip_context_dim + E, self.rnn_size)
        self.layer2 = nn.LSTMCell(opt.video_context_dim + E, self.rnn_size)
        self.fusion_layer = nn.Linear(self.rnn_size * 3, self.rnn_size)      # registered, never used (:783)
        self.attention = Attention(opt)
        self.dropout0 = nn.Dropout(0.5)
        self.dropout1 = nn.Dropout(0.5)
        self.dropout2 = nn.Dropout(0.5)


class ThreestreamModel(OldModel):
    def __init__(self, opt):
        super(ThreestreamModel, self).__init__(opt)
        self.core = ThreeStream_Core(opt)


class ShowAttendTellCore(nn.Module):
    """Parameter container with the reference's layout (OldModel_NEW.py:190-216): nn.LSTM(E + input_dim, H, num_layers, bias=False) +
    the additive-attention projections.  No arithmetic: see ShowAttendTellModel."""

    def __init__(self, opt):
        super(ShowAttendTellCore, self).__init__()
        self.opt = opt
        t = opt.CG_input_feats_type
        self.CG_input_dim = (opt.video_context_dim if 'V' in t else 0) + (opt.event_context_dim if 'E' in t else 0) + \
                            (opt.clip_context_dim if 'C' in t else 0)                                        # :219-227
        self.rnn = getattr(nn, opt.CG_rnn_type.upper())(opt.CG_input_encoding_size + self.CG_input_dim, opt.CG_rnn_size, opt.CG_num_layers,
                                                        bias=False, dropout=opt.CG_drop_prob)
        if opt.CG_att_hid_size > 0:
            self.ctx2att = nn.Linear(opt.clip_context_dim, opt.CG_att_hid_size)
            self.h2att = nn.Linear(opt.CG_rnn_size, opt.CG_att_hid_size)
            self.alpha_net = nn.Linear(opt.CG_att_hid_size, 1)
        else:
            self.ctx2att = nn.Linear(opt.clip_context_dim, 1)
            self.h2att = nn.Linear(opt.CG_rnn_size, 1)


class ShowAttendTellModel(nn.Module):
    """`caption_model='show_attend_tell'` (models/__init__.py:7-8, OldModel_NEW.py:1009-1012) as a PARAMETER CONTAINER: constructible,
    state_dict-compatible with the reference (embed, logit [V+1, H], core.rnn.*, core.ctx2att / h2att / alpha_net), movable, savable.
    experiments/train_SST.sh selects it while it trains the proposal encoder with the captioner idle (train.py:291-295), so that recipe
    can build its cg_model through this package and write / read the reference's checkpoints.  Its arithmetic is an ablation outside the
    ECHR hot path: forward / sample raise."""

    def __init__(self, opt):
        super(ShowAttendTellModel, self).__init__()
        self.opt = opt
        self.vocab_size, self.seq_length, self.ss_prob = opt.CG_vocab_size, opt.CG_seq_length, 0.0
        self.rnn_size, self.num_layers = opt.CG_rnn_size, opt.CG_num_layers
        t = opt.CG_init_feats_type
        init_dim = (opt.video_context_dim if 'V' in t else 0) + (opt.event_context_dim if 'E' in t else 0) + (opt.clip_context_dim if 'C' in t else 0)
        if init_dim:
            self.init_linear = nn.Linear(init_dim, self.num_layers * self.rnn_size)                          # :36-37
        self.embed = nn.Embedding(self.vocab_size + 1, opt.CG_input_encoding_size)
        self.logit = nn.Linear(self.rnn_size, self.vocab_size + 1)                                           # :49-51
        self.dropout = nn.Dropout(opt.CG_drop_prob)
        with torch.no_grad():                                                                                # init_weights :66-70
            self.embed.weight.uniform_(-0.1, 0.1)
            self.logit.bias.zero_()
            self.logit.weight.uniform_(-0.1, 0.1)
        self.core = ShowAttendTellCore(opt)

    def _idle(self, *a, **k):
        raise NotImplementedError("caption_model='show_attend_tell' is a parameter container here (the recipe that selects it, "
                                  "experiments/train_SST.sh, never runs the captioner); the HIP path implements 'three_stream'")

    forward = sample = get_logprobs_state = _idle


def _ablation(name):
    class _Unsupported(nn.Module):
        def __init__(self, *a, **k):
            raise NotImplementedError('%s is an ablation variant outside the ECHR hot path (SURVEY section 2 row 4)' % name)
    _Unsupported.__name__ = name
    return _Unsupported


# names the reference's models/__init__.py imports (models/__init__.py:1); ThreestreamModel is live, ShowAttendTellModel holds parameters
for _n in ('AllImgModel', 'H3Model', 'TwostreamModel', 'Twostream_jump_Model', 'TwostreamModel_3LSTM',
           'H3denseModel', 'H3denaddModel', 'ThreestreamModel_2stream', 'ThreestreamModel_2stream_LDA',
           'ThreestreamModel_2stream_CC'):
    globals()[_n] = _ablation(_n)
