"""TSRM8 event-relation encoder -- drop-in for the reference's models/MA_attention_8_NEW.py.

Same class names, constructor side effects on `opt`, parameter names/shapes (state_dict compatible) and
call signatures as the reference (MA_attention_8_NEW.py:9-49, :82-99); the arithmetic runs in
libechr_hip.so (echr_tsrm_fwd / echr_tsrm_bwd).  The two static helpers keep the reference's numpy
contract (float64 in/out) for callers that use them directly; the forward path generates the
embedding on device instead of host numpy + H2D copy (:39-41).
"""
import numpy as np
import torch
import torch.nn as nn

from .. import functional as EF


class MA_Attention8(nn.Module):
    def __init__(self, opt):
        super(MA_Attention8, self).__init__()
        # side effects on the shared option namespace (MA_attention_8_NEW.py:13-22)
        ect = opt.event_context_type
        if 'ER1' in ect:
            opt.TSRM_input_dim = opt.video_dim
        elif 'ER2' in ect:
            opt.TSRM_input_dim = opt.hidden_dim
        elif 'ER3' in ect:
            opt.TSRM_input_dim = opt.video_dim + opt.hidden_dim
        else:
            raise AssertionError('feature_type wrong')
        opt.d_pos_vec = opt.d_feats
        self.h2a_layer = nn.Linear(10, 10)                 # registered but never used by the reference (:23)
        self.output_dim = opt.d_o
        self.use_posit = opt.use_posit
        self.d_pos_vec = opt.d_pos_vec
        self.event_emb = nn.Linear(opt.TSRM_input_dim, opt.d_feats)
        self.fST_type = vars(opt).get('fST_type', 'fST0')
        self.enc_attn = attention_module_multi_head(opt.d_pos_vec, opt.d_feats, (opt.d_feats, opt.d_feats, opt.d_o),
                                                    group=opt.n_head, fST_type=self.fST_type)
        self._drop_state = None                            # set by CaptionGenerator per forward

    def native_params(self):
        e = self.enc_attn
        w_out = e.linear_out_1.weight.view(e.linear_out_1.weight.shape[0], -1)      # [d_o, d_feats, 1, 1] -> [d_o, d_feats]
        return (self.event_emb.weight, self.event_emb.bias, e.pair_pos_fc1.weight, e.pair_pos_fc1.bias,
                e.pair_pos_fc2.weight, e.pair_pos_fc2.bias, e.query_1.weight, e.query_1.bias,
                e.key_1.weight, e.key_1.bias, w_out, e.linear_out_1.bias)

    def fst_mode(self):
        """echr_tsrm_args.fst_mode: how gate and affinity combine (MA_attention_8_NEW.py:148-157); 4 = use_posit off."""
        return fst_mode_of(self.fST_type, self.use_posit)

    def forward(self, feats, soi_select_list, ev_tensors=None, drop=None):
        if ev_tensors is None:
            soi = np.asarray(soi_select_list, dtype=np.int64).reshape(-1, 2)
            t = torch.from_numpy(np.stack([soi[:, 0], soi[:, 1] - soi[:, 0]]).astype(np.int32)).to(feats.device)
            ev_start, ev_len = t[0].contiguous(), t[1].contiguous()
        else:
            ev_start, ev_len = ev_tensors
        if drop is None:
            drop = EF.DropState(training=False)
        # host-known index bounds (event_index_tensors): they let inference over many pairs tabulate the pair MLP (echr_tsrm_args.max_len / max_span)
        params = self.native_params()
        infer = not (torch.is_grad_enabled() and (feats.requires_grad or any(p.requires_grad for p in params)))      # no backward pass can follow
        return EF.TSRMFunction.apply(feats, ev_start, ev_len, self.enc_attn.group, drop, self._grad_sink(),
                                     (1 if infer else 0,) + tuple(getattr(ev_len, 'echr_bounds', (0, 0))) + (self.fst_mode(),), *params)

    def _grad_sink(self):
        arena = getattr(self, '_echr_arena_ref', None)
        if arena is None:
            return None
        e = self.enc_attn
        return EF.GradSink(arena, (self.event_emb.weight, self.event_emb.bias, e.pair_pos_fc1.weight, e.pair_pos_fc1.bias,
                                   e.pair_pos_fc2.weight, e.pair_pos_fc2.bias, e.query_1.weight, e.query_1.bias,
                                   e.key_1.weight, e.key_1.bias, e.linear_out_1.weight, e.linear_out_1.bias))

    @staticmethod
    def extract_position_matrix(bbox, nongt_dim):
        """[N,N,2] float64: (max(|c_i-c_j|/l_i, 1e-3), log(l_j/l_i)); lengths are float32 like the reference (:66-79)."""
        bbox = np.asarray(bbox)
        s, e = bbox[:, :1], bbox[:, 1:2]
        ctr = (s + e) * 0.5
        ln = (e - s).astype(np.float32)
        rel_c = np.abs(ctr - ctr.T) / ln
        rel_c = np.where(rel_c > 1e-3, rel_c, 1e-3)
        rel_l = np.log(ln.T / ln)
        return np.stack([rel_c, rel_l.astype(np.float64)], axis=2)

    @staticmethod
    def extract_position_embedding(position_mat, feat_dim, wave_length=10000):
        """[N,M,feat_dim] float64 sinusoidal embedding, ordered per coordinate as (sin block, cos block) (:51-64)."""
        n, m, _ = position_mat.shape
        nfreq = int(feat_dim // 4)
        freq = np.power(float(wave_length), (4.0 / feat_dim) * np.arange(nfreq, dtype=np.float64))
        ang = (100.0 * position_mat)[..., None] / freq                  # [N,M,2,nfreq]
        return np.concatenate([np.sin(ang), np.cos(ang)], axis=3).reshape(n, m, feat_dim)


def fst_mode_of(fST_type, use_posit):
    if not use_posit:
        return 4
    if fST_type not in ('fST0', 'fST1', 'fST2', 'fST3'):
        # (the reference leaves `weighted_aff` undefined for any other string and fails with a NameError at :159)
        raise ValueError('fST_type must be fST0..fST3 (got %r)' % (fST_type,))
    return int(fST_type[-1])


class attention_module_multi_head(nn.Module):
    """Parameter container with the reference's layout (MA_attention_8_NEW.py:82-99); used through MA_Attention8."""

    def __init__(self, pos_emb_dim, roi_emb_dim, dim=(1024, 1024, 1024), group=16, fST_type='fST0'):
        super(attention_module_multi_head, self).__init__()
        self.d_q, self.d_k, self.d_o = dim
        self.dim_group = (dim[0] // group, dim[1] // group, dim[2] // group)
        self.pos_emb_dim = pos_emb_dim
        self.roi_emb_dim = roi_emb_dim
        self.group = group
        self.fST_type = fST_type
        self.pair_pos_fc1 = nn.Linear(pos_emb_dim, pos_emb_dim)
        self.pair_pos_fc2 = nn.Linear(pos_emb_dim, group)
        self.query_1 = nn.Linear(roi_emb_dim, self.d_q)
        self.key_1 = nn.Linear(roi_emb_dim, self.d_k)
        self.softmax_1 = nn.Softmax(dim=2)
        self.linear_out_1 = nn.Conv2d(in_channels=group * roi_emb_dim, out_channels=self.d_o, kernel_size=(1, 1), stride=1,
                                      groups=group)
        self.dropout = nn.Dropout(0.3)
        self._drop_calls = 0

    def forward(self, roi_feat, position_embedding, use_posit=True):
        """The gated multi-head relation attention on its own (MA_attention_8_NEW.py:101-177): roi_feat [N,d_feats] = embedded events,
        position_embedding [N,N,pos_emb_dim].  Runs through echr_tsrm_attn_fwd; forward only (MA_Attention8.forward is the
        differentiable, fused entry the caption path uses)."""
        w_out = self.linear_out_1.weight.reshape(self.linear_out_1.weight.shape[0], -1)
        ps = (self.pair_pos_fc1.weight, self.pair_pos_fc1.bias, self.pair_pos_fc2.weight, self.pair_pos_fc2.bias,
              self.query_1.weight, self.query_1.bias, self.key_1.weight, self.key_1.bias, w_out, self.linear_out_1.bias)
        drop = EF.DropState(int(torch.initial_seed()) & 0xFFFFFFFFFFFFFFFF, self._drop_calls, self.training, self.dropout.p)
        if self.training:
            self._drop_calls += 1
        with torch.no_grad():
            if position_embedding is None:          # (use_posit off: the kernels never read it)
                position_embedding = roi_feat.new_zeros(roi_feat.shape[0], roi_feat.shape[0], self.pos_emb_dim)
            return EF.tsrm_attention(roi_feat, position_embedding, self.group, ps, self.d_o, drop, fst_mode_of(self.fST_type, use_posit))
