"""CPU oracle for the ECHR caption hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A plain PyTorch-CPU / NumPy restatement of the reference's algorithm for the path named in
BASELINE.json (hierarchical encoder + attention caption decoder).  Only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this module; the product
package `echr_amd` never does (it fails loudly when its HIP library is missing).

Parity status: PINNED.  `tools/make_golden.py` imports the reference's own modules from
/root/reference in the build container and checks every function below against them
(tests/golden/*.npz hold the reference's outputs; tests/test_oracle_golden.py re-checks this file
against those fixtures on any box).

Each function cites the reference lines it restates (paths relative to the reference root).
The restatement is functional: parameters arrive as a dict keyed by the reference's
state_dict names, dropout masks arrive as explicit multiplicative tensors (None = eval mode).
It deliberately keeps the reference's cost structure (e.g. ctx2att is re-projected every
timestep, models/OldModel_NEW.py:381) because it doubles as the timed CPU baseline.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


# ----------------------------------------------------------------------------------------------
# event-relation encoder (TSRM8): models/MA_attention_8_NEW.py
# ----------------------------------------------------------------------------------------------

def position_matrix(soi):
    """Pairwise (delta-centre, delta-log-length) matrix.  MA_attention_8_NEW.py:66-79.

    soi: int array [N,2] of half-open [start,end).  float64 result [N,N,2]; note the reference
    casts lengths to float32 (:70), so the length ratio and its log are float32 arithmetic."""
    soi = np.asarray(soi)
    start = soi[:, 0:1]
    end = soi[:, 1:2]
    center = 0.5 * (start + end)                                   # float64 [N,1]
    length = (end - start).astype('float32')                       # float32 [N,1]
    d_center = np.maximum(np.abs((center - center.T) / length), 1e-3)   # [i,j] = |c_i-c_j| / l_i
    d_length = np.log(length.T / length)                           # [i,j] = log(l_j / l_i), float32
    return np.concatenate((d_center[:, :, None], d_length[:, :, None]), axis=2)


def position_embedding(pos_mat, feat_dim, wave_length=10000):
    """Sinusoidal embedding of the position matrix.  MA_attention_8_NEW.py:51-64.

    Output [N,N,feat_dim] float64 ordered (dc-sin, dc-cos, dl-sin, dl-cos), feat_dim/4 each."""
    n, m, _ = pos_mat.shape
    k = np.arange(0, feat_dim / 4)
    dim = np.power(np.full((1,), wave_length), (4.0 / feat_dim) * k).reshape(1, 1, 1, -1)
    arg = (100.0 * pos_mat)[:, :, :, None] / dim                   # [N,N,2,feat_dim/4]
    emb = np.concatenate((np.sin(arg), np.cos(arg)), axis=3)       # [N,N,2,feat_dim/2]
    return emb.reshape(n, m, feat_dim)


def tsrm_forward(P, feats, soi, n_head=16, drop_mask=None, prefix='fusion_model.', fST_type='fST0', use_posit=1):
    """MA_Attention8.forward + attention_module_multi_head.forward (the recipe: fST0, use_posit=1; the other combinators of :148-157 too).

    MA_attention_8_NEW.py:35-49 and :101-177.  feats [N,Din] -> [N,d_o].
    drop_mask: multiplicative mask for the post-softmax dropout (:162), shape [N,G,N]."""
    N = feats.shape[0]
    G = n_head
    x = F.linear(feats, P[prefix + 'event_emb.weight'], P[prefix + 'event_emb.bias'])     # :44
    d = x.shape[1]
    gate = None
    if use_posit:                                                  # :38-43, :105-116
        pos = position_embedding(position_matrix(soi), P[prefix + 'enc_attn.pair_pos_fc1.weight'].shape[1])
        pos = torch.tensor(pos, dtype=torch.float32)               # :41 (float64 -> float32)
        p1 = F.linear(pos.view(-1, pos.shape[2]), P[prefix + 'enc_attn.pair_pos_fc1.weight'],
                      P[prefix + 'enc_attn.pair_pos_fc1.bias'])    # :110
        gate = F.linear(torch.tanh(p1), P[prefix + 'enc_attn.pair_pos_fc2.weight'],
                        P[prefix + 'enc_attn.pair_pos_fc2.bias'])  # :114
        gate = gate.view(N, N, G).transpose(1, 2)                  # [N,G,N]  :116
    q = F.linear(x, P[prefix + 'enc_attn.query_1.weight'], P[prefix + 'enc_attn.query_1.bias'])
    k = F.linear(x, P[prefix + 'enc_attn.key_1.weight'], P[prefix + 'enc_attn.key_1.bias'])
    dg = d // G                                                    # python-2 integer division (:125)
    qb = q.view(N, G, dg).transpose(0, 1)                          # [G,N,dg]
    kb = k.view(N, G, dg).transpose(0, 1)
    aff = torch.bmm(qb, kb.transpose(1, 2)) * (1.0 / math.sqrt(float(dg)))   # [G,N,N]  :138-140
    aff = aff.transpose(0, 1)                                      # [N,G,N]  :143
    if not use_posit:
        wa = aff                                                   # :157
    elif fST_type == 'fST0':
        wa = gate * aff                                            # :149
    elif fST_type == 'fST1':
        wa = gate + aff                                            # :151
    elif fST_type == 'fST2':
        wa = torch.log(gate.clamp(min=1e-6)) + aff                 # :153
    elif fST_type == 'fST3':
        wa = gate                                                  # :155
    else:
        raise ValueError(fST_type)
    w = torch.softmax(wa, dim=2)                                   # :160
    if drop_mask is not None:
        w = w * drop_mask                                          # :162
    out_t = w.reshape(N * G, N).matmul(x)                          # [N*G, d]  V = un-projected x (:135,:169)
    out_t = out_t.view(N, G * d, 1, 1)
    out = F.conv2d(out_t, P[prefix + 'enc_attn.linear_out_1.weight'],
                   P[prefix + 'enc_attn.linear_out_1.bias'], groups=G)       # :173
    return out.view(N, -1)


# ----------------------------------------------------------------------------------------------
# context builders: CaptionGenerator.py
# ----------------------------------------------------------------------------------------------

def video_context(lda, c3d=None, tap=None, video_context_type='VL'):
    """Scene context: 'VL' -> lda_feats, 'VC' -> c3d_feats.mean(0), 'VH' -> tap_feats.mean(0), concatenated.  CaptionGenerator.py:87-104."""
    parts = []
    if 'VL' in video_context_type:
        parts.append(lda)
    if 'VC' in video_context_type:
        parts.append(c3d.mean(0))
    if 'VH' in video_context_type:
        parts.append(tap.mean(0))
    return parts[0] if len(parts) == 1 else torch.cat(parts, 0)


def event_pool(c3d, soi):
    """Per-event mean over its C3D rows.  CaptionGenerator.py:111-114."""
    return torch.cat([c3d[int(s):int(e)].mean(0, keepdim=True) for s, e in soi], 0)


def event_context(P, tap, c3d, ind, soi, n_head=16, drop_mask=None, event_context_type='ER3', fST_type='fST0', use_posit=1):
    """Event context: 'ER1' mean-pooled C3D -> TSRM, 'ER2' SST hidden at the anchor -> TSRM, 'ER3' their concatenation -> TSRM.
    CaptionGenerator.py:106-130."""
    ec = event_pool(c3d, soi)
    if 'ER1' in event_context_type:
        return tsrm_forward(P, ec, soi, n_head, drop_mask, fST_type=fST_type, use_posit=use_posit)   # :115-117
    eh = tap[torch.as_tensor(np.asarray(ind), dtype=torch.long)]                          # :121
    if 'ER2' in event_context_type:
        return tsrm_forward(P, eh, soi, n_head, drop_mask, fST_type=fST_type, use_posit=use_posit)   # :123-125
    return tsrm_forward(P, torch.cat((ec, eh), 1), soi, n_head, drop_mask, fST_type=fST_type, use_posit=use_posit)


def clip_context(c3d, soi):
    """'CC' frame-level context: zero-padded [N,A,D] + mask [N,A].  CaptionGenerator.py:140-167."""
    lens = [int(e) - int(s) for s, e in soi]
    A = max(lens)
    clip = c3d.new_zeros(len(soi), A, c3d.shape[1])
    mask = c3d.new_zeros(len(soi), A)
    for i, (s, e) in enumerate(soi):
        clip[i, :lens[i]] = c3d[int(s):int(e)]
        mask[i, :lens[i]] = 1
    return clip, mask


# ----------------------------------------------------------------------------------------------
# decoder: models/OldModel_NEW.py
# ----------------------------------------------------------------------------------------------

def attention(P, h, clip, mask, prefix='lm_model.core.attention.'):
    """Additive attention with post-softmax mask + renormalise.  OldModel_NEW.py:376-401.

    Softmax runs over ALL A slots (padding included, :394), then x mask, then / sum (:395-397)."""
    N, A, D = clip.shape
    p_att = F.linear(clip.reshape(-1, D), P[prefix + 'ctx2att.weight'], P[prefix + 'ctx2att.bias']).view(N, A, -1)
    q = F.linear(h, P[prefix + 'h2att.weight'], P[prefix + 'h2att.bias'])
    dot = torch.tanh(p_att + q.unsqueeze(1))
    e = F.linear(dot.view(N * A, -1), P[prefix + 'alpha_net.weight'], P[prefix + 'alpha_net.bias']).view(N, A)
    w = torch.softmax(e, dim=1)
    w = w * mask
    w = w / w.sum(1, keepdim=True)
    return torch.bmm(w.unsqueeze(1), clip).squeeze(1), w


def lstm_cell(x, h, c, w_ih, w_hh, b_ih, b_hh):
    """nn.LSTMCell semantics (gate order i,f,g,o; two bias vectors)."""
    g = F.linear(x, w_ih, b_ih) + F.linear(h, w_hh, b_hh)
    i, f, gg, o = g.chunk(4, 1)
    i, f, gg, o = torch.sigmoid(i), torch.sigmoid(f), torch.tanh(gg), torch.sigmoid(o)
    c2 = f * c + i * gg
    return o * torch.tanh(c2), c2


def core_step(P, xt, video, event, clip, mask, state, dm=None, prefix='lm_model.core.'):
    """ThreeStream_Core.forward.  OldModel_NEW.py:801-823.

    state = (h [3,N,H] of the *dropped* outputs, c [3,N,H]).  dm = optional (m0,m1,m2) masks."""
    h, c = state
    N = xt.shape[0]
    vid = video.unsqueeze(0).expand(N, video.shape[0])
    L = lambda k: (P[prefix + 'layer%d.weight_ih' % k], P[prefix + 'layer%d.weight_hh' % k],
                   P[prefix + 'layer%d.bias_ih' % k], P[prefix + 'layer%d.bias_hh' % k])
    h0, c0 = lstm_cell(torch.cat((xt, event), 1), h[0], c[0], *L(0))           # :807-809
    att, _ = attention(P, h[1], clip, mask, prefix + 'attention.')            # :811 (query = previous dropped h1)
    h1, c1 = lstm_cell(torch.cat((xt, att), 1), h[1], c[1], *L(1))             # :812-813
    h2, c2 = lstm_cell(torch.cat((xt, vid), 1), h[2], c[2], *L(2))             # :816-817
    if dm is not None:
        h0, h1, h2 = h0 * dm[0], h1 * dm[1], h2 * dm[2]                       # :810,:814,:818
    return torch.cat((h0, h1, h2), 1), (torch.stack((h0, h1, h2)), torch.stack((c0, c1, c2)))


def logprobs_state(P, it, video, event, clip, mask, state, dm=None, dm_out=None):
    """OldModel.get_logprobs_state.  OldModel_NEW.py:133-137 (implicit-dim log_softmax on 2-D = dim 1)."""
    xt = F.embedding(it, P['lm_model.embed.weight'])
    out, state = core_step(P, xt, video, event, clip, mask, state, dm)
    if dm_out is not None:
        out = out * dm_out
    logits = F.linear(out, P['lm_model.logit.weight'], P['lm_model.logit.bias'])
    return torch.log_softmax(logits, dim=1), state


def n_decoder_steps(seq):
    """Number of iterations of the teacher-forced loop incl. its early break.  OldModel_NEW.py:105,122."""
    seq = np.asarray(seq)
    S = 0
    for i in range(seq.shape[1] - 1):
        if i >= 1 and seq[:, i].sum() == 0:
            break
        S += 1
    return S


def init_hidden(P, video, event, clip, init_feats_type=''):
    """OldModel.init_hidden.  OldModel_NEW.py:72-96: zeros, or h(-1) = c(-1) = init_linear(cat([video (broadcast) | event | clip.mean(1)]))
    viewed [N,3,H] and transposed to [3,N,H].  clip.mean(1) runs over all A PADDED slots of the clip tensor, like the reference's."""
    N = event.shape[0]
    H = P['lm_model.core.layer0.weight_hh'].shape[1]
    if not any(k in init_feats_type for k in 'VEC'):
        return (event.new_zeros(3, N, H), event.new_zeros(3, N, H))            # :75-78
    feats = []
    if 'V' in init_feats_type:
        feats.append(video.reshape(1, -1).expand(N, video.numel()))
    if 'E' in init_feats_type:
        feats.append(event)
    if 'C' in init_feats_type:
        feats.append(clip.mean(1))
    x = torch.cat(feats, 1)
    m = (x @ P['lm_model.init_linear.weight'].t() + P['lm_model.init_linear.bias']).view(N, 3, H).transpose(0, 1)
    return (m, m)


def decoder_forward(P, video, event, clip, mask, seq, drop=None, init_feats_type=''):
    """OldModel.forward, teacher forcing (ss_prob = 0).  OldModel_NEW.py:98-130.

    drop: None (eval) or callable (site, step, shape) -> multiplicative mask tensor, with
    site in {'h0','h1','h2','out'}.  Returns log-probs [N,S,V1]."""
    N = event.shape[0]
    H = P['lm_model.core.layer0.weight_hh'].shape[1]
    state = init_hidden(P, video, event, clip, init_feats_type)
    outs = []
    for i in range(seq.shape[1] - 1):
        if i >= 1 and int(seq[:, i].sum()) == 0:                               # :122
            break
        dm = dm_out = None
        if drop is not None:
            dm = tuple(drop(s, i, (N, H)) for s in ('h0', 'h1', 'h2'))
            dm_out = drop('out', i, (N, 3 * H))
        lp, state = logprobs_state(P, seq[:, i], video, event, clip, mask, state, dm, dm_out)
        outs.append(lp)
    return torch.stack(outs, 1)


def decoder_sample(P, video, event, clip, mask, seq_length, init_feats_type=''):
    """Greedy OldModel.sample (sample_max=1, eval mode).  OldModel_NEW.py:139-187.

    Returns (seq int64 [N,<=seq_length], logp float32 same shape), or ([],[]) if nothing generated."""
    N = event.shape[0]
    H = P['lm_model.core.layer0.weight_hh'].shape[1]
    state = init_hidden(P, video, event, clip, init_feats_type)
    seq, slp = [], []
    logprobs = None
    unfinished = None
    for t in range(seq_length + 1):
        if t == 0:
            it = torch.zeros(N, dtype=torch.long)
        else:
            sample_lp, it = torch.max(logprobs, 1)                             # :158 (lowest index on ties)
        logprobs, state = logprobs_state(P, it, video, event, clip, mask, state)
        if t >= 1:
            unfinished = (it > 0) if t == 1 else unfinished & (it > 0)
            if int(unfinished.sum()) == 0:
                break
            it = it * unfinished.type_as(it)
            seq.append(it)
            slp.append(sample_lp)
    if not seq:
        return [], []
    return torch.stack(seq, 1), torch.stack(slp, 1)


def lm_criterion(logp, target, mask):
    """LanguageModelCriterion.forward.  misc/utils.py:66-75."""
    S = logp.shape[1]
    target = target[:, :S]
    mask = mask[:, :S]
    nll = -logp.gather(2, target.unsqueeze(2)).squeeze(2) * mask
    return nll.sum() / (mask.sum() + 1e-6)


def caption_forward(P, tap, c3d, lda, labels, ind, soi, mode='train', drop=None, n_head=16, seq_length=None, video_context_type='VL',
                    event_context_type='ER3', fST_type='fST0', use_posit=1, init_feats_type=''):
    """CaptionGenerator.forward for the live modes 'train' / 'eval'.  CaptionGenerator.py:17-43."""
    video = video_context(lda, c3d, tap, video_context_type)
    N = len(soi)
    dmask = drop('tsrm', 0, (N, n_head, N)) if drop is not None else None
    event = event_context(P, tap, c3d, ind, soi, n_head, dmask, event_context_type, fST_type, use_posit)
    clip, mask = clip_context(c3d, soi)
    if mode == 'train':
        return decoder_forward(P, video, event, clip, mask, labels, drop, init_feats_type)
    return decoder_sample(P, video, event, clip, mask, seq_length, init_feats_type)


# ----------------------------------------------------------------------------------------------
# SST proposal encoder + its criterion: models/sst_model.py:31-40, misc/utils.py:78-99
# ----------------------------------------------------------------------------------------------

def sst_forward(P, x, drop_mask=None, prefix=''):
    """SST.forward: nn.LSTM(D -> H, 2 layers, batch_first) over one video [1,T,D] + Linear(H -> K) + sigmoid (sst_model.py:31-40).

    P uses nn.LSTM's parameter names (rnn.weight_ih_l0 ...).  drop_mask: multiplicative [T,H] mask on layer 0's output
    (nn.LSTM's inter-layer dropout), None in eval mode.  Returns (tap_feats [T,H], scores [T,K])."""
    T = x.shape[0]
    H = P[prefix + 'rnn.weight_hh_l0'].shape[1]
    inp = x
    for l in range(2):
        w_ih, w_hh = P[prefix + 'rnn.weight_ih_l%d' % l], P[prefix + 'rnn.weight_hh_l%d' % l]
        b_ih, b_hh = P[prefix + 'rnn.bias_ih_l%d' % l], P[prefix + 'rnn.bias_hh_l%d' % l]
        h, c = x.new_zeros(1, H), x.new_zeros(1, H)
        outs = []
        for t in range(T):
            h, c = lstm_cell(inp[t:t + 1], h, c, w_ih, w_hh, b_ih, b_hh)
            outs.append(h)
        out = torch.cat(outs, 0)
        inp = out * drop_mask if (l == 0 and drop_mask is not None) else out
    scores = torch.sigmoid(F.linear(out, P[prefix + 'scores.weight'], P[prefix + 'scores.bias']))
    return out, scores


def tap_criterion(scores, masks, labels, w1):
    """TAPModelCriterion.forward (misc/utils.py:78-99): weighted BCE, mean over T*K, times K."""
    w0 = 1.0 - w1
    labels = labels * masks
    weights = labels * w0.expand_as(labels) + (1.0 - labels) * w1.expand_as(labels)
    loss = F.binary_cross_entropy((scores.reshape(-1) * masks.reshape(-1)), labels.reshape(-1), weight=weights.reshape(-1))
    return loss * w0.shape[0]


# ----------------------------------------------------------------------------------------------
# optimiser step: misc/utils.py:107-111 + torch.optim.Adam as train.py:209,315-317 configures it
# ----------------------------------------------------------------------------------------------

def clamp_adam_step(p, g, m, v, step, lr, beta1=0.9, beta2=0.999, eps=1e-8, clip=100.0):
    """One element-wise clamp(+-clip) followed by an Adam update (weight_decay 0, no amsgrad).

    In-place on numpy/torch float32 arrays p, m, v; `step` is the 1-based step count."""
    g = g.clamp(-clip, clip) if isinstance(g, torch.Tensor) else np.clip(g, -clip, clip)
    m *= beta1
    m += (1 - beta1) * g
    v *= beta2
    v += (1 - beta2) * g * g
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    denom = (v.sqrt() if isinstance(v, torch.Tensor) else np.sqrt(v)) / math.sqrt(bc2) + eps
    p -= (lr / bc1) * (m / denom)
    return p


# ----------------------------------------------------------------------------------------------
# proposal selection (index outputs): eval_utils.py:259-287
# ----------------------------------------------------------------------------------------------

def top_proposals_nms(scores, overlap=0.8, topN=1000):
    """gettop1000_nms's index outputs: (pick order, props [M,2], scores [M]).  eval_utils.py:290-331: candidates (n, k < min(n, K))
    -> [n-k, n+1]; greedy 1-D NMS by descending score with the inclusive (+1) temporal IoU, float64 like numpy.  Ties in the score
    are broken towards the LATER candidate (a stable ascending sort's last element); the reference's unstable argsort leaves that
    case undefined."""
    scores = np.asarray(scores)
    T, K = scores.shape
    props, sc = [], []
    for n in range(T):
        for k in range(min(n, K)):
            props.append([n - k, n + 1])
            sc.append(float(scores[n, k]))
    props = np.array(props).reshape(-1, 2)
    sc = np.array(sc)
    if len(sc) == 0:
        return [], props, sc
    t1, t2 = props[:, 0], props[:, 1]
    ind = np.argsort(sc, kind='stable')
    area = (t2 - t1 + 1).astype(float)
    pick = []
    while len(ind) > 0 and len(pick) < topN:
        i = ind[-1]
        pick.append(int(i))
        ind = ind[:-1]
        tt1 = np.maximum(t1[i], t1[ind])
        tt2 = np.minimum(t2[i], t2[ind])
        wh = np.maximum(0., tt2 - tt1 + 1.0)
        o = wh / (area[i] + area[ind] - wh)
        ind = ind[np.nonzero(o <= overlap)[0]]
    return pick, props[pick, :], sc[pick]


def top_proposals(scores, tap_masks, topN=1000, score_thres=0.0):
    """gettop1000's index outputs: (index_select_list, featstamp_list, confidence).  eval_utils.py:259-287."""
    scores = np.asarray(scores) * np.asarray(tap_masks)
    T, K = scores.shape
    flat = np.sort(scores.reshape(-1))
    thr = max(flat[-min(len(flat), topN)], score_thres)
    ind, feat, conf = [], [], []
    for n in range(T):
        for k in range(K):
            if n >= k and scores[n, k] >= thr:
                ind.append(n)
                feat.append([n - k, n + 1])
                conf.append(float(scores[n, k]))
    return ind, feat, conf


def topn_nms(props, prop_scores, sent_score, nms_overlap=0.999, topN=1000):
    """eval_utils.py:230-256 `gettopN_nms`, element by element (test infrastructure: plain loops): pick order only."""
    props = [(float(a), float(b)) for a, b in props]
    order = list(np.argsort(np.asarray(prop_scores)))            # eval_utils.py:238
    pick = []
    while order and len(pick) < topN:                            # :241
        i = order[-1]
        keep, same = [], []
        for j in order:
            wh = max(0.0, min(props[i][1], props[j][1]) - max(props[i][0], props[j][0]) + 1e-3)          # :243-245
            o = wh / ((props[i][1] - props[i][0] + 1e-3) + (props[j][1] - props[j][0] + 1e-3) - wh)     # :246
            if o >= nms_overlap:
                same.append(j)                                   # :247
            if o <= nms_overlap:
                keep.append(j)                                   # :251
        best = same[0]
        for j in same:                                           # :248 argmax -> first maximum
            if sent_score[j] > sent_score[best]:
                best = j
        pick.append(int(best))                                   # :249-250
        order = keep
    return pick


def rerank(vid_info):
    """eval_utils.py:334-345 `reranking`."""
    sc = sorted(v['re_score'] for v in vid_info)
    thr = sc[-min(len(sc), 10)]
    return [v for v in vid_info if v['re_score'] >= thr]
