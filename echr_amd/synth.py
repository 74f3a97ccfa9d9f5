"""Synthetic options, parameters and video batches for the ECHR caption hot path.

There is no dataset in this environment, so tests, golden fixtures and bench.py all draw their
inputs from `np.random.RandomState` streams defined here (portable between the build container
and the GPU box).  Shapes follow the ECHR recipe (reference: experiments/train_ECHR.sh:5 and the
defaults in opts.py:81-161): the batch axis is the N events of ONE video (opts.py:93,187).
"""
import math
from types import SimpleNamespace

import numpy as np


def default_opt(vocab_size=5000, seq_length=19, **over):
    """The option namespace CaptionGenerator reads, at the values train_ECHR.sh selects."""
    opt = SimpleNamespace(
        caption_model='three_stream', CG_num_layers=3,
        video_context_type='VL', event_context_type='ER3', clip_context_type='CC',
        lda_dim=100, video_dim=500, hidden_dim=512,
        fusion_model='TSRM8', use_posit=1, n_head=16, d_feats=512, d_o=512, fST_type='fST0',
        CG_rnn_size=512, CG_rnn_type='lstm', CG_input_encoding_size=512, CG_att_hid_size=512,
        CG_fc_feat_size=512, CG_drop_prob=0.5, CG_input_feats_type='', CG_init_feats_type='',
        CG_vocab_size=vocab_size, CG_seq_length=seq_length,
        video_context_dim=0, event_context_dim=0, clip_context_dim=0,
        K=256, tap_model='SST', tap_rnn_type='LSTM', rnn_num_layers=2, rnn_dropout=0.5,
        grad_clip=100.0, lr=5e-5, optim_alpha=0.9, optim_beta=0.999, optim_epsilon=1e-8, weight_decay=0,
    )
    for k, v in over.items():
        setattr(opt, k, v)
    return opt


def state_dict_shapes(opt):
    """Name -> shape of CaptionGenerator.state_dict() (printed from the reference)."""
    V1 = opt.CG_vocab_size + 1
    H = opt.CG_rnn_size
    E = opt.CG_input_encoding_size
    Ha = opt.CG_att_hid_size
    D = opt.video_dim
    Df = opt.d_feats
    G = opt.n_head
    et = opt.event_context_type                       # TSRM input (MA_attention_8_NEW.py:13-20): ER1 pooled C3D, ER2 SST state, ER3 both
    tsrm_in = opt.video_dim if 'ER1' in et else (opt.hidden_dim if 'ER2' in et else opt.video_dim + opt.hidden_dim)
    vt = opt.video_context_type
    vi = (opt.lda_dim if 'VL' in vt else 0) + (opt.video_dim if 'VC' in vt else 0) + (opt.hidden_dim if 'VH' in vt else 0)
    ev, cl = opt.d_o, opt.video_dim                   # context widths for ER3 / CC; vi: VL / VC / VH (CaptionGenerator.py:56-64)
    s = {
        'fusion_model.h2a_layer.weight': (10, 10), 'fusion_model.h2a_layer.bias': (10,),
        'fusion_model.event_emb.weight': (Df, tsrm_in), 'fusion_model.event_emb.bias': (Df,),
        'fusion_model.enc_attn.pair_pos_fc1.weight': (Df, Df), 'fusion_model.enc_attn.pair_pos_fc1.bias': (Df,),
        'fusion_model.enc_attn.pair_pos_fc2.weight': (G, Df), 'fusion_model.enc_attn.pair_pos_fc2.bias': (G,),
        'fusion_model.enc_attn.query_1.weight': (Df, Df), 'fusion_model.enc_attn.query_1.bias': (Df,),
        'fusion_model.enc_attn.key_1.weight': (Df, Df), 'fusion_model.enc_attn.key_1.bias': (Df,),
        'fusion_model.enc_attn.linear_out_1.weight': (opt.d_o, Df, 1, 1), 'fusion_model.enc_attn.linear_out_1.bias': (opt.d_o,),
        'lm_model.embed.weight': (V1, E),
        'lm_model.logit.weight': (V1, 3 * H), 'lm_model.logit.bias': (V1,),
    }
    it = getattr(opt, 'CG_init_feats_type', '')       # non-zero initial state (OldModel_NEW.py:38-39,55-63): Linear(selected context widths -> 3H)
    init_in = (vi if 'V' in it else 0) + (ev if 'E' in it else 0) + (cl if 'C' in it else 0)
    if init_in:
        s['lm_model.init_linear.weight'] = (3 * H, init_in)
        s['lm_model.init_linear.bias'] = (3 * H,)
    for k, cin in ((0, ev + E), (1, cl + E), (2, vi + E)):
        s['lm_model.core.layer%d.weight_ih' % k] = (4 * H, cin)
        s['lm_model.core.layer%d.weight_hh' % k] = (4 * H, H)
        s['lm_model.core.layer%d.bias_ih' % k] = (4 * H,)
        s['lm_model.core.layer%d.bias_hh' % k] = (4 * H,)
    s.update({
        'lm_model.core.fusion_layer.weight': (H, 3 * H), 'lm_model.core.fusion_layer.bias': (H,),
        'lm_model.core.attention.ctx2att.weight': (Ha, D), 'lm_model.core.attention.ctx2att.bias': (Ha,),
        'lm_model.core.attention.h2att.weight': (Ha, H), 'lm_model.core.attention.h2att.bias': (Ha,),
        'lm_model.core.attention.alpha_net.weight': (1, Ha), 'lm_model.core.attention.alpha_net.bias': (1,),
    })
    return s


def _init_range(name, shapes, H):
    if name in ('lm_model.embed.weight', 'lm_model.logit.weight'):
        return 0.1                                    # OldModel_NEW.py:66-70
    if name == 'lm_model.logit.bias':
        return 0.0
    if '.core.layer' in name:
        return 1.0 / math.sqrt(H)                     # nn.LSTMCell default
    wname = name[:-len('.bias')] + '.weight' if name.endswith('.bias') else name
    fan_in = int(np.prod(shapes[wname][1:]))          # nn.Linear / Conv2d: bound = 1/sqrt(fan_in)
    return 1.0 / math.sqrt(fan_in)


def make_params(opt, seed=0):
    """Deterministic parameter set keyed by state-dict name (float32 numpy arrays).

    Filled in sorted-name order from one RandomState so that the reference-side golden tool and
    the build draw identical values without storing any weights."""
    rs = np.random.RandomState(seed)
    out = {}
    shapes = state_dict_shapes(opt)
    for name in sorted(shapes):
        shp = shapes[name]
        r = _init_range(name, shapes, opt.CG_rnn_size)
        out[name] = rs.uniform(-r, r, size=shp).astype(np.float32) if r > 0 else np.zeros(shp, np.float32)
    return out


def make_video(N, A, L, V1, seed=1234, T_v=None, full_len=False, min_len=4, video_dim=500, hidden_dim=512, lda_dim=100, disjoint=False):
    """One synthetic video: features, N events (half-open [s,e) segment intervals), captions.

    full_len=True makes every event exactly A segments long (the dense N x A x D case BASELINE
    quotes); otherwise lengths are uniform in [min_len, A] with at least one event of length A
    (SURVEY 8-d config 2).  Captions: column 0 = BOS (0), tokens in [1, V1), zero padded; at least
    one caption uses all L-2 token slots so the decoder runs the full L-1 steps.  disjoint=True lays the events out
    back to back on a T_v = N*A video (every event owns its own A feature rows: the dense [N x A x D] block).  The mask follows
    dataloader.py:437-439 (`nonzeros + 2` ones)."""
    rs = np.random.RandomState(seed)
    if disjoint:                      # N non-overlapping events of A segments: N*A distinct feature rows
        full_len, T_v = True, N * A
    if T_v is None:
        T_v = A + max(8, A // 4)
    lens = np.full(N, A, dtype=np.int64) if full_len else rs.randint(min(min_len, A), A + 1, size=N)
    lens[rs.randint(0, N)] = A
    starts = np.array([rs.randint(0, T_v - l + 1) for l in lens], dtype=np.int64)
    if disjoint:
        starts = np.arange(N, dtype=np.int64) * A
    soi = np.stack([starts, starts + lens], axis=1)
    ind = soi[:, 1] - 1                                   # proposal anchored at its last segment
    c3d = rs.standard_normal((T_v, video_dim)).astype(np.float32)
    tap = (0.5 * rs.standard_normal((T_v, hidden_dim))).astype(np.float32)
    lda = np.abs(rs.standard_normal(lda_dim)).astype(np.float32)
    lda /= lda.sum()
    labels = np.zeros((N, L), dtype=np.int64)
    masks = np.zeros((N, L), dtype=np.float32)
    max_tok = L - 2
    cap_len = rs.randint(1, max_tok + 1, size=N) if max_tok >= 1 else np.zeros(N, np.int64)
    if max_tok >= 1:
        cap_len[rs.randint(0, N)] = max_tok
    for i in range(N):
        labels[i, 1:1 + cap_len[i]] = rs.randint(1, V1, size=cap_len[i])
        masks[i, :cap_len[i] + 2] = 1.0
    return dict(c3d=c3d, tap=tap, lda=lda, labels=labels, masks=masks,
                ind=ind.astype(np.int64), soi=soi.astype(np.int64), T_v=T_v)


# Named parity / bench cases (BASELINE.json `configs`).  `opt` = overrides of default_opt.
CASES = {
    # every width shrunk (all multiples of 4) so that full tensors fit in a fixture
    'tiny': dict(opt=dict(video_dim=20, hidden_dim=24, lda_dim=12, d_feats=32, d_o=32, n_head=4, CG_rnn_size=32,
                          CG_input_encoding_size=16, CG_att_hid_size=24, CG_vocab_size=30, CG_seq_length=5),
                 video=dict(N=3, A=7, L=7, seed=11)),
    # config 1: N=2 events on a 16-segment video, 10 decoder steps, ECHR widths, V1 = 5001
    'c1': dict(opt=dict(CG_vocab_size=5000, CG_seq_length=9), video=dict(N=2, A=16, L=11, seed=21, T_v=16)),
    # config 2/3 shape with ragged event lengths in [4,128]
    'c2': dict(opt=dict(CG_vocab_size=5000, CG_seq_length=19), video=dict(N=64, A=128, L=21, seed=31)),
    # config 2/3 shape, every event 128 segments long (the dense batch 64 x 128 seg x 500-d bench workload)
    'c2full': dict(opt=dict(CG_vocab_size=5000, CG_seq_length=19), video=dict(N=64, A=128, L=21, seed=41, full_len=True)),
    # scene context from all three sources (CaptionGenerator.py:87-104): cat(lda, c3d.mean(0), tap.mean(0)), 100 + 500 + 512 wide
    'vctx': dict(opt=dict(video_context_type='VLVCVH', CG_vocab_size=300, CG_seq_length=7), video=dict(N=12, A=40, L=9, seed=61)),
    # event-context variants (CaptionGenerator.py:106-130): TSRM over the pooled C3D rows only / the anchors' SST states only
    'er1': dict(opt=dict(event_context_type='ER1', CG_vocab_size=300, CG_seq_length=7), video=dict(N=12, A=40, L=9, seed=62)),
    'er2': dict(opt=dict(event_context_type='ER2', CG_vocab_size=300, CG_seq_length=7), video=dict(N=12, A=40, L=9, seed=63)),
    # gate / affinity combinators of the event encoder (MA_attention_8_NEW.py:148-157) and the encoder without its position branch
    'fst1': dict(opt=dict(fST_type='fST1', CG_vocab_size=300, CG_seq_length=7), video=dict(N=12, A=40, L=9, seed=64)),
    # (fST2 takes log(clamp(gate, 1e-6)): a gate just above zero amplifies the documented 1-ulp deviation of the device's position embedding by
    # 1 / gate, so the case keeps its gates away from zero -- heads 0..7 shifted by +1 (gates 0.36 .. 1.73), heads 8..15 by -1 (all clamped))
    'fst2': dict(opt=dict(fST_type='fST2', CG_vocab_size=300, CG_seq_length=7), video=dict(N=12, A=40, L=9, seed=65), gate_shift=1.0),
    'fst3': dict(opt=dict(fST_type='fST3', CG_vocab_size=300, CG_seq_length=7), video=dict(N=12, A=40, L=9, seed=66)),
    'noposit': dict(opt=dict(use_posit=0, CG_vocab_size=300, CG_seq_length=7), video=dict(N=12, A=40, L=9, seed=67)),
    # non-zero initial decoder state (OldModel_NEW.py:79-96): h(-1) = c(-1) = init_linear(cat([scene | event | clip.mean(1)])), ragged events so
    # that the mean over the PADDED frame slots differs from a mean over each event's own rows
    'init': dict(opt=dict(CG_init_feats_type='VEC', CG_vocab_size=300, CG_seq_length=7), video=dict(N=12, A=40, L=9, seed=68)),
    'initc': dict(opt=dict(CG_init_feats_type='C', CG_vocab_size=300, CG_seq_length=7), video=dict(N=12, A=40, L=9, seed=69)),
    # EXACTLY the layout bench.py times (BASELINE config 3): 64 disjoint 128-segment events on a T_v = 8192 video
    'c3bench': dict(opt=dict(CG_vocab_size=5000, CG_seq_length=19), video=dict(N=64, A=128, L=21, seed=1234, disjoint=True)),
}


def sst_param_shapes(opt):
    """Name -> shape of SST.state_dict() (models/sst_model.py:12,22-23: nn.LSTM(video_dim, hidden_dim, 2 layers) + Linear(hidden_dim, K))."""
    H, D, K = opt.hidden_dim, opt.video_dim, opt.K
    s = {}
    for l, cin in ((0, D), (1, H)):
        s['rnn.weight_ih_l%d' % l] = (4 * H, cin)
        s['rnn.weight_hh_l%d' % l] = (4 * H, H)
        s['rnn.bias_ih_l%d' % l] = (4 * H,)
        s['rnn.bias_hh_l%d' % l] = (4 * H,)
    s['scores.weight'] = (K, H)
    s['scores.bias'] = (K,)
    return s


def make_sst_params(opt, seed=7):
    rs = np.random.RandomState(seed)
    r = 1.0 / math.sqrt(opt.hidden_dim)
    shapes = sst_param_shapes(opt)
    return {k: rs.uniform(-r, r, size=shapes[k]).astype(np.float32) for k in sorted(shapes)}


def make_c5(seed=51, N=64, T_v=256, L=21, V1=5001):
    """BASELINE config 5: ONE 256-segment video; SST proposal encoder over all 256 segments -> tap_feats -> caption path on N
    proposals of 4..256 segments (at least one spans the whole video), joint loss lambda1 * tap + lambda2 * cg (train.py:322-329).
    Returns (opt, caption params, SST params, video dict incl. the proposal-loss inputs)."""
    opt = default_opt(vocab_size=V1 - 1, seq_length=L - 2)
    opt.lambda1, opt.lambda2 = 0.01, 1.0                   # opts.py:194-196
    vid = make_video(N, T_v, L, V1, seed=seed, T_v=T_v)
    rs = np.random.RandomState(seed + 1000)
    vid['tap_labels'] = (rs.uniform(size=(T_v, opt.K)) > 0.9).astype(np.float32)
    vid['tap_masks'] = (np.arange(T_v)[:, None] >= np.arange(opt.K)[None, :]).astype(np.float32)
    vid['w1'] = rs.uniform(0.05, 0.3, size=(opt.K,)).astype(np.float32)
    return opt, make_params(opt, 0), make_sst_params(opt), vid


def make_case(name, param_seed=0):
    """(opt, params, video) of a named case."""
    c = CASES[name]
    opt = default_opt(**c['opt'])
    vid = make_video(V1=opt.CG_vocab_size + 1, video_dim=opt.video_dim, hidden_dim=opt.hidden_dim,
                     lda_dim=opt.lda_dim, **c['video'])
    params = make_params(opt, param_seed)
    if c.get('gate_shift'):
        b = params['fusion_model.enc_attn.pair_pos_fc2.bias']
        b[:len(b) // 2] += np.float32(c['gate_shift'])
        b[len(b) // 2:] -= np.float32(c['gate_shift'])
    return opt, params, vid


# Greedy-decoding cases whose captions END AT DIFFERENT STEPS (OldModel_NEW.py:171-183: per-row `unfinished` state).  With the plain
# synthetic initialisation a caption ends at its first word or never; here the token embedding is scaled up (the recurrences become
# token-driven, so the <eos> margin moves from step to step) and the <eos> row of the logit layer is widened.  The event lists of the
# fixtures (tests/golden/case_eosmix.npz: `soi`, `ind`) are selections / re-orderings of the video made here, chosen by
# tools/make_golden.py from the reference's own decode: 'b' puts the 64 earliest-finishing events first (one 64-event group is done
# long before the others), 'c' keeps only events that finish (the whole batch stops before seq_length).
EOSMIX = {
    'a': dict(N=64, seed=109, eos_scale=2.0),
    'b': dict(N=150, seed=322, eos_scale=2.5),
    'c': dict(N=150, seed=322, eos_scale=2.5),
}


def make_eosmix(name, A=24, embed_scale=10.0):
    """(opt, params, video) of a mixed-finish decoding case at the ECHR widths (V1 = 5001, seq_length 19)."""
    c = EOSMIX[name]
    opt = default_opt(vocab_size=5000, seq_length=19)
    params = make_params(opt, 0)
    params['lm_model.embed.weight'] = (params['lm_model.embed.weight'] * np.float32(embed_scale)).astype(np.float32)
    params['lm_model.logit.weight'][0] *= np.float32(c['eos_scale'])
    vid = make_video(c['N'], A, 21, 5001, seed=c['seed'], T_v=4 * A)
    return opt, params, vid


# Training in the PEAKED regime: a trained captioner's softmax puts > 0.9 of its mass on one word at most positions, so d logits =
# softmax - onehot spans dozens of binades inside one row (and inside one 256-wide k segment of the h2 operand format), which the
# random-initialisation cases (near-uniform softmax) never exercise.  Weights cannot be trained here and shipped (fixtures store no
# weights), so the regime is constructed: the logit layer is scaled up (every row's soft-max collapses onto its arg-max) and the token
# embedding too (token-driven recurrences: the peak moves from step to step); the CAPTIONS are then chosen so that most positions'
# target IS that arg-max -- tools/make_golden.py do_peaked runs the reference's own teacher-forced forward (train mode, the build's
# dropout masks) to its fixed point and stores labels / masks in tests/golden/case_peaked.npz; one row in ten keeps a random word
# (a confidently wrong prediction: loss terms of ~100, d logits of -1 / +1).
PEAKED = dict(N=64, A=128, L=21, seed=77, logit_scale=60.0, embed_scale=10.0, wrong_every=10)


def make_peaked(labels=None, masks=None):
    """(opt, params, video) of the peaked-softmax training case at the ECHR widths (N64 x A<=128 ragged, S = 20, V1 = 5001).  `labels` /
    `masks`: the fixture's captions (tests/golden/case_peaked.npz); without them the video carries random captions (what do_peaked starts from)."""
    c = PEAKED
    opt = default_opt(vocab_size=5000, seq_length=c['L'] - 2)
    params = make_params(opt, 0)
    params['lm_model.logit.weight'] = (params['lm_model.logit.weight'] * np.float32(c['logit_scale'])).astype(np.float32)
    params['lm_model.embed.weight'] = (params['lm_model.embed.weight'] * np.float32(c['embed_scale'])).astype(np.float32)
    vid = make_video(c['N'], c['A'], c['L'], 5001, seed=c['seed'])
    if labels is not None:
        vid['labels'], vid['masks'] = np.asarray(labels, np.int64).copy(), np.asarray(masks, np.float32).copy()
    return opt, params, vid
