#!/usr/bin/env python3
"""Generate golden fixtures from the REFERENCE's own code (build container only).

Imports /root/reference (never copied, never shipped), fills its modules with the deterministic
parameters of echr_amd.synth, runs its PyTorch-CPU path on the named synthetic cases and writes
small .npz fixtures under tests/golden/.  It also checks the repo's CPU oracle
(oracle/echr_ref_cpu.py) against the reference on the same inputs and prints the max deviations
(the oracle is "pinned" by this run + tests/test_oracle_golden.py).

Two in-process shims let the Python-2 / CUDA-only reference run under Python 3 on CPU; neither
touches a reference file:
  1. torch.Tensor.cuda -> identity            (MA_attention_8_NEW.py:41 hard-codes .cuda())
  2. torch.Tensor.view int-casts float sizes   (MA_attention_8_NEW.py:125,133 pass `d_q / group`)
Train-mode fixtures patch torch.nn.functional.dropout in-process so that the reference consumes
the build's counter-based (Philox) masks in its own call order.

Usage:  PYTHONDONTWRITEBYTECODE=1 python tools/make_golden.py [--cases tiny c1 c2 c2full]
"""
import argparse
import copy
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = '/root/reference'
sys.dont_write_bytecode = True
sys.path.insert(0, REF)

from echr_amd import philox, synth          # noqa: E402
from oracle import echr_ref_cpu as O        # noqa: E402
from oracle import summary as SM            # noqa: E402

# ---- shims -------------------------------------------------------------------------------------
torch.Tensor.cuda = lambda self, *a, **k: self
_orig_view = torch.Tensor.view


def _view(self, *shape):
    if len(shape) == 1 and isinstance(shape[0], (tuple, list, torch.Size)):
        shape = tuple(shape[0])
    shape = tuple(int(s) if isinstance(s, float) and float(s).is_integer() else s for s in shape)
    return _orig_view(self, *shape)


torch.Tensor.view = _view

import models                                # noqa: E402  (reference)
from CaptionGenerator import CaptionGenerator as RefCG   # noqa: E402  (reference)
import misc.utils as ref_utils               # noqa: E402  (reference)
import eval_utils as ref_eval                # noqa: E402  (reference)
import torch.nn.functional as F              # noqa: E402

GOLD = os.path.join(ROOT, 'tests', 'golden')
SEED, OFFSET = 0x5EED0123456789, 7


class MaskFeeder:
    """Replaces F.dropout: hands out Philox masks in the reference's call order:
    TSRM [N,G,N] once, then per step h0, h1, h2 [N,H] and out [N,3H]."""

    def __init__(self, p_out):
        self.calls = 0
        self.p_out = p_out

    def __call__(self, x, p=0.5, training=True, inplace=False):
        if not training:
            return x
        if self.calls == 0:
            site, step = philox.SITE_TSRM, 0
        else:
            k = self.calls - 1
            step, site = k // 4, (philox.SITE_H0, philox.SITE_H1, philox.SITE_H2, philox.SITE_OUT)[k % 4]
        self.calls += 1
        m = philox.scale_mask(tuple(x.shape), p, SEED, OFFSET, site, step)
        return x * torch.from_numpy(m)


def oracle_drop(opt):
    sites = dict(tsrm=(philox.SITE_TSRM, 0.3), h0=(philox.SITE_H0, 0.5), h1=(philox.SITE_H1, 0.5),
                 h2=(philox.SITE_H2, 0.5), out=(philox.SITE_OUT, opt.CG_drop_prob))

    def drop(site, step, shape):
        s, p = sites[site]
        return torch.from_numpy(philox.scale_mask(tuple(shape), p, SEED, OFFSET, s, step))
    return drop


def build_ref(opt, params):
    m = RefCG(copy.copy(opt))
    sd = {k: torch.from_numpy(v.copy()) for k, v in params.items()}
    missing = set(m.state_dict().keys()) ^ set(sd.keys())
    assert not missing, missing
    for k, v in m.state_dict().items():
        assert tuple(v.shape) == tuple(sd[k].shape), (k, v.shape, sd[k].shape)
    m.load_state_dict(sd)
    return m


def run_ref(m, vid, train_mode, opt):
    tap, c3d, lda = (torch.from_numpy(vid[k]) for k in ('tap', 'c3d', 'lda'))
    labels = torch.from_numpy(vid['labels'])
    masks = torch.from_numpy(vid['masks'])
    m.zero_grad()
    orig = F.dropout
    if train_mode:
        m.train()
        F.dropout = MaskFeeder(opt.CG_drop_prob)
    else:
        m.eval()
    try:
        pred = m(tap, c3d, lda, labels, vid['ind'], vid['soi'].tolist(), mode='train')
    finally:
        F.dropout = orig
    loss = ref_utils.LanguageModelCriterion()(pred, labels[:, 1:], masks[:, 1:])
    loss.backward()
    grads = {k: (p.grad.detach().numpy().copy() if p.grad is not None else None) for k, p in m.named_parameters()}
    return pred.detach().numpy(), float(loss), grads


def run_oracle(opt, params, vid, train_mode):
    P = {k: torch.from_numpy(v.copy()).requires_grad_(True) for k, v in params.items()}
    tap, c3d, lda = (torch.from_numpy(vid[k]) for k in ('tap', 'c3d', 'lda'))
    labels = torch.from_numpy(vid['labels'])
    masks = torch.from_numpy(vid['masks'])
    drop = oracle_drop(opt) if train_mode else None
    pred = O.caption_forward(P, tap, c3d, lda, labels, vid['ind'], vid['soi'], 'train', drop, opt.n_head, video_context_type=opt.video_context_type, event_context_type=opt.event_context_type, fST_type=getattr(opt, 'fST_type', 'fST0'), use_posit=opt.use_posit, init_feats_type=opt.CG_init_feats_type)
    loss = O.lm_criterion(pred, labels[:, 1:], masks[:, 1:])
    loss.backward()
    grads = {k: (p.grad.numpy().copy() if p.grad is not None else None) for k, p in P.items()}
    return pred.detach().numpy(), float(loss), grads


def rel(a, b):
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


def do_case(name):
    opt, params, vid = synth.make_case(name)
    full = name == 'tiny'
    out = {}
    m = build_ref(opt, params)
    for mode in ('eval', 'train'):
        t0 = time.time()
        pred, loss, grads = run_ref(m, vid, mode == 'train', opt)
        t_ref = time.time() - t0
        opred, oloss, ograds = run_oracle(opt, params, vid, mode == 'train')
        dev = max(rel(ograds[k], grads[k]) for k in grads if grads[k] is not None)
        unused = sorted(k for k in grads if grads[k] is None)
        assert unused == sorted(k for k in ograds if ograds[k] is None), (unused,)
        print('[%s/%s] ref %.2fs loss %.6f | oracle-vs-ref: max|dlogp| %.2e  dloss %.2e  max rel grad %.2e | unused %s'
              % (name, mode, t_ref, loss, np.abs(opred - pred).max(), abs(oloss - loss), dev, unused))
        assert np.abs(opred - pred).max() < 2e-5 and abs(oloss - loss) < 1e-5 and dev < 1e-4
        out[mode + '|loss'] = np.float64(loss)
        if full:
            out[mode + '|logp'] = pred
            for k, g in grads.items():
                if g is not None:
                    out[mode + '|grad|' + k] = g
        else:
            for k, v in SM.summarize_logp(pred).items():
                out[mode + '|logp|' + k] = v
            for k, v in SM.summarize_grads(grads).items():
                out[mode + '|grad|' + k] = v
    # greedy sampling (eval mode), index output bit-exact
    m.eval()
    tap, c3d, lda = (torch.from_numpy(vid[k]) for k in ('tap', 'c3d', 'lda'))
    import io
    import contextlib
    with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
        seq, slp = m(tap, c3d, lda, [], vid['ind'], vid['soi'].tolist(), mode='eval')
    P = {k: torch.from_numpy(v) for k, v in params.items()}
    with torch.no_grad():
        oseq, oslp = O.caption_forward(P, tap, c3d, lda, None, vid['ind'], vid['soi'], 'eval', None, opt.n_head,
                                       opt.CG_seq_length, video_context_type=opt.video_context_type, event_context_type=opt.event_context_type, fST_type=getattr(opt, 'fST_type', 'fST0'), use_posit=opt.use_posit, init_feats_type=opt.CG_init_feats_type)
    assert torch.equal(seq, oseq), 'oracle greedy seq differs'
    print('[%s/sample] seq %s  oracle max|dlogp| %.2e' % (name, tuple(seq.shape), float((slp - oslp).abs().max())))
    out['sample|seq'] = seq.numpy().astype(np.int64)
    out['sample|logp'] = slp.numpy()
    # event context / TSRM output in eval mode
    with torch.no_grad():
        ev = m.get_event_context(tap, c3d, lda, vid['ind'], vid['soi'].tolist())
    out['event_context'] = ev.numpy()
    np.savez_compressed(os.path.join(GOLD, 'case_%s.npz' % name), **out)
    print('  wrote case_%s.npz (%d arrays)' % (name, len(out)))


def do_c5():
    """BASELINE config 5 as ONE unit: the reference's SST (eval: no inter-layer dropout, sst_model.py:25-26) over a 256-segment video
    -> tap_feats -> reference CaptionGenerator on 64 proposals of 4..256 segments -> lambda1 * tap_loss + lambda2 * cg_loss
    (train.py:322-329) -> backward into BOTH models.  Caption model in eval and in train mode (injected Philox masks)."""
    opt, params, sst_params, vid = synth.make_c5()
    m = build_ref(opt, params)
    tapm = models.setup_tap(copy.copy(opt))
    tapm.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sst_params.items()})
    tapm.eval()
    c3d, lda = torch.from_numpy(vid['c3d']), torch.from_numpy(vid['lda'])
    labels, masks = torch.from_numpy(vid['labels']), torch.from_numpy(vid['masks'])
    tl, tm, tw = (torch.from_numpy(vid[k]) for k in ('tap_labels', 'tap_masks', 'w1'))
    out = {}
    for mode in ('eval', 'train'):
        m.zero_grad()
        tapm.zero_grad()
        orig = F.dropout
        if mode == 'train':
            m.train()
            F.dropout = MaskFeeder(opt.CG_drop_prob)
        else:
            m.eval()
        t0 = time.time()
        try:
            tap_feats, props = tapm(c3d)
            pred = m(tap_feats, c3d, lda, labels, vid['ind'], vid['soi'].tolist(), mode='train')
        finally:
            F.dropout = orig
        tap_loss = ref_utils.TAPModelCriterion()(props, tm, tl, tw)
        cg_loss = ref_utils.LanguageModelCriterion()(pred, labels[:, 1:], masks[:, 1:])
        total = opt.lambda1 * tap_loss + opt.lambda2 * cg_loss
        total.backward()
        t_ref = time.time() - t0
        grads = {k: (p.grad.detach().numpy().copy() if p.grad is not None else None) for k, p in m.named_parameters()}
        sgrads = {k: p.grad.detach().numpy().copy() for k, p in tapm.named_parameters()}
        # the oracle on the same joint path
        P = {k: torch.from_numpy(v.copy()).requires_grad_(True) for k, v in params.items()}
        SP = {k: torch.from_numpy(v.copy()).requires_grad_(True) for k, v in sst_params.items()}
        otap, oprops = O.sst_forward(SP, c3d)
        opred = O.caption_forward(P, otap, c3d, lda, labels, vid['ind'], vid['soi'], 'train', oracle_drop(opt) if mode == 'train' else None, opt.n_head)
        ototal = opt.lambda1 * O.tap_criterion(oprops, tm, tl, tw) + opt.lambda2 * O.lm_criterion(opred, labels[:, 1:], masks[:, 1:])
        ototal.backward()
        # alpha_net.bias: the true gradient is exactly zero (softmax shift invariance); both sides hold rounding noise
        dev = max(rel(P[k].grad.numpy(), grads[k]) for k in grads if grads[k] is not None and not k.endswith('alpha_net.bias'))
        sdev = max(rel(SP[k].grad.numpy(), sgrads[k]) for k in sgrads)
        print('[c5/%s] ref %.2fs tap_loss %.6f cg_loss %.6f | oracle-vs-ref: max|dlogp| %.2e dtotal %.2e max rel grad %.2e (cg) %.2e (sst)'
              % (mode, t_ref, float(tap_loss), float(cg_loss), float((opred - pred).abs().max()), abs(float(ototal) - float(total)), dev, sdev))
        assert float((opred - pred).abs().max()) < 2e-5 and dev < 1e-4 and sdev < 1e-4
        out[mode + '|tap_loss'] = np.float64(float(tap_loss))
        out[mode + '|cg_loss'] = np.float64(float(cg_loss))
        out[mode + '|total'] = np.float64(float(total))
        for k, v in SM.summarize_logp(pred.detach().numpy()).items():
            out[mode + '|logp|' + k] = v
        for k, v in SM.summarize_grads(grads).items():
            out[mode + '|grad|' + k] = v
        for k, v in SM.summarize_grads(sgrads).items():
            out[mode + '|sstgrad|' + k] = v
        if mode == 'eval':
            tf = tap_feats.detach().numpy()
            out['tap_feats|slice'] = tf[::8, ::16].copy()
            out['props|slice'] = props.detach().numpy()[::8, ::16].copy()
    np.savez_compressed(os.path.join(GOLD, 'case_c5.npz'), **out)
    print('  wrote case_c5.npz (%d arrays)' % len(out))


def do_peaked():
    """Training parity in the peaked-softmax regime (synth.PEAKED): captions = the fixed point of the REFERENCE's own teacher-forced arg-max
    under the build's train-mode dropout masks (step t's output depends on labels[:, <= t] only, so S sweeps reach it), one row in ten with a
    random (confidently wrong) word; then the usual eval / train summaries of the reference on those captions, and the oracle checked
    against it."""
    opt, params, vid = synth.make_peaked()
    c = synth.PEAKED
    m = build_ref(opt, params)
    labels = vid['labels'].copy()
    N, L = labels.shape
    rs = np.random.RandomState(c['seed'] + 1)
    cap_len = (labels[:, 1:] != 0).sum(1)                 # the random captions' lengths are kept
    wrong = rs.randint(1, 5001, size=labels.shape)
    is_wrong = rs.randint(0, c['wrong_every'], size=labels.shape) == 0
    tap, c3d, lda = (torch.from_numpy(vid[k]) for k in ('tap', 'c3d', 'lda'))
    for sweep in range(L):
        m.train()
        orig = F.dropout
        F.dropout = MaskFeeder(opt.CG_drop_prob)
        try:
            with torch.no_grad():
                pred = m(tap, c3d, lda, torch.from_numpy(labels), vid['ind'], vid['soi'].tolist(), mode='train').numpy()
        finally:
            F.dropout = orig
        new = labels.copy()
        S = pred.shape[1]
        pp = pred.copy()
        pp[:, :, 0] = -np.inf                            # a word inside a caption is never <eos>
        am = pp.argmax(2)                                 # [N, S]: the model's word for position t + 1
        for n in range(N):
            for t in range(min(S, cap_len[n])):
                new[n, t + 1] = wrong[n, t + 1] if is_wrong[n, t + 1] else am[n, t]
        changed = int((new != labels).sum())
        labels = new
        print('[peaked] sweep %d: %d labels changed' % (sweep, changed))
        if changed == 0:
            break
    assert changed == 0, 'no fixed point'
    vid['labels'] = labels
    out = {'labels': labels, 'masks': vid['masks']}
    for mode in ('eval', 'train'):
        pred, loss, grads = run_ref(m, vid, mode == 'train', opt)
        opred, oloss, ograds = run_oracle(opt, params, vid, mode == 'train')
        dev = max(rel(ograds[k], grads[k]) for k in grads if grads[k] is not None)
        act = vid['masks'][:, 1:1 + pred.shape[1]] > 0
        p1 = np.exp(pred.max(2))[act]
        tgt = np.take_along_axis(pred, labels[:, 1:1 + pred.shape[1], None], 2)[:, :, 0]
        print('[peaked/%s] loss %.6f | top-1 prob > 0.9 on %.0f %% of the active rows, target prob > 0.9 on %.0f %%, min logp %.1f | oracle-vs-ref: '
              'max|dlogp| %.2e  dloss %.2e  max rel grad %.2e' % (mode, loss, 100 * (p1 > 0.9).mean(), 100 * (np.exp(tgt)[act] > 0.9).mean(),
                                                                 pred.min(), np.abs(opred - pred).max(), abs(oloss - loss), dev))
        assert np.abs(opred - pred).max() < 1e-6 * np.abs(pred).max() + 2e-5 and abs(oloss - loss) < 1e-5 * abs(loss) and dev < 1e-4
        out[mode + '|loss'] = np.float64(loss)
        out[mode + '|top1_gt_0.9'] = np.float64((p1 > 0.9).mean())
        for k, v in SM.summarize_logp(pred).items():
            out[mode + '|logp|' + k] = v
        for k, v in SM.summarize_grads(grads).items():
            out[mode + '|grad|' + k] = v
    np.savez_compressed(os.path.join(GOLD, 'case_peaked.npz'), **out)
    print('  wrote case_peaked.npz (%d arrays)' % len(out))


def do_eosmix():
    """Greedy decoding where events finish at DIFFERENT steps (OldModel_NEW.py:171-183): the reference's own `seq` / `seqLogprobs` on
    synth.make_eosmix inputs.  The event lists are chosen here from the reference's decode and stored in the fixture."""
    import io
    import contextlib
    out = {}

    def decode(m, vid, soi, ind):
        tap, c3d, lda = (torch.from_numpy(vid[k]) for k in ('tap', 'c3d', 'lda'))
        with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
            return m(tap, c3d, lda, [], ind, soi.tolist(), mode='eval')

    def raw_decode(m, vid, soi, ind, S):
        """The un-masked arg-max chain of every row over all S+1 steps (what the network consumes, :158,171) + top-1/top-2 margins."""
        tap, c3d, lda = (torch.from_numpy(vid[k]) for k in ('tap', 'c3d', 'lda'))
        with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
            video = m.get_video_context(tap, c3d, lda, ind, soi.tolist())
            event = m.get_event_context(tap, c3d, lda, ind, soi.tolist())
            clip, mask = m.get_clip_context(tap, c3d, lda, ind, soi.tolist())
            lm = m.lm_model
            state = lm.init_hidden(video, event, clip)
            it = torch.zeros(len(soi), dtype=torch.long)
            toks, margins = [], []
            for t in range(S + 1):
                lp, state = lm.get_logprobs_state(it, video, event, clip, mask, state)
                top = lp.topk(2, dim=1)
                it = top.indices[:, 0]
                toks.append(it.numpy())
                margins.append((top.values[:, 0] - top.values[:, 1]).numpy())
        return np.stack(toks, 1), np.stack(margins, 1)

    for name in ('a', 'b', 'c'):
        opt, params, vid = synth.make_eosmix(name)
        m = build_ref(opt, params)
        m.eval()
        soi, ind = vid['soi'], vid['ind']
        S = opt.CG_seq_length
        if name in ('b', 'c'):
            toks, _ = raw_decode(m, vid, soi, ind, S)
            fin = np.array([(np.nonzero(r == 0)[0][0] if (r == 0).any() else 99) for r in toks])
            order = np.argsort(fin, kind='stable')
            if name == 'b':          # the 64 earliest finishers first, the rest in their original order
                first = order[:64]
                rest = np.array([i for i in range(len(soi)) if i not in set(first.tolist())])
                sel = np.concatenate([first, rest])
            else:                    # only events that finish, re-checked below on the smaller batch (the event encoder sees the set)
                sel = order[:48]
            soi, ind = soi[sel], ind[sel]
        toks, margins = raw_decode(m, vid, soi, ind, S)
        fin = np.array([(np.nonzero(r == 0)[0][0] if (r == 0).any() else 99) for r in toks])
        seq, slp = decode(m, vid, soi, ind)
        P = {k: torch.from_numpy(v) for k, v in params.items()}
        with torch.no_grad():
            oseq, oslp = O.caption_forward(P, torch.from_numpy(vid['tap']), torch.from_numpy(vid['c3d']), torch.from_numpy(vid['lda']), None,
                                           ind, soi, 'eval', None, opt.n_head, S)
        assert torch.equal(seq, oseq), 'oracle greedy seq differs'
        assert float((slp - oslp).abs().max()) < 2e-5
        T = seq.shape[1]
        nz = seq.numpy() == 0
        print('[eosmix/%s] N %d seq %s | finish steps: %d distinct, never %d, min margin %.2e | zeros %d | oracle max|dlogp| %.2e'
              % (name, len(soi), tuple(seq.shape), len(set(fin.tolist())), int((fin == 99).sum()), margins[:, :T + 1].min(), int(nz.sum()),
                 float((slp - oslp).abs().max())))
        assert len(set(fin.tolist())) >= 3 and margins[:, :T + 1].min() > 2e-5
        if name == 'a':
            assert (fin == 99).any() and T == S
        if name == 'b':
            g0 = fin[:64].max()
            print('   group 0 done after step %d, groups 1/2 run to %d' % (g0, T))
            assert g0 + 3 < T and (fin[64:] == 99).any()
        if name == 'c':
            assert (fin < 99).all() and T < S, (fin, T)
        out[name + '|soi'] = soi.astype(np.int64)
        out[name + '|ind'] = ind.astype(np.int64)
        out[name + '|seq'] = seq.numpy().astype(np.int64)
        out[name + '|logp'] = slp.numpy()
        out[name + '|min_margin'] = np.float64(margins[:, :T + 1].min())
    np.savez_compressed(os.path.join(GOLD, 'case_eosmix.npz'), **out)
    print('wrote case_eosmix.npz')


def do_position():
    RefMA = models.MA_Attention8
    out = {}
    sois = {'a': np.array([[0, 9], [4, 16], [2, 3], [10, 138]]),
            'b': synth.make_video(12, 40, 5, 10, seed=5)['soi']}
    for k, soi in sois.items():
        pm = RefMA.extract_position_matrix(np.array(soi.tolist()), len(soi))
        pe = RefMA.extract_position_embedding(pm, 512)
        assert np.array_equal(pm, O.position_matrix(soi)) and np.array_equal(pe, O.position_embedding(pm, 512))
        out[k + '|soi'] = soi
        out[k + '|pos_matrix'] = pm
        out[k + '|pos_emb_f32'] = pe.astype(np.float32)
        pe32 = RefMA.extract_position_embedding(pm, 32)
        out[k + '|pos_emb32_f32'] = pe32.astype(np.float32)
    np.savez_compressed(os.path.join(GOLD, 'position.npz'), **out)
    print('wrote position.npz')


def do_adam():
    """clip_gradient(+-clip) + torch.optim.Adam exactly as train.py:209,315-317 wires them."""
    rs = np.random.RandomState(3)
    p0 = rs.standard_normal(4097).astype(np.float32)
    gs = [(rs.standard_normal(4097) * (150.0 if i == 1 else 1.0)).astype(np.float32) for i in range(4)]
    p = torch.nn.Parameter(torch.from_numpy(p0.copy()))
    opt = torch.optim.Adam([p], lr=5e-5, betas=(0.9, 0.999), eps=1e-8, weight_decay=0)
    out = {'p0': p0}
    for i, g in enumerate(gs):
        p.grad = torch.from_numpy(g.copy())
        ref_utils.clip_gradient(opt, 100.0)
        opt.step()
        out['g%d' % i] = g
        out['p%d' % (i + 1)] = p.detach().numpy().copy()
    st = opt.state[p]
    out['exp_avg'] = st['exp_avg'].numpy().copy()
    out['exp_avg_sq'] = st['exp_avg_sq'].numpy().copy()
    # oracle check
    po, mo, vo = p0.copy(), np.zeros_like(p0), np.zeros_like(p0)
    for i, g in enumerate(gs):
        O.clamp_adam_step(po, g.copy(), mo, vo, i + 1, 5e-5)
    print('[adam] oracle-vs-torch max|dp| %.2e' % np.abs(po - out['p4']).max())
    assert np.abs(po - out['p4']).max() < 1e-6
    np.savez_compressed(os.path.join(GOLD, 'adam.npz'), **out)
    print('wrote adam.npz')


def do_proposals():
    """gettop1000 index outputs for seeded score grids (eval_utils.py:259-287)."""
    out = {}
    for i, (T, K, topN) in enumerate(((40, 16, 50), (96, 64, 100), (30, 64, 1000))):
        rs = np.random.RandomState(100 + i)
        scores = rs.uniform(0, 1, size=(T, K)).astype(np.float32)
        tmask = np.tril(np.ones((T, K), np.float32))[:, :K] if T >= K else np.ones((T, K), np.float32)
        tmask = (np.arange(T)[:, None] >= np.arange(K)[None, :]).astype(np.float32)
        ind, feat, _, _, conf = ref_eval.gettop1000(scores, tmask, [], 100.0, lambda s, e, n, d: [s, e], topN=topN)
        oind, ofeat, oconf = O.top_proposals(scores, tmask, topN)
        assert ind == oind and feat == ofeat and np.allclose(conf, oconf)
        out['g%d|scores' % i] = scores
        out['g%d|mask' % i] = tmask
        out['g%d|topN' % i] = np.int64(topN)
        out['g%d|ind' % i] = np.array(ind, np.int64)
        out['g%d|feat' % i] = np.array(feat, np.int64)
    # gettop1000_nms (eval_utils.py:290-331): unique scores, so the reference's unstable argsort is deterministic
    for i, (T, K, topN, ov) in enumerate(((40, 16, 50, 0.8), (96, 64, 100, 0.5), (64, 64, 1000, 0.9))):
        rs = np.random.RandomState(200 + i)
        scores = rs.permutation(T * K).reshape(T, K).astype(np.float32) / np.float32(T * K)
        ind, props, _, ts, sc = ref_eval.gettop1000_nms(scores, None, [], 100.0, lambda s, e, n, d: [s, e], overlap=ov, topN=topN)
        pick, oprops, osc = O.top_proposals_nms(scores, ov, topN)
        assert np.array_equal(props, oprops) and np.array_equal(sc, osc) and np.array_equal(ind, oprops[:, 1] - 1)
        out['n%d|scores' % i] = scores
        out['n%d|topN' % i] = np.int64(topN)
        out['n%d|overlap' % i] = np.float64(ov)
        out['n%d|props' % i] = np.asarray(props, np.int64)
        out['n%d|conf' % i] = np.asarray(sc, np.float64)
    # gettopN_nms (eval_utils.py:230-256) on materialised proposals and reranking (:334-345); distinct scores keep the reference's
    # (unstable) argsort deterministic
    for i, (P, thr, topN) in enumerate(((40, 0.5, 1000), (200, 0.8, 25), (64, 0.999, 1000))):
        rs = np.random.RandomState(300 + i)
        t1 = rs.randint(0, 80, size=P)
        props = np.stack([t1, t1 + rs.randint(1, 40, size=P)], axis=1).astype(np.float64)
        psc = (rs.permutation(P).astype(np.float64) + 1.0) / P
        ssc = (rs.permutation(P).astype(np.float64) + 1.0) / P
        rp, rsco, pick = ref_eval.gettopN_nms(props, psc, ssc, nms_overlap=thr, topN=topN)
        assert [int(x) for x in pick] == O.topn_nms(props, psc, ssc, thr, topN)
        out['m%d|props' % i] = props
        out['m%d|pscore' % i] = psc
        out['m%d|sscore' % i] = ssc
        out['m%d|thr' % i] = np.float64(thr)
        out['m%d|topN' % i] = np.int64(topN)
        out['m%d|pick' % i] = np.asarray(pick, np.int64)
    for i, n in enumerate((4, 10, 37)):
        rs = np.random.RandomState(400 + i)
        info = [{'re_score': float(x), 'id': j} for j, x in enumerate(rs.permutation(n) / float(n))]
        kept = ref_eval.reranking(info)
        assert [v['id'] for v in kept] == [v['id'] for v in O.rerank(info)]
        out['r%d|scores' % i] = np.array([v['re_score'] for v in info])
        out['r%d|kept' % i] = np.array([v['id'] for v in kept], np.int64)
    np.savez_compressed(os.path.join(GOLD, 'proposals.npz'), **out)
    print('wrote proposals.npz')


def do_sst():
    """The reference's SST (nn.LSTM + head, eval mode) and TAPModelCriterion on seeded inputs; also pins the oracle's hand-written
    LSTM restatement against them."""
    opt = synth.default_opt(**synth.CASES['tiny']['opt'], K=8)
    rs = np.random.RandomState(77)
    T, D, H, K = 9, opt.video_dim, opt.hidden_dim, opt.K
    torch.manual_seed(5)
    m = models.setup_tap(copy.copy(opt))
    sd = {k: torch.from_numpy(rs.uniform(-0.3, 0.3, size=tuple(v.shape)).astype(np.float32)) for k, v in m.state_dict().items()}
    m.load_state_dict(sd)
    m.eval()
    x = torch.from_numpy(rs.standard_normal((T, D)).astype(np.float32))
    masks = torch.from_numpy((np.arange(T)[:, None] >= np.arange(K)[None, :]).astype(np.float32))
    labels = torch.from_numpy((rs.uniform(size=(T, K)) > 0.7).astype(np.float32))
    w1 = torch.from_numpy(rs.uniform(0.05, 0.4, size=(K,)).astype(np.float32))
    tap, sc = m(x)
    loss = ref_utils.TAPModelCriterion()(sc, masks, labels, w1) + 0.1 * (tap * tap).sum()
    loss.backward()
    out = {'x': x.numpy(), 'masks': masks.numpy(), 'labels': labels.numpy(), 'w1': w1.numpy(), 'tap': tap.detach().numpy(),
           'scores': sc.detach().numpy(), 'loss': np.float64(loss.item())}
    for k, v in sd.items():
        out['param|' + k] = v.numpy()
    for k, p in m.named_parameters():
        out['grad|' + k] = p.grad.numpy().copy()
    P = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    otap, osc = O.sst_forward(P, x)
    oloss = O.tap_criterion(osc, masks, labels, w1) + 0.1 * (otap * otap).sum()
    oloss.backward()
    dev = max(rel(P[k].grad.numpy(), out['grad|' + k]) for k in sd)
    print('[sst] oracle-vs-ref: max|dtap| %.2e max|dscore| %.2e dloss %.2e max rel grad %.2e'
          % ((otap - tap).abs().max(), (osc - sc).abs().max(), abs(oloss.item() - loss.item()), dev))
    assert (otap - tap).abs().max() < 1e-6 and dev < 1e-4
    np.savez_compressed(os.path.join(GOLD, 'sst.npz'), **out)
    print('wrote sst.npz')


def do_checkpoint():
    """A checkpoint written by the REFERENCE's own modules (its own random init) in train.py's dict layout (:456-461), plus
    the reference's eval-mode log-probs for those weights on the 'tiny' inputs: the build must load it as is."""
    opt, _, vid = synth.make_case('tiny')
    torch.manual_seed(123)
    m = RefCG(copy.copy(opt))
    tap = models.setup_tap(copy.copy(opt))
    torch.save({'iteration': 7, 'cg_model': m.state_dict(), 'tap_model': tap.state_dict()}, os.path.join(GOLD, 'ref_tiny_checkpoint.pth'))
    m.eval()
    with torch.no_grad():
        pred = m(torch.from_numpy(vid['tap']), torch.from_numpy(vid['c3d']), torch.from_numpy(vid['lda']), torch.from_numpy(vid['labels']),
                 vid['ind'], vid['soi'].tolist(), mode='train')
    np.savez_compressed(os.path.join(GOLD, 'ref_tiny_checkpoint_out.npz'), logp=pred.numpy())
    print('wrote ref_tiny_checkpoint.pth / ref_tiny_checkpoint_out.npz')


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--cases', nargs='*', default=['tiny', 'c1', 'c2', 'c2full', 'c3bench'])
    ap.add_argument('--skip-aux', action='store_true')
    ap.add_argument('--only', choices=['position', 'adam', 'proposals', 'checkpoint', 'sst', 'c5', 'eosmix', 'peaked'], help='regenerate one auxiliary fixture only')
    a = ap.parse_args()
    os.makedirs(GOLD, exist_ok=True)
    torch.manual_seed(0)
    if a.only:
        {'position': do_position, 'adam': do_adam, 'proposals': do_proposals, 'checkpoint': do_checkpoint, 'sst': do_sst, 'c5': do_c5, 'eosmix': do_eosmix, 'peaked': do_peaked}[a.only]()
        sys.exit(0)
    if not a.skip_aux:
        do_position()
        do_adam()
        do_proposals()
        do_checkpoint()
        do_sst()
        do_c5()
        do_eosmix()
        do_peaked()
    for c in a.cases:
        do_case(c)
