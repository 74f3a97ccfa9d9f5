"""Shared helpers of the parity tests (the oracle is only ever the checker)."""
import os

import numpy as np
import torch

from echr_amd import philox, synth

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
SEED, OFFSET = 0x5EED0123456789, 7          # the dropout stream tools/make_golden.py used


def gold(name):
    return dict(np.load(os.path.join(GOLD, name), allow_pickle=False))


def oracle_drop(opt):
    sites = dict(tsrm=(philox.SITE_TSRM, 0.3), h0=(philox.SITE_H0, 0.5), h1=(philox.SITE_H1, 0.5),
                 h2=(philox.SITE_H2, 0.5), out=(philox.SITE_OUT, opt.CG_drop_prob))

    def drop(site, step, shape):
        s, p = sites[site]
        return torch.from_numpy(philox.scale_mask(tuple(shape), p, SEED, OFFSET, s, step))
    return drop


def run_oracle(opt, params, vid, train_mode, backward=True):
    from oracle import echr_ref_cpu as O
    P = {k: torch.from_numpy(v.copy()).requires_grad_(backward) for k, v in params.items()}
    tap, c3d, lda = (torch.from_numpy(vid[k]) for k in ('tap', 'c3d', 'lda'))
    labels = torch.from_numpy(vid['labels'])
    masks = torch.from_numpy(vid['masks'])
    drop = oracle_drop(opt) if train_mode else None
    pred = O.caption_forward(P, tap, c3d, lda, labels, vid['ind'], vid['soi'], 'train', drop, opt.n_head, video_context_type=opt.video_context_type, event_context_type=opt.event_context_type, fST_type=getattr(opt, 'fST_type', 'fST0'), use_posit=opt.use_posit,
                             init_feats_type=opt.CG_init_feats_type)
    loss = O.lm_criterion(pred, labels[:, 1:], masks[:, 1:])
    grads = None
    if backward:
        loss.backward()
        grads = {k: (p.grad.numpy().copy() if p.grad is not None else None) for k, p in P.items()}
    return pred.detach().numpy(), float(loss.detach()), grads


def build_gpu_model(opt, params, train_mode):
    import echr_amd
    m = echr_amd.CaptionGenerator(opt)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()})
    m = m.cuda()
    m.train(train_mode)
    m.set_dropout_state(SEED, OFFSET)
    return m


def run_gpu(opt, params, vid, train_mode, backward=True):
    from echr_amd.misc.utils import LanguageModelCriterion
    m = build_gpu_model(opt, params, train_mode)
    dev = torch.device('cuda')
    tap, c3d, lda = (torch.from_numpy(vid[k]).to(dev) for k in ('tap', 'c3d', 'lda'))
    labels = torch.from_numpy(vid['labels'])
    masks = torch.from_numpy(vid['masks'])
    pred = m(tap, c3d, lda, labels, vid['ind'], vid['soi'], mode='train')
    loss = LanguageModelCriterion()(pred, labels[:, 1:].to(dev), masks[:, 1:].to(dev))
    grads = None
    if backward:
        loss.backward()
        grads = {k: (p.grad.detach().cpu().numpy() if p.grad is not None else None) for k, p in m.named_parameters()}
    torch.cuda.synchronize()
    return pred.detach().cpu().numpy(), float(loss.detach()), grads, m


def relerr(a, b, floor=0.0):
    """max|a-b| relative to the reference tensor's max-norm (`floor` = scale below which a tensor counts as numerical zero)."""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), floor, 1e-30))


# d loss / d alpha_net.bias is exactly zero in real arithmetic (softmax is shift invariant); the reference's value is
# rounding noise (1e-10..1e-9).  Gradient tensors below this max-norm are compared on an absolute scale.
GRAD_FLOOR = 1e-5


NOISE_ONLY = ('lm_model.core.attention.alpha_net.bias',)


def grad_close(name, a, b, tol):
    """Gradient comparison relative to the reference tensor's max-norm; parameters whose true gradient is exactly zero
    (NOISE_ONLY) only need to stay at rounding-noise level."""
    if name in NOISE_ONLY or float(np.abs(np.asarray(b)).max()) == 0.0:
        # exact zeros in the reference (e.g. single-slot events: attention weights are identically 1) vs rounding noise here
        return float(np.abs(np.asarray(a)).max()) < 1e-6 and float(np.abs(np.asarray(b)).max()) < 1e-6
    # 1e-9 absolute: tensors whose true gradient vanishes (e.g. pair_pos_fc2.bias with a single event: ~1e-11 on both sides) are rounding noise
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max()) <= tol * max(float(np.abs(b).max()), GRAD_FLOOR) + 1e-9
