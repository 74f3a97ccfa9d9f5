#!/usr/bin/env python3
"""Reference-shaped training / evaluation driver on synthetic videos.

Follows the iteration protocol of the reference's train.py (:253-331) with the same module calls:
  tap_feats, pred_proposals = tap_model(c3d_feats)                                   (:285)
  pred = cg_model(tap_feats, c3d_feats, lda_feats, cg_labels, ind, soi, mode='train') (:298)
  cg_loss = LanguageModelCriterion()(pred, labels[:, 1:], masks[:, 1:])               (:300)
  total = lambda1 * tap_loss + lambda2 * cg_loss  ('tap_cg' joint mode, :322-329), backward,
  clip_gradient + optimizer.step every m_batch videos (:281-283,313-317), step LR decay (:232-240),
and saves checkpoints in the reference's dict layout (:456-461) so that either code base can resume the other's run.
Both models run natively (echr_amd's SST: echr_sst_fwd/bwd; the caption path: echr_tsrm_* / echr_decoder_*), both optimisers are
the fused ClampAdam, and `--resume` restores models AND optimiser state (train.py:214-216).

usage: python examples/train_synthetic.py [--iters 20] [--m_batch 2] [--joint] [--save /tmp/echr_ckpt.pth] [--resume /tmp/echr_ckpt.pth]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import echr_amd
from echr_amd import models, synth
from echr_amd.misc import utils
from echr_amd.optim import ClampAdam


def make_loader(opt, n_videos, N, A, L, seed=0):
    V1 = opt.CG_vocab_size + 1
    vids = [synth.make_video(N, A, L, V1, seed=seed + i, video_dim=opt.video_dim, hidden_dim=opt.hidden_dim, lda_dim=opt.lda_dim)
            for i in range(n_videos)]
    rs = np.random.RandomState(seed)
    for v in vids:       # proposal labels / masks / class weights of the SST head (shapes as dataloader.py:320-365 produces them)
        T = v['T_v']
        v['tap_labels'] = (rs.uniform(size=(T, opt.K)) > 0.9).astype(np.float32)
        v['tap_masks'] = (np.arange(T)[:, None] >= np.arange(opt.K)[None, :]).astype(np.float32)
        v['w1'] = rs.uniform(0.05, 0.3, size=(opt.K,)).astype(np.float32)
    return vids


def set_lr_for_epoch(optimizer, base_lr, epoch, start=8, every=3, rate=0.5):
    """Step decay as train.py:232-240."""
    lr = base_lr if epoch <= start or start < 0 else base_lr * rate ** ((epoch - start) // every)
    utils.set_lr(optimizer, lr)
    return lr


def save_checkpoint(path, iteration, cg_model, tap_model, cg_opt, tap_opt):
    torch.save({'iteration': iteration, 'cg_model': cg_model.state_dict(), 'tap_model': tap_model.state_dict(),
                'cg_optimizer': cg_opt.state_dict(), 'tap_optimizer': tap_opt.state_dict()}, path)


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--iters', type=int, default=20)
    ap.add_argument('--m_batch', type=int, default=1)
    ap.add_argument('--joint', action='store_true', help="'tap_cg' mode: gradients flow through tap_feats into the SST")
    ap.add_argument('--events', type=int, default=16)
    ap.add_argument('--segments', type=int, default=32)
    ap.add_argument('--vocab', type=int, default=500)
    ap.add_argument('--lr', type=float, default=5e-4)
    ap.add_argument('--save', type=str, default='')
    ap.add_argument('--resume', type=str, default='', help='checkpoint written by --save (or by the reference): models + optimiser state')
    ap.add_argument('--quiet', action='store_true')
    ap.add_argument('--no-fused', action='store_true', help="keep the autograd path also where one echr_train_step call per iteration applies "
                                                             "('pre_cg' mode with m_batch = 1)")
    a = ap.parse_args(argv)
    dev = torch.device('cuda')
    opt = synth.default_opt(vocab_size=a.vocab, seq_length=10, K=32, lr=a.lr)
    torch.manual_seed(0)
    tap_model = models.setup_tap(opt).to(dev)
    cg_model = echr_amd.CaptionGenerator(opt).to(dev)
    tap_model.train()
    cg_model.train()
    tap_opt = ClampAdam(tap_model.parameters(), lr=opt.lr, betas=(opt.optim_alpha, opt.optim_beta), eps=opt.optim_epsilon)
    cg_opt = ClampAdam(cg_model.parameters(), lr=opt.lr, betas=(opt.optim_alpha, opt.optim_beta), eps=opt.optim_epsilon,
                       arena=cg_model.build_arena())
    # the reference clamps the running gradient after EVERY backward (train.py:315-317); ClampAdam keeps that trajectory for any m_batch
    # (the clamp deferred to the fused step kernel is applied first whenever another backward accumulates)
    start = 0
    if a.resume:
        ck = torch.load(a.resume, map_location=dev)
        cg_model.load_state_dict(ck['cg_model'])
        tap_model.load_state_dict(ck['tap_model'])
        cg_opt.load_state_dict(ck['cg_optimizer'])
        tap_opt.load_state_dict(ck['tap_optimizer'])
        start = int(ck['iteration'])
    cg_crit, tap_crit = utils.LanguageModelCriterion(), utils.TAPModelCriterion()
    # 'pre_cg' mode (train_ECHR.sh) with m_batch = 1: the whole iteration around the caption model -- zero_grad, forward, criterion,
    # backward, clip_gradient, step (train.py:281-317) -- is ONE library call
    # 'tap_cg' mode with m_batch = 1: the caption side is the same call; d loss / d tap_feats comes back in tap_grad and goes into the proposal
    # encoder together with its own loss, cg_model's parameter gradients + Adam finish on the library's helper streams meanwhile
    fused = None
    if a.m_batch == 1 and not a.no_fused:
        from echr_amd.fused import FusedTrainStep
        fused = FusedTrainStep(cg_model, cg_opt, grad_clip=opt.grad_clip)
    loader = make_loader(opt, 8, a.events, a.segments, opt.CG_seq_length + 2)
    history = []
    for it in range(start, start + a.iters):
        v = loader[it % len(loader)]
        set_lr_for_epoch(cg_opt, opt.lr, it // len(loader))
        c3d, lda = torch.from_numpy(v['c3d']).to(dev), torch.from_numpy(v['lda']).to(dev)
        if fused is not None and a.joint:
            tap_opt.zero_grad()
            tap_feats, pred_proposals = tap_model(c3d)
            tap_loss = 0.01 * tap_crit(pred_proposals, torch.from_numpy(v['tap_masks']).to(dev), torch.from_numpy(v['tap_labels']).to(dev),
                                       torch.from_numpy(v['w1']).to(dev))                       # lambda1 (opts.py:194-196); lambda2 = 1
            g_tap = torch.zeros_like(tap_feats)
            cg_loss = fused(tap_feats.detach(), c3d, lda, v['labels'], v['ind'], v['soi'], torch.from_numpy(v['labels'])[:, 1:],
                            torch.from_numpy(v['masks'])[:, 1:], tap_grad=g_tap, defer_update=True)
            torch.autograd.backward([tap_loss, tap_feats], [None, g_tap])
            utils.clip_gradient(tap_opt, opt.grad_clip)
            tap_opt.step()
            history.append(float(cg_loss))
            if not a.quiet and (it % 5 == 0 or it == start + a.iters - 1):
                print('iter %3d  cg_loss %.4f' % (it, history[-1]), flush=True)
            continue
        if fused is not None:
            with torch.no_grad():
                tap_feats, _ = tap_model(c3d)
            cg_loss = fused(tap_feats, c3d, lda, v['labels'], v['ind'], v['soi'], torch.from_numpy(v['labels'])[:, 1:], torch.from_numpy(v['masks'])[:, 1:])
            history.append(float(cg_loss))
            if not a.quiet and (it % 5 == 0 or it == start + a.iters - 1):
                print('iter %3d  cg_loss %.4f' % (it, history[-1]), flush=True)
            continue
        if it % a.m_batch == 0:
            cg_opt.zero_grad()
            tap_opt.zero_grad()
        tap_feats, pred_proposals = tap_model(c3d)
        if not a.joint:
            tap_feats = tap_feats.detach()                                       # 'pre_cg' mode: the proposal net is idle
        pred = cg_model(tap_feats, c3d, lda, v['labels'], v['ind'], v['soi'], mode='train')
        cg_loss = cg_crit(pred, torch.from_numpy(v['labels'])[:, 1:].to(dev), torch.from_numpy(v['masks'])[:, 1:].to(dev))
        loss = cg_loss
        if a.joint:
            tap_loss = tap_crit(pred_proposals, torch.from_numpy(v['tap_masks']).to(dev), torch.from_numpy(v['tap_labels']).to(dev),
                                torch.from_numpy(v['w1']).to(dev))
            loss = 0.01 * tap_loss + 1.0 * cg_loss                                # lambda1, lambda2 defaults (opts.py:194-196)
        loss.backward()
        utils.clip_gradient(cg_opt, opt.grad_clip)                                # after every backward, as train.py:315,325-326
        if a.joint:
            utils.clip_gradient(tap_opt, opt.grad_clip)
        if (it + 1) % a.m_batch == 0:
            cg_opt.step()
            if a.joint:
                tap_opt.step()
        history.append(float(cg_loss.detach()))
        if not a.quiet and (it % 5 == 0 or it == start + a.iters - 1):
            print('iter %3d  cg_loss %.4f' % (it, history[-1]), flush=True)
    if fused is not None:
        fused.join()                                                              # a deferred update of the last iteration
    cg_model.eval()
    with torch.no_grad():
        v = loader[0]
        c3d, lda = torch.from_numpy(v['c3d']).to(dev), torch.from_numpy(v['lda']).to(dev)
        tap_model.eval()
        tap_feats, _ = tap_model(c3d)
        seq, logp = cg_model(tap_feats, c3d, lda, [], v['ind'], v['soi'], mode='eval')
    if not a.quiet:
        print('greedy captions (token ids) of video 0:', seq[:3].tolist() if len(seq) else seq)
    if a.save:
        save_checkpoint(a.save, start + a.iters, cg_model, tap_model, cg_opt, tap_opt)
    return history, cg_model, tap_model


if __name__ == '__main__':
    main()
