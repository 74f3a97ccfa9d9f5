#!/usr/bin/env python3
"""Upper bound on what ANY fusion of the event encoder (TSRM) could gain on the c3 iteration: time the iteration with the encoder's ~25 launches
replaced by ONE trivial kernel (a slice of its input: numerically wrong, diagnostic only), alternating with the real encoder on the same box."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench, echr_amd
from echr_amd import functional as EF
from echr_amd.misc.utils import LanguageModelCriterion, clip_gradient
from echr_amd.optim import ClampAdam

dev = torch.device('cuda', 0)
opt, params, vid = bench.make_workload(0, False)
model = echr_amd.CaptionGenerator(opt)
model.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()})
model = model.to(dev).train()
crit = LanguageModelCriterion()
arena = model.build_arena()
optim = ClampAdam(model.parameters(), lr=opt.lr, arena=arena)
tap, c3d, lda = (torch.from_numpy(vid[k]).to(dev) for k in ('tap', 'c3d', 'lda'))
labels = torch.from_numpy(vid['labels'])
tgt, msk = labels[:, 1:].to(dev), torch.from_numpy(vid['masks'])[:, 1:].to(dev)
real_apply = EF.TSRMFunction.apply


class Stub(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ech, *rest):
        ctx.shape = ech.shape
        return ech[:, :512].contiguous()

    @staticmethod
    def backward(ctx, g):
        ge = g.new_zeros(ctx.shape)
        ge[:, :512] = g
        return (ge,) + (None,) * 18


def iteration():
    optim.zero_grad()
    loss = crit(model(tap, c3d, lda, labels, vid['ind'], vid['soi'], mode='train'), tgt, msk)
    loss.backward()
    clip_gradient(optim, opt.grad_clip)
    optim.step()


def timed(n=300):
    for _ in range(10):
        iteration()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        iteration()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for rep in range(3):
    EF.TSRMFunction.apply = real_apply
    a = timed()
    EF.TSRMFunction.apply = Stub.apply
    b = timed()
    print('real event encoder %.3f ms / iteration   encoder stubbed out %.3f ms   (bound on any fusion gain: %.0f us)' % (a, b, (a - b) * 1e3), flush=True)
