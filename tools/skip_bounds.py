#!/usr/bin/env python3
"""Upper bounds on what removing / hiding a class of work could gain on the c3 iteration: time the iteration with that class's launches
skipped (`diag_skip`: results are wrong while set -- diagnostic only), alternating with the full iteration on the same box."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench, echr_amd
from echr_amd import _lib
from echr_amd.misc.utils import LanguageModelCriterion, clip_gradient
from echr_amd.optim import ClampAdam

lib = _lib.load()
dev = torch.device('cuda', 0)
opt, params, vid = bench.make_workload(0, False)
model = echr_amd.CaptionGenerator(opt)
model.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()})
model = model.to(dev).train()
crit = LanguageModelCriterion()
optim = ClampAdam(model.parameters(), lr=opt.lr, arena=model.build_arena())
tap, c3d, lda = (torch.from_numpy(vid[k]).to(dev) for k in ('tap', 'c3d', 'lda'))
labels = torch.from_numpy(vid['labels'])
tgt, msk = labels[:, 1:].to(dev), torch.from_numpy(vid['masks'])[:, 1:].to(dev)


from echr_amd.fused import FusedTrainStep
fused = FusedTrainStep(model, optim, grad_clip=opt.grad_clip)
tgt_h, msk_h = labels[:, 1:].numpy(), vid['masks'][:, 1:]


def iteration():          # the product path: one library call per iteration
    fused(tap, c3d, lda, labels, vid['ind'], vid['soi'], tgt_h, msk_h)


def timed(n=300):
    for _ in range(10):
        iteration()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        iteration()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for name, bits in (('h2 operand packs (10 launches)', 1), ('clamp+Adam', 2), ('att_post', 4), ('embedding scatter-add', 8), ('all four', 15),
                   ('fp32-path products (12 launches)', 16), ('h2 products (10 launches)', 32), ('packs + h2 products', 33)):
    lib.echr_config_set(b'diag_skip', 0)
    a = timed()
    lib.echr_config_set(b'diag_skip', bits)
    b = timed()
    lib.echr_config_set(b'diag_skip', 0)
    print('%-34s full %.3f ms   skipped %.3f ms   bound %.0f us' % (name, a, b, (a - b) * 1e3), flush=True)
