#!/usr/bin/env python3
"""Loss trajectories of the c3 bench workload on the autograd path and on the one-call path (echr_train_step), two runs each from the
same initial state: shows run-to-run spread (fp32 atomics + Adam) against any path-to-path difference."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench, echr_amd
from echr_amd.misc.utils import LanguageModelCriterion, clip_gradient
from echr_amd.optim import ClampAdam
from echr_amd.fused import FusedTrainStep
dev = torch.device('cuda', 0)
opt, params, vid = bench.make_workload(0, False)
tap, c3d, lda = (torch.from_numpy(vid[k]).to(dev) for k in ('tap', 'c3d', 'lda'))
labels = torch.from_numpy(vid['labels'])
tgt = labels[:, 1:].to(dev); msk = torch.from_numpy(vid['masks'])[:, 1:].to(dev)
crit = LanguageModelCriterion()
K = int(sys.argv[1]) if len(sys.argv) > 1 else 60
def run(fused_on):
    model = echr_amd.CaptionGenerator(opt)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()})
    model = model.to(dev).train()
    model.set_dropout_state(1234, 0)
    optim = ClampAdam(model.parameters(), lr=opt.lr, betas=(opt.optim_alpha, opt.optim_beta), eps=opt.optim_epsilon, arena=model.build_arena())
    fused = FusedTrainStep(model, optim, grad_clip=opt.grad_clip) if fused_on else None
    out = []
    for it in range(K):
        if fused is not None:
            loss = fused(tap, c3d, lda, labels, vid['ind'], vid['soi'], tgt, msk)
        else:
            optim.zero_grad()
            loss = crit(model(tap, c3d, lda, labels, vid['ind'], vid['soi'], mode='train'), tgt, msk)
            loss.backward(); clip_gradient(optim, opt.grad_clip); optim.step()
        out.append(loss.detach().clone())
    torch.cuda.synchronize()
    return [float(x) for x in out]
runs = {'autograd A': run(False), 'autograd B': run(False), 'fused A': run(True), 'fused B': run(True)}
for it in sorted(set([0, 1, 2, 3, 5, 9, 19, 39, K - 1])):
    if it < K:
        print('it %3d  ' % it + '  '.join('%s %.6f' % (k, v[it]) for k, v in runs.items()))
