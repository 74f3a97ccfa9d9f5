#!/usr/bin/env python3
"""How many extra HIP streams does the training iteration tolerate?  (GPU box.)  Runs the c3 iteration with k extra torch streams
that each get one tiny kernel per iteration ordered after the main stream (the shape of a collective's stream in a data-parallel run),
and prints ms / iteration.  The library itself uses the caller's stream + 2 helper streams."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import echr_amd
from echr_amd.misc.utils import LanguageModelCriterion, clip_gradient
from echr_amd.optim import ClampAdam

opt, params, vid = bench.make_workload(0, False)
dev = torch.device('cuda')
model = echr_amd.CaptionGenerator(opt)
model.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()})
model = model.to(dev).train()
model.build_arena()
optim = ClampAdam(model.parameters(), lr=5e-5)
tap, c3d, lda = (torch.from_numpy(vid[k]).to(dev) for k in ('tap', 'c3d', 'lda'))
labels = torch.from_numpy(vid['labels'])
tgt, msk = labels[:, 1:].to(dev), torch.from_numpy(vid['masks'])[:, 1:].to(dev)
crit = LanguageModelCriterion()
scratch = torch.zeros(1024, device=dev)


def iteration(extra, mode):
    optim.zero_grad()
    loss = crit(model(tap, c3d, lda, labels, vid['ind'], vid['soi'], mode='train'), tgt, msk)
    loss.backward()
    for s in extra:
        if mode == 'dep':
            s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            scratch.add_(1.0)
        if mode == 'dep':
            torch.cuda.current_stream().wait_stream(s)
    clip_gradient(optim, opt.grad_clip)
    optim.step()


for k, mode in ((0, 'none'), (1, 'idle'), (1, 'free'), (1, 'dep'), (2, 'dep'), (3, 'dep'), (4, 'dep'), (0, 'none')):
    extra = [torch.cuda.Stream() for _ in range(k)]
    for _ in range(5):
        iteration(extra if mode != 'idle' else [], mode)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(30):
        iteration(extra if mode != 'idle' else [], mode)
    torch.cuda.synchronize()
    print('extra streams %d (%s): %.3f ms / iteration' % (k, mode, (time.perf_counter() - t0) / 30 * 1e3), flush=True)
