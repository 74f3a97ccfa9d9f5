#!/usr/bin/env python3
"""Greedy-decode (mode='eval') throughput of the caption path at benchmark and evaluation sizes (GPU box only)."""
import gc
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import echr_amd
from echr_amd import synth

dev = torch.device('cuda')
opt = synth.default_opt(vocab_size=5000, seq_length=19)
params = synth.make_params(opt, 0)
m = echr_amd.CaptionGenerator(opt)
m.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()})
m = m.to(dev).eval()
SIZES = [(n, 8192 if n <= 64 else 256) for n in (int(a) for a in sys.argv[1:] if a.isdigit())] or [(64, 8192), (1000, 256)]
for N, T_v in SIZES:
    vid = synth.make_video(N, 128, 21, 5001, seed=7, T_v=T_v if N > 64 else None, full_len=(N == 64), disjoint=(N == 64))
    tap, c3d, lda = (torch.from_numpy(vid[k]).to(dev) for k in ('tap', 'c3d', 'lda'))
    with torch.no_grad():
        for _ in range(2):
            seq, lp = m(tap, c3d, lda, [], vid['ind'], vid['soi'], mode='eval')
        torch.cuda.synchronize()
        gc.collect()
        gc.disable()          # (a generation-2 collection of the interpreter -- ~40 ms with torch loaded -- otherwise lands in one of the five timed decodes)
        try:
            t0 = time.perf_counter()
            reps = 5
            for _ in range(reps):
                seq, lp = m(tap, c3d, lda, [], vid['ind'], vid['soi'], mode='eval')
            torch.cuda.synchronize()
        finally:
            gc.enable()
    dt = (time.perf_counter() - t0) / reps
    steps = seq.shape[1] if len(seq) else 0
    print('N=%4d events: %.2f ms per decode, %d generated steps (+1 BOS step), %.0f event-timesteps/s' % (N, dt * 1e3, steps, N * (steps + 1) / dt), flush=True)
