#!/usr/bin/env python3
"""Time the native SST proposal encoder (fwd, fwd+bwd) on a T-segment video (GPU box only)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from echr_amd import models, synth

opt = synth.default_opt()
m = models.setup_tap(opt).cuda()
m.train()
TS = [int(a) for a in sys.argv[1:] if a.isdigit()] or [128, 256, 512]
for T in TS:
    x = torch.randn(T, 500, device='cuda')
    for mode in ('fwd', 'fwd+bwd'):
        for rep in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                tap, sc = m(x)
                if mode != 'fwd':
                    for p in m.parameters():
                        p.grad = None
                    (tap.sum() + sc.sum()).backward()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 5
        print('T=%4d %-8s %.3f ms  (%.1f us per LSTM step-launch)' % (T, mode, dt * 1e3, dt * 1e6 / (2 * T * (1 if mode == "fwd" else 2))), flush=True)
if 'nomiopen' in sys.argv:
    sys.exit(0)
# stock MIOpen nn.LSTM for comparison (same module used as a plain torch module)
import torch.nn as nn
ref = nn.LSTM(500, 512, 2, batch_first=True).cuda()
for T in TS:
    x = torch.randn(1, T, 500, device='cuda')
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            out, _ = ref(x)
            for p in ref.parameters():
                p.grad = None
            out.sum().backward()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 5
    print('T=%4d stock nn.LSTM (MIOpen) fwd+bwd %.3f ms' % (T, dt * 1e3), flush=True)
