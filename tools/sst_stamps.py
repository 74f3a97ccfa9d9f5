#!/usr/bin/env python3
"""Diagnostic: phase timeline of the persistent SST recurrence kernels (csrc/sst.hip), workgroup 0, s_memrealtime (100 MHz).
usage: python tools/sst_stamps.py [bwd]"""
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from echr_amd import _lib, models, synth

lib = _lib.load()
BWD = 'bwd' in sys.argv
opt = synth.default_opt()
m = models.setup_tap(opt).cuda()
m.train()
T = 256
x = torch.randn(T, 500, device='cuda')
for it in range(4):
    if it == 3:
        lib.echr_config_set(b'persist_stamps', 4 if BWD else 3)
    tap, sc = m(x)
    if BWD:
        for p in m.parameters():
            p.grad = None
        (tap.sum() + sc.sum()).backward()
    torch.cuda.synchronize()
buf = np.zeros(4 * 256 * 16, dtype=np.uint64)
S = lib.echr_persist_read_stamps(buf.ctypes.data, buf.size)
assert S == 256, S
st = buf[:S * 16].reshape(S, 16).astype(np.float64) / 100.0
a = st[4:250]
print('%s: step period %.2f us' % ('reverse' if BWD else 'forward', (st[249, 0] - st[4, 0]) / 245))
names = ['prefetch->poll done', 'barrier', 'GEMV + reductions', 'cell math + publish']
for i, n in enumerate(names):
    print('   %-22s %.2f us' % (n, (a[:, i + 1] - a[:, i]).mean()))
print('   %-22s %.2f us' % ('publish -> next step', (st[5:251, 0] - st[4:250, 4]).mean()))
