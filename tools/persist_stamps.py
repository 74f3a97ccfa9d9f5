#!/usr/bin/env python3
"""Diagnostic: phase timeline of the persistent recurrence kernels (csrc/persist.hip) on the c3 bench workload.
Prints, per role (gate workgroup 0, q workgroup 128, attention-only workgroup 160, LSTM workgroup 0), the average
time between consecutive stamps over the steady-state timesteps (s_memrealtime, 100 MHz)."""
import ctypes as C
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import echr_amd
from echr_amd import _lib
from echr_amd.misc.utils import LanguageModelCriterion

lib = _lib.load()
C5 = 'c5' in sys.argv          # BASELINE config 5's proposals (4..256 segments: the BIG instantiations); tap_feats are synthetic here
opt, params, vid = bench.make_workload(0, False, C5)
dev = torch.device('cuda')
model = echr_amd.CaptionGenerator(opt)
model.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()})
model = model.to(dev).train()
tap, c3d, lda = (torch.from_numpy(vid[k]).to(dev) for k in ('tap', 'c3d', 'lda'))
labels = torch.from_numpy(vid['labels'])
BWD = 'bwd' in sys.argv
crit = LanguageModelCriterion()
tgt, msk = labels[:, 1:].to(dev), torch.from_numpy(vid['masks'])[:, 1:].to(dev)
for it in range(4):
    if it == 3:
        lib.echr_config_set(b'persist_stamps', 2 if BWD else 1)
    if BWD:
        model.zero_grad()
        crit(model(tap, c3d, lda, labels, vid['ind'], vid['soi'], mode='train'), tgt, msk).backward()
    else:
        with torch.no_grad():
            model(tap, c3d, lda, labels, vid['ind'], vid['soi'], mode='train')
    torch.cuda.synchronize()
S = 20
buf = np.zeros(4 * 256 * 16, dtype=np.uint64)
got = lib.echr_persist_read_stamps(buf.ctypes.data, buf.size)
assert got == S, got
st = buf[:4 * S * 16].reshape(4, S, 16).astype(np.float64) / 100.0     # us
names = {0: ['step', 'waitH1', 'mfmaA', '-', '-', 'waitQ', 'att', 'pubC', 'waitC', 'mfmaC', 'gate', 'pubH1'],
         1: ['step', 'waitH1', 'mfmaA', 'qepi', 'pubQ', 'waitQ', 'att', 'pubC'],
         2: ['step', '-', '-', '-', '-', 'waitQ', 'att', 'pubC', '-', '-', '-', '-', 'qload', 'score', 'ctx+lds'],
         3: ['step', 'waitH', 'mfma', 'gate', 'pub']}
if BWD:
    names = {0: ['step', 'waitDQ', 'mfmaQ', 'waitHH', 'gate', 'pubDG', '-', '-', '-', '-', 'waitDA', 'att', 'pubDQ'],
             1: ['step', '-', '-', '-', '-', '-', 'waitDG', 'mfmaA', 'pubDA', '-', 'waitDA', 'att', 'pubDQ'],
             2: ['step', '-', '-', '-', '-', '-', '-', '-', '-', '-', 'waitDA', 'att', 'pubDQ'],
             3: ['step', 'waitG', 'mfma', 'gate', 'pub']}
    rl = ((0, 'GD wg 0'), (1, 'P wg 32'), (2, 'att wg 160'), (3, 'lstm-bwd wg 0'))
else:
    rl = ((0, 'gate wg'), (1, 'q wg'), (2, 'att-only wg'), (3, 'lstm wg 0'))
raw = st[0, 0]
first = st[0, S - 1, 0] if BWD else st[0, 0, 0]
last = st[0, 0, 0] if BWD else st[0, S - 1, 0]
print('workgroup 0: set-up %.2f us (entry -> ready), ready -> first step stamp %.2f us, first -> last step stamp %.2f us' % (raw[14] - raw[15], first - raw[14], last - first))
if not BWD:
    sub = [st[0, i >> 1, 12 + (i & 1)] for i in range(5)]
    print('   set-up split: alpha %.2f | operand loads issued %.2f | W_hh1 image %.2f | W_att image %.2f | e^2p + context bound %.2f | barrier %.2f' % (
        sub[0] - raw[15], sub[1] - sub[0], sub[2] - sub[1], sub[3] - sub[2], sub[4] - sub[3], raw[14] - sub[4]))
for role, rn in rl:
    a = st[role]
    if BWD:
        a = a[::-1]          # the reverse recurrence walks t downwards
    print('%s: step period %.2f us' % (rn, (a[S - 1, 0] - a[2, 0]) / (S - 3)))
    prev_i = 0
    out = []
    for i in range(1, len(names[role])):
        if names[role][i] == '-':
            continue
        if i == 12 and not BWD:
            prev_i = 5
        if (a[3:S - 1, i] == 0).any() or (a[3:S - 1, prev_i] == 0).any():          # a stamp this instantiation does not write
            if not (a[3:S - 1, i] == 0).any():
                prev_i = i
            continue
        d = (a[3:S - 1, i] - a[3:S - 1, prev_i]).mean()
        out.append('%s %.2f' % (names[role][i], d))
        prev_i = i
    if BWD and role == 1:          # the d h1 tile runs behind the attention role (stamp 9 follows stamp 12)
        out.append('tileB+pubHH %.2f' % (a[3:S - 1, 9] - a[3:S - 1, 12]).mean())
    print('   ' + ' | '.join(out))
