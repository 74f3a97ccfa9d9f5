#!/usr/bin/env python3
"""Diagnostic: phase timeline of the persistent greedy decoder (csrc/persist.hip, SAMP instantiations) at the benchmark size (N = 64,
A = 128, V1 = 5001).  Average time between consecutive stamps over the steady-state steps (s_memrealtime, 100 MHz)."""
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import echr_amd
from echr_amd import _lib, synth

lib = _lib.load()
dev = torch.device('cuda')
opt = synth.default_opt(vocab_size=5000, seq_length=19)
params = synth.make_params(opt, 0)
m = echr_amd.CaptionGenerator(opt)
m.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()})
m = m.to(dev).eval()
vid = synth.make_video(64, 128, 21, 5001, seed=7, T_v=8192, full_len=True, disjoint=True)
tap, c3d, lda = (torch.from_numpy(vid[k]).to(dev) for k in ('tap', 'c3d', 'lda'))
with torch.no_grad():
    for it in range(3):
        if it == 2:
            lib.echr_config_set(b'persist_stamps', 1)
        seq, lp = m(tap, c3d, lda, [], vid['ind'], vid['soi'], mode='eval')
        torch.cuda.synchronize()
S = 19
buf = np.zeros(4 * 256 * 16, dtype=np.uint64)
got = lib.echr_persist_read_stamps(buf.ctypes.data, buf.size)
assert got == S, got
st = buf[:4 * S * 16].reshape(4, S, 16).astype(np.float64) / 100.0     # us
names = {0: ['step', '-', '-', '-', '-', 'waitQ', 'att', 'pubC', 'waitC', 'mfmaC', 'gate(+waitTok)', 'pubH1'],
         1: ['step', 'waitH1', 'mfmaA', 'qepi', 'pubQ', 'waitQ', 'att', 'pubC'],
         2: ['step', '-', '-', '-', '-', 'waitQ', 'att', 'pubC'],
         3: ['step', 'waitTok', 'cell+pubH', 'waitH0H2', 'phaseA', 'waitH1', 'phaseB', 'fold+pubTok', 'rec']}
raw = st[0, 0]
print('workgroup 0: set-up %.2f us, first -> last step stamp %.2f us' % (raw[14] - raw[15], st[0, S - 1, 0] - st[0, 0, 0]))
for role, rn in ((0, 'gate wg'), (1, 'q wg'), (2, 'att-only wg'), (3, 'lstm + logits wg 0')):
    a = st[role]
    print('%s: step period %.2f us' % (rn, (a[S - 1, 0] - a[2, 0]) / (S - 3)))
    prev_i, out = 0, []
    for i in range(1, len(names[role])):
        if names[role][i] == '-':
            continue
        out.append('%s %.2f' % (names[role][i], (a[3:S - 1, i] - a[3:S - 1, prev_i]).mean()))
        prev_i = i
    print('   ' + ' | '.join(out))
# absolute offsets within a step, relative to the gate workgroup's step stamp
g0 = st[0, 3:S - 1, 0]
for role, rn in ((0, 'gate'), (3, 'lstm')):
    print(rn, 'offsets vs gate step start:', ' '.join('%d:%.1f' % (i, (st[role, 3:S - 1, i] - g0).mean()) for i in range(13) if st[role, 3, i] > 0))
