#!/usr/bin/env python3
"""Where the caller's stream is, on the device clock, at the boundaries of one config-5 iteration (fused.JointTrainStep) -- UNPROFILED: an event
is recorded on the caller's stream around each library call of the iteration, and the medians of the differences are printed.  (rocprofv3
slows the host enough to change which side bounds this iteration, so its timeline cannot answer this.)
usage: python tools/c5_chain.py [steps]"""
import os
import sys
import types

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
args = types.SimpleNamespace(overlap=1, no_arena=False, fused='auto')
dev = torch.device('cuda:0')
torch.cuda.set_device(dev)
wl = bench.Workload(args, 'c5', 0, dev, False)
lib = wl.fused.lib
marks = []          # (label, event) of the current iteration


def wrap(name, label_after):
    fn = getattr(lib, name)

    def w(*a):
        rc = fn(*a)
        if name == 'echr_train_step' and os.environ.get('C5_SERIAL') == '1':
            # comparison: the caption side's deferred tail + update joined into the caller's stream BEFORE the proposal encoder's backward is
            # queued -- what that backward costs when it has the chip to itself
            lib.echr_stream_join(bench.L.stream_ptr() if hasattr(bench, 'L') else __import__('echr_amd')._lib.stream_ptr())
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        marks.append((label_after, e))
        return rc
    return fn, w


class Lib(object):          # a proxy: ctypes function objects cannot be patched in place
    def __init__(self, lib, over):
        self._lib, self._over = lib, over

    def __getattr__(self, k):
        return self._over.get(k) or getattr(self._lib, k)


over = {}
for name, label in (('echr_sst_fwd_states', 'sst forward (tap_feats)'), ('echr_train_step', 'caption call: d tap_feats + loss on the stream'),
                    ('echr_sst_head_fwd', 'proposal head forward'), ('echr_tap_bce_fwd_ws', 'BCE forward'), ('echr_tap_bce_bwd', 'BCE backward'),
                    ('echr_sst_bwd', 'sst backward (head grads + reverse recurrence + parameter gradients)'),
                    ('echr_train_step_prepare', 'prepare (caller-stream part)')):
    over[name] = wrap(name, label)[1]
import echr_amd.fused as F
proxy = Lib(lib, over)
j = None
# the objects hold `lib` references: swap them
wl.fused.lib = proxy
rows = []
it = wl.iteration
# find the JointTrainStep through the closure
def find(fn, depth=0):
    for c in getattr(fn, '__closure__', None) or []:
        v = c.cell_contents
        if isinstance(v, F.JointTrainStep):
            return v
        if callable(v) and depth < 3 and hasattr(v, '__closure__'):
            r = find(v, depth + 1)
            if r is not None:
                return r
    return None


j = find(it)
assert j is not None, 'bench built no JointTrainStep'
j.lib = proxy
own = torch.cuda.Stream() if os.environ.get('ECHR_CU_PARTITION') == '1' else torch.cuda.current_stream()
torch.cuda.synchronize()
torch.cuda.set_stream(own)          # (the CU-partition mode needs a caller that is not on the null stream)
for s in range(steps + 5):
    marks.clear()
    e0 = torch.cuda.Event(enable_timing=True)
    e0.record()
    it()
    e1 = torch.cuda.Event(enable_timing=True)
    e1.record()
    wl.fused.join()
    e2 = torch.cuda.Event(enable_timing=True)
    e2.record()
    torch.cuda.synchronize()
    if s >= 5:
        t = [('start', 0.0)] + [(l, e0.elapsed_time(e)) for l, e in marks] + [('tap update + loss sum (end of the caller-stream chain)', e0.elapsed_time(e1)),
                                                                              ('caption tail + update joined', e0.elapsed_time(e2))]
        rows.append(t)
labels = [l for l, _ in rows[0]]
med = np.median(np.array([[x for _, x in r] for r in rows]), axis=0)
prev = 0.0
print('config 5, %d iterations, medians [us] on the caller\'s stream' % steps)
for l, m in zip(labels, med):
    print('  %9.1f  (+%7.1f)  %s' % (m * 1e3, (m - prev) * 1e3, l))
    prev = m
