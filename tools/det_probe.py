#!/usr/bin/env python3
"""Which gradients of one training iteration differ run to run (bitwise), under which switches: the order-dependent (atomic) sums of the path.
usage: python tools/det_probe.py  -- prints, per configuration, the tensors whose two runs differ and by how much"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from echr_amd import _lib, synth
from tests import util as U
from tests.test_gpu_timed_path import _fused, _device_inputs

lib = _lib.load()
opt, params, vid = synth.make_case('c2')
tap, c3d, lda, labels, tgt_h, msk_h = _device_inputs(vid)


def run():
    torch.manual_seed(1234)
    m, o, f = _fused(opt, params, True)
    loss = float(f(tap, c3d, lda, labels, vid['ind'], vid['soi'], tgt_h, msk_h, step=False))
    torch.cuda.synchronize()
    return loss, {k: p.grad.detach().cpu().numpy().copy() for k, p in m.named_parameters() if p.grad is not None}


RESET = {b'persist': 1, b'persist_bwd': 1, b'gemm_split': 0, b'deterministic': 0}
for name, sets in (('default', {}), ('persist off', {b'persist': 0, b'persist_bwd': 0}), ('persist off + split 1', {b'persist': 0, b'persist_bwd': 0, b'gemm_split': 1}),
                   ('deterministic', {b'deterministic': 1})):
    for k, v in sets.items():
        lib.echr_config_set(k, v)
    a, b = run(), run()
    diff = {k: float(np.abs(a[1][k] - b[1][k]).max() / max(np.abs(a[1][k]).max(), 1e-30)) for k in a[1] if not np.array_equal(a[1][k], b[1][k])}
    print('== %s: loss equal %s; %d of %d gradient tensors differ' % (name, a[0] == b[0], len(diff), len(a[1])))
    for k, v in sorted(diff.items(), key=lambda kv: -kv[1]):
        print('   %-50s %.2e' % (k, v))
    for k in sets:
        lib.echr_config_set(k, RESET[k])
