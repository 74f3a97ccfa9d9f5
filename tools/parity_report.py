"""Actual parity errors of the HIP path against the reference-generated fixtures (tests/golden), per GEMM path (GPU box only)."""
import sys, os; sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo')); sys.path.insert(0, os.path.join(os.environ.get('GRAFT_REPO_ROOT', '/root/repo'), 'tests'))
import numpy as np, torch
import util as U
from echr_amd import synth, _lib
from oracle import summary as SM
lib = _lib.load()
for case in ('c2full', 'c1'):
    opt, params, vid = synth.make_case(case)
    g = U.gold('case_%s.npz' % case)
    for h2 in (1, 0):
        lib.echr_config_set(b'gemm_h2', h2)
        pred, loss, grads, _ = U.run_gpu(opt, params, vid, False)
        s = SM.summarize_logp(pred)
        gs = SM.summarize_grads(grads)
        worst = 0.0; wk = ''
        for key, v in gs.items():
            name = key.split('|')[0]
            if name in U.NOISE_ONLY or key.endswith('|l2') or key.endswith('|linf'): continue
            scale = max(float(g['eval|grad|' + name + '|linf']), 1e-30)
            e = float(np.abs(v - g['eval|grad|' + key]).max() / scale)
            if e > worst: worst, wk = e, key
        print(case, 'h2' if h2 else 'bf16x3', 'max|dlogp| %.2e' % np.abs(s['slice'] - g['eval|logp|slice']).max(), 'dloss rel %.2e' % abs(loss / float(g['eval|loss']) - 1), 'worst grad slice rel-to-linf %.2e (%s)' % (worst, wk))
lib.echr_config_set(b'gemm_h2', 1)
