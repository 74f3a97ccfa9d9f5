#!/usr/bin/env python3
"""Time the fp32 MFMA GEMM on the shapes one training iteration of the c3 workload issues (GPU box only)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from echr_amd import _lib as L

SHAPES = [  ('wgradNT', 'NT', 2048, 512, 1280), ('dXT_NT', 'NT', 1280, 512, 2048),
  # (name, layout, M, N, K)
    ('logits', 'NT', 1280, 5001, 1536), ('dOUT', 'NN', 1280, 1536, 5001), ('g_w_logit', 'TN', 5001, 1536, 1280),
    ('gin', 'NT', 1280, 2048, 512), ('dXT', 'NN', 1280, 512, 2048), ('g_w_hh', 'TN', 2048, 512, 1280),
    ('g_w_att', 'TN', 2048, 500, 1280), ('g_w_h2a', 'TN', 512, 512, 1280), ('g_w_c2a', 'TN', 512, 500, 8192),
    ('pall', 'NT', 8192, 512, 500), ('fc1', 'NT', 4096, 512, 512), ('g_w_fc1', 'TN', 512, 512, 4096),
]


def run(name, layout, M, N, K, reps=20, algo=0):
    lib = L.load()
    dev = torch.device('cuda')
    if layout == 'NT':
        A, B = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev)
        st = (K, 1, 1, K)
    elif layout == 'NN':
        A, B = torch.randn(M, K, device=dev), torch.randn(K, N, device=dev)
        st = (K, 1, N, 1)
    else:
        A, B = torch.randn(K, M, device=dev), torch.randn(K, N, device=dev)
        st = (1, M, N, 1)
    Cc = torch.zeros(M, N, device=dev)
    d = L.GemmDesc()
    d.A, d.B, d.C = A.data_ptr(), B.data_ptr(), Cc.data_ptr()
    d.M, d.N, d.K = M, N, K
    d.sam, d.sak, d.sbk, d.sbn = st
    d.ldc, d.batch, d.alpha, d.beta, d.split_k, d.algo = N, 1, 1.0, 0.0, -1, algo
    s = L.stream_ptr()
    for _ in range(3):
        L.check(lib.echr_gemm_f32(C.byref(d), s))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        lib.echr_gemm_f32(C.byref(d), s)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    return us, 2.0 * M * N * K / us / 1e6


if __name__ == '__main__':
    from echr_amd import _lib
    lib = _lib.load()

    def knobs(tile='', split=0):          # tuning overrides go through echr_config_set (the environment is only read once, at load)
        lib.echr_config_set(b'gemm_tile', ord(tile[0]) if tile else 0)
        lib.echr_config_set(b'gemm_split', int(split))

    cfgs = [('auto', {}), ('t64s1', dict(tile='6', split=1)), ('t64s2', dict(tile='6', split=2))]
    print('%-10s %-3s %5s %5s %5s | ' % ('name', 'lay', 'M', 'N', 'K') + ' '.join('%14s' % c[0] for c in cfgs))
    for sh in SHAPES:
        cells = []
        for cname, env in cfgs:
            knobs(**env)
            us, tf = run(*sh)
            cells.append('%6.0fus %4.0fTF' % (us, tf))
        knobs()
        if sh[1] == 'NT':
            us, tf = run(*sh, algo=1)
            cells.append('bf16x3 %6.0fus %4.0fTF' % (us, tf))
            for sp in (1, 2, 4):
                knobs('s', sp)
                us, tf = run(*sh, algo=1)
                cells.append('x3/64 s%d %5.0fus %4.0fTF' % (sp, us, tf))
            knobs()
        print('%-10s %-3s %5d %5d %5d | ' % sh + ' '.join('%14s' % c for c in cells), flush=True)
