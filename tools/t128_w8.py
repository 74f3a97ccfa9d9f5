#!/usr/bin/env python3
"""The eight-wave form of the exact-fp32 128 x 128 tile (gemm_f32_t128_kernel<.., W8>, launches of at most one workgroup per CU) against the
four-wave form (ECHR_T128_W8=0) and the 64 x 64 tile: value check on NT shapes, then timings; split-K sweep for the 72-tile d OUTD product."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools import t128_bench as T
from tools.gemm_bench import run

lib = T.lib
lib.echr_config_set(b'gemm_tile', ord('t'))
for (M, N, K, pad) in [(128, 128, 64, 0), (762, 1536, 5001, 0), (762, 1536, 5001, 3), (130, 257, 100, 0), (762, 5001, 1536, 0), (513, 502, 1000, 1), (300, 131, 67, 2)]:
    for split in (-1, 1, 3):
        e = T.check('NT', M, N, K, pad, split)
        print('check NT %5d %5d %5d pad %d split %2d rel err %.2e %s' % (M, N, K, pad, split, e, 'ok' if e < 1e-5 else 'BAD'), flush=True)
for sh in [('logits', 'NT', 762, 5001, 1536), ('dOUTD', 'NT', 762, 1536, 5004), ('fc1', 'NT', 4096, 512, 512), ('sq2048', 'NT', 2048, 2048, 2048),
           ('dWlogit', 'NT', 5001, 1536, 764), ('pall', 'NT', 8192, 512, 500), ('4096^3', 'NT', 4096, 4096, 4096)]:
    out = []
    for code in (ord('6'), ord('t')):
        lib.echr_config_set(b'gemm_tile', code)
        us, tf = run(*sh)
        out.append('%7.1fus %5.1fTF' % (us, tf))
    sp = []
    for s_ in (2, 3, 4):
        lib.echr_config_set(b'gemm_split', s_)
        us, tf = run(*sh)
        sp.append('%.0f' % us)
    lib.echr_config_set(b'gemm_split', 0)
    lib.echr_config_set(b'gemm_tile', 0)
    print('%-8s %5d %5d %5d | 64x64 %s | t128 %s | t128 split 2/3/4 %s' % (sh[0], sh[2], sh[3], sh[4], out[0], out[1], '/'.join(sp)), flush=True)
