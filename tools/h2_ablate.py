#!/usr/bin/env python3
"""What bounds the h2 product at the bench shapes: the full kernel against its two halves, stand-alone (diag_skip 64 = the LDS-DMA loads,
waits and barriers only -- no fragment reads, no MFMAs; 128 = fragment reads, MFMAs and folds only -- nothing is loaded).  If "loads only"
takes what the full kernel takes, the product is bound by the memory system's delivery rate into the CU, not by how the waves are organised
around it (loader / consumer specialisation can then gain at most the difference)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from echr_amd import _lib as L
from tools.h2_bench import pack, desc, timeit

lib = L.load()
dev = torch.device('cuda')
SH = [('logits_c', 762, 5001, 1536), ('dOUT_c', 762, 1536, 5001), ('g_w_logit_c', 5001, 1536, 762), ('gin_x3', 3840, 2048, 512),
      ('pall', 8192, 512, 500), ('logits', 1280, 5001, 1536), ('big', 4096, 4096, 4096)]
print('%-12s %5s %5s %5s | %9s %9s %9s | tiles  KB ingested/tile  GB/s per busy CU (loads only)' % ('shape', 'M', 'N', 'K', 'full us', 'loads us', 'math us'))
for name, M, N, K in SH:
    A, B = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev)
    Ax, Bx = pack(A), pack(B)
    Cc = torch.zeros(M, N, device=dev)
    d = desc(Ax, Bx, Cc, M, N, K, split=1)
    t = []
    for bits in (0, 64, 128):
        lib.echr_config_set(b'diag_skip', bits)
        t.append(timeit(lambda: lib.echr_gemm_f32(C.byref(d), L.stream_ptr())))
    lib.echr_config_set(b'diag_skip', 0)
    tiles = ((M + 127) // 128) * ((N + 127) // 128)
    kb = ((K + 31) // 32) * 33.0
    busy = min(tiles, 512) if tiles > 256 else tiles
    cus = min(tiles, 256)
    print('%-12s %5d %5d %5d | %9.1f %9.1f %9.1f | %5d  %8.0f  %8.1f' % (name, M, N, K, t[0], t[1], t[2], tiles, kb, tiles * kb * 1e3 / t[1] / cus / 1e3), flush=True)
