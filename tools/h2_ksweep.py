#!/usr/bin/env python3
"""h2 GEMM duration against K at fixed M x N (GPU box): separates the per-launch fixed cost from the per-k-block cost."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from h2_bench import pack, desc, timeit, lib, dev, L

for M, N in ((4096, 512), (1280, 2048), (1280, 5001)):
    for K in (64, 128, 256, 512, 1024, 2048, 4096):
        A, B = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev)
        Ax, Bx = pack(A), pack(B)
        Cc = torch.zeros(M, N, device=dev)
        d = desc(Ax, Bx, Cc, M, N, K, split=1)
        us = timeit(lambda: lib.echr_gemm_f32(C.byref(d), L.stream_ptr()), reps=50)
        print('M=%d N=%d K=%5d  %7.1f us  (%.2f us per k block)' % (M, N, K, us, us / (K / 32)), flush=True)
