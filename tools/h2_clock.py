#!/usr/bin/env python3
"""30 launches of the h2 product on the logits shape in one ablation mode (argv[1]: 0 = whole kernel, 64 = loads only, 128 = compute only), for
a counter run: GRBM_GUI_ACTIVE / duration = the clock the chip held during the launch (tools/r6_h2_clock.sh)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from echr_amd import _lib as L
from tools.h2_bench import pack, desc

lib = L.load()
dev = torch.device('cuda')
M, N, K = (int(x) for x in (sys.argv[2:5] if len(sys.argv) > 4 else (762, 5001, 1536)))
A, B = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev)
Ax, Bx = pack(A), pack(B)
Cc = torch.zeros(M, N, device=dev)
d = desc(Ax, Bx, Cc, M, N, K, split=1)
lib.echr_config_set(b'diag_skip', int(sys.argv[1]))
for _ in range(30):
    lib.echr_gemm_f32(C.byref(d), L.stream_ptr())
torch.cuda.synchronize()
