#!/usr/bin/env python3
"""A/B of the exact-fp32 GEMM on the large NT shapes: gemm_f32_nt128_kernel (auto) against the 64x64 tile (tile override '6')."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.gemm_bench import run
from echr_amd import _lib
lib = _lib.load()
for sh in [('logits', 'NT', 1280, 5001, 1536), ('dWlogitNT', 'NT', 5001, 1536, 1280), ('dOUTD_NT', 'NT', 1280, 1536, 5004), ('gin', 'NT', 1280, 2048, 512),
           ('pall', 'NT', 8192, 512, 500), ('4096^3', 'NT', 4096, 4096, 4096)]:
    lib.echr_config_set(b'gemm_tile', 0)
    a = run(*sh)
    lib.echr_config_set(b'gemm_tile', ord('6'))
    b = run(*sh)
    lib.echr_config_set(b'gemm_tile', 0)
    print('%-10s %5d %5d %5d | nt128 %7.1f us %6.1f TF | 64x64 %7.1f us %6.1f TF' % (sh[0], sh[2], sh[3], sh[4], a[0], a[1], b[0], b[1]), flush=True)
