#!/usr/bin/env python3
"""Exact-fp32 products of ONE native_f32 iteration (shapes from ECHR_GEMM_LOG=1 on the c3 workload): the 128 x 128 tile for every layout
(gemm_f32_t128_kernel, tile override 't') against the 64 x 64 tile ('6') and the library's own choice ('auto'), stand-alone, back to back;
first a value check of the new kernel against torch (fp64 reference) on ragged / unaligned shapes."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from echr_amd import _lib as L
from tools.gemm_bench import run

lib = L.load()
lib.echr_config_set(b'gemm_h2', 0)
lib.echr_config_set(b'gemm_bf16x3', 0)


def check(layout, M, N, K, pad=0, split=-1):
    dev = torch.device('cuda')
    g = torch.Generator(device='cpu').manual_seed(M * 7 + N * 3 + K)
    if layout == 'NT':
        A = torch.randn(M, K + pad, generator=g).to(dev)[:, :K]; B = torch.randn(N, K + pad, generator=g).to(dev)[:, :K]
        st = (K + pad, 1, 1, K + pad); ref = A.double() @ B.double().t()
    elif layout == 'NN':
        A = torch.randn(M, K + pad, generator=g).to(dev)[:, :K]; B = torch.randn(K, N + pad, generator=g).to(dev)[:, :N]
        st = (K + pad, 1, N + pad, 1); ref = A.double() @ B.double()
    else:
        A = torch.randn(K, M + pad, generator=g).to(dev)[:, :M]; B = torch.randn(K, N + pad, generator=g).to(dev)[:, :N]
        st = (1, M + pad, N + pad, 1); ref = A.double().t() @ B.double()
    Cc = torch.zeros(M, N, device=dev)
    d = L.GemmDesc()
    d.A, d.B, d.C = A.data_ptr(), B.data_ptr(), Cc.data_ptr()
    d.M, d.N, d.K = M, N, K
    d.sam, d.sak, d.sbk, d.sbn = st
    d.ldc, d.batch, d.alpha, d.beta, d.split_k, d.algo = N, 1, 1.0, 0.0, split, 0
    L.check(lib.echr_gemm_f32(C.byref(d), L.stream_ptr()), 'gemm')
    torch.cuda.synchronize()
    err = float((Cc.double() - ref).abs().max() / ref.abs().max())
    return err


if __name__ == '__main__':
    lib.echr_config_set(b'gemm_tile', ord('t'))
    bad = 0
    for lay in ('NT', 'NN', 'TN'):
        for (M, N, K, pad) in [(128, 128, 64, 0), (762, 1536, 5001, 0), (762, 1536, 5001, 3), (130, 257, 100, 0), (5001, 1536, 764, 0), (512, 500, 8192, 0),
                               (513, 502, 1000, 1), (2048, 500, 1280, 0), (300, 131, 67, 2)]:
            for split in (-1, 1):
                e = check(lay, M, N, K, pad, split)
                ok = e < 1e-5          # (one fp32 fma chain over K <= 8192 terms: ~sqrt(K) ulp)
                bad += not ok
                print('check %s %5d %5d %5d pad %d split %2d  rel err %.2e %s' % (lay, M, N, K, pad, split, e, 'ok' if ok else 'BAD'), flush=True)
    lib.echr_config_set(b'gemm_tile', 0)
    if bad:
        sys.exit('value check failed')
    SH = [('logits', 'NT', 762, 5001, 1536), ('dOUTD', 'NT', 762, 1536, 5004), ('dWlogit', 'NT', 5001, 1536, 764), ('gin', 'NT', 1280, 2048, 512),
          ('pall', 'NT', 8192, 512, 500), ('fc1', 'NT', 4096, 512, 512), ('dXT', 'NN', 1280, 512, 2048), ('g_w_hh', 'TN', 2048, 512, 1280),
          ('g_w_att', 'TN', 2048, 500, 1280), ('g_w_h2a', 'TN', 512, 512, 1280), ('g_w_c2a', 'TN', 512, 500, 8192), ('g_w_fc1', 'TN', 512, 512, 4096),
          ('4096^3', 'NT', 4096, 4096, 4096), ('4096^3', 'NN', 4096, 4096, 4096), ('4096^3', 'TN', 4096, 4096, 4096)]
    SH += [('g_w_ih0', 'TN', 2048, 1024, 764), ('g_w_ih2', 'TN', 2048, 612, 764), ('dXT_c', 'NN', 764, 512, 2048), ('dOUTD_NN', 'NN', 764, 1536, 5001)]
    print('%-9s %-3s %5s %5s %5s | %16s %16s %16s %16s %16s %16s' % ('name', 'lay', 'M', 'N', 'K', 'auto', '64x64', 't128', '128x64 (a)', '64x128 (b)', 't128 split 1/2/4/8'))
    for sh in SH:
        cells = []
        for code in (0, ord('6'), ord('t'), ord('a'), ord('b')):
            lib.echr_config_set(b'gemm_tile', code)
            us, tf = run(*sh)
            cells.append('%7.1fus %5.1fTF' % (us, tf))
        sp = []
        for s_ in (1, 2, 4, 8):
            lib.echr_config_set(b'gemm_split', s_)
            us, tf = run(*sh)
            sp.append('%.0f' % us)
        lib.echr_config_set(b'gemm_split', 0)
        lib.echr_config_set(b'gemm_tile', 0)
        print('%-9s %-3s %5d %5d %5d | ' % sh + ' '.join(cells) + '  ' + '/'.join(sp), flush=True)
