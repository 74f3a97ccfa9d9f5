#!/usr/bin/env python3
"""Accuracy + speed of the h2-packed GEMM (echr_h2_pack + echr_gemm_f32 algo=ECHR_GEMM_H2) on the c3 shapes (GPU box only)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from echr_amd import _lib as L

lib = L.load()
dev = torch.device('cuda')


def pack(x, transposed=False):
    """x: [R,K] (or, transposed=True, a [K,R] tensor whose transpose is the operand)"""
    if transposed:
        K, R = x.shape
        s_row, s_col = 1, x.stride(0)
    else:
        R, K = x.shape
        s_row, s_col = x.stride(0), 1
    buf = torch.empty(lib.echr_h2_bytes(R, K), device=dev, dtype=torch.uint8)
    L.check(lib.echr_h2_pack(x.data_ptr(), R, K, s_row, s_col, buf.data_ptr(), L.stream_ptr()), 'pack')
    return buf


def desc(Ax, Bx, Cc, M, N, K, split=-1):
    d = L.GemmDesc()
    d.A, d.B, d.C = Ax.data_ptr(), Bx.data_ptr(), Cc.data_ptr()
    d.M, d.N, d.K = M, N, K
    d.sam, d.sak, d.sbk, d.sbn = K, 1, 1, K
    d.ldc, d.batch, d.alpha, d.beta, d.split_k, d.algo = N, 1, 1.0, 0.0, split, 2
    return d


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


SHAPES = [('logits', 1280, 5001, 1536), ('dOUT', 1280, 1536, 5001), ('g_w_logit', 5001, 1536, 1280), ('gin', 1280, 2048, 512),
          ('dXT', 1280, 512, 2048), ('g_w_hh', 2048, 512, 1280), ('g_w_c2a', 512, 500, 8192), ('pall', 8192, 512, 500),
          ('fc1', 4096, 512, 512), ('big', 4096, 4096, 4096), ('odd', 77, 130, 45), ('gin_x3', 3840, 2048, 512),
          ('wg_x3', 6144, 512, 1280), ('dx_x3', 1280, 1536, 2048), ('fc1_eval', 1000000, 512, 512),
          ('logits_c', 762, 5001, 1536), ('dOUT_c', 762, 1536, 5001), ('g_w_logit_c', 5001, 1536, 762)]      # _c: the active rows of the bench workload
if __name__ == '__main__':
    only = os.environ.get('H2_ONLY')
    for name, M, N, K in SHAPES:
        if only and name not in only.split(','):
            continue
        torch.manual_seed(0)
        A, B = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev)
        if name == 'odd':      # wide dynamic range: per-row scales 1e-9..1e2 and per-element exponents over 20 binades
            A = A * torch.exp(torch.empty(M, 1, device=dev).uniform_(-20, 5)) * torch.exp2(torch.randint(-10, 10, (M, K), device=dev).float())
            B = B * 1e-3
        Ax, Bx = pack(A), pack(B)
        Bt = B.t().contiguous()
        Bx2 = pack(Bt, transposed=True)
        assert torch.equal(Bx, Bx2), 'transposed pack differs'
        Cc = torch.zeros(M, N, device=dev)
        d = desc(Ax, Bx, Cc, M, N, K, split=int(os.environ.get('H2_SPLIT', '-1')))
        L.check(lib.echr_gemm_f32(C.byref(d), L.stream_ptr()), 'gemm')
        ref = A.double() @ B.double().t()
        bound = (A.abs().double() @ B.abs().double().t())
        err = ((Cc.double() - ref).abs() / bound).max().item()
        us = timeit(lambda: lib.echr_gemm_f32(C.byref(d), L.stream_ptr()))
        usp = timeit(lambda: lib.echr_h2_pack(B.data_ptr(), N, K, K, 1, Bx.data_ptr(), L.stream_ptr()))
        uspt = timeit(lambda: lib.echr_h2_pack(Bt.data_ptr(), N, K, 1, N, Bx.data_ptr(), L.stream_ptr()))
        print('%-10s %5d %5d %5d | gemm %7.1f us %6.1f TF/s | err/bound %.2e | pack B %6.1f us (T: %6.1f us) %5.2f TB/s' %
              (name, M, N, K, us, 2.0 * M * N * K / us / 1e6, err, usp, uspt, (N * K * 10.0) / usp / 1e6), flush=True)
