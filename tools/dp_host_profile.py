#!/usr/bin/env python3
"""Host time of one data-parallel iteration on the one-call path (fused.DataParallelStep) with ONE rank on RCCL: where the python / runtime
time of the multi-rank host path goes (cProfile over 200 iterations issued back to back).  GPU box only."""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
os.environ.setdefault('MASTER_PORT', '29533')
import torch
import torch.distributed as dist

import bench


def main():
    torch.cuda.set_device(0)
    args = bench.argparse.Namespace(overlap=False, no_arena=False, fused='auto', c5=False, mode='train')
    if os.environ.get('DP_WARM_FIRST') == '1':          # the library's helper streams exist (and have submitted work) BEFORE RCCL creates its own
        w0 = bench.Workload(args, 'train', 0, torch.device('cuda', 0), False)
        for _ in range(3):
            w0.iteration()
        torch.cuda.synchronize()
        del w0
    if os.environ.get('ECHR_STREAMS_FIRST', '1') != '0':          # what bench.py does: the library's helper streams before RCCL's
        from echr_amd import _lib as L0
        L0.check(L0.load().echr_streams_init(), 'streams_init')
    fd = os.dup(1); os.dup2(2, 1)          # (RCCL prints its version banner to stdout)
    be = os.environ.get('DP_BACKEND', 'nccl')
    if be == 'nccl':
        dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
    elif be == 'gloo':
        dist.init_process_group('gloo', rank=0, world_size=1)
    if be != 'none':
        dist.barrier()
    os.dup2(fd, 1); os.close(fd)
    from echr_amd import _lib
    _lib.load().echr_config_set(b'persist_coop', int(os.environ.get('ECHR_PERSIST_COOP', '1')))          # bench.py's multi-rank default
    wl = bench.Workload(args, 'train', 0, torch.device('cuda', 0), os.environ.get('DP_PLAIN') != '1')          # DP_PLAIN=1: the single-rank path in a process that has RCCL up
    for _ in range(10):
        wl.iteration()
    torch.cuda.synchronize()
    n = 200
    t0 = time.perf_counter()
    for _ in range(n):
        wl.iteration()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print('issue %.3f ms / iteration, wall %.3f ms / iteration' % (1e3 * (t1 - t0) / n, 1e3 * (t2 - t0) / n))
    if os.environ.get('DP_NO_PROFILE') == '1':
        if dist.is_initialized():
            dist.destroy_process_group()
        return
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(n):
        wl.iteration()
    pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr)
    st.sort_stats('cumulative').print_stats(28)
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
