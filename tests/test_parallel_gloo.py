"""CPU, world_size 2 (gloo): the data-parallel gradient exchange is a SUM all-reduce without 1/R, so that R ranks x 1
video == the reference's m_batch = R accumulation (train.py:281-283,313-317)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from echr_amd import parallel


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(7, 5), torch.nn.Linear(5, 3), torch.nn.Linear(3, 3))
    # the third layer never receives a gradient (like core.fusion_layer / fusion_model.h2a_layer in the reference)
    rs = np.random.RandomState(10 + rank)
    x = torch.from_numpy(rs.standard_normal((4, 7)).astype(np.float32))
    net[1](net[0](x)).pow(2).sum().backward()
    local = [p.grad.clone() for p in parallel.live_grads(net)]
    nb = parallel.allreduce_gradients(net, bucket_bytes=64)       # tiny buckets: exercise the multi-bucket path
    q.put((rank, nb, [g.numpy() for g in local], [p.grad.numpy().copy() for p in parallel.live_grads(net)],
           [p.grad is None for p in net[2].parameters()]))
    dist.destroy_process_group()


def test_allreduce_is_sum_over_ranks_and_skips_unused_params():
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    (_, nb0, l0, r0, un0), (_, nb1, l1, r1, un1) = res
    assert nb0 == nb1 and nb0 > 1
    assert all(un0) and all(un1)
    for a, b, s0, s1 in zip(l0, l1, r0, r1):
        assert np.allclose(a + b, s0, atol=1e-6) and np.allclose(s0, s1)     # SUM, no averaging; identical on every rank


def test_shard_videos_partition():
    got = sorted(sum((parallel.shard_videos(11, r, 4) for r in range(4)), []))
    assert got == list(range(11))
