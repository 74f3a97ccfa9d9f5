"""CPU, world_size 2 (gloo): the data-parallel gradient exchange is a SUM all-reduce without 1/R, so that R ranks x 1
video == the reference's m_batch = R accumulation (train.py:281-283,313-317)."""
import os

import pytest
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


def _worker_early(rank, world, port, q, algo='auto'):
    """Arena path with the early (overlapped) collective: hook on the middle layer, then the remainder at the end.
    algo 'rs_ag': the early ranges (64-float aligned arena slots) go through reduce-scatter + all-gather, as choose_algo picks from 4 ranks on."""
    from echr_amd.arena import ParamArena
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    os.environ['ECHR_DP_ALGO'] = algo
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(7, 5), torch.nn.Linear(5, 3), torch.nn.Linear(3, 3))
    arena = ParamArena(net)
    rs = np.random.RandomState(20 + rank)
    x = torch.from_numpy(rs.standard_normal((4, 7)).astype(np.float32))
    net[1](net[0](x)).pow(2).sum().backward()
    local = [p.grad.clone() for p in parallel.live_grads(net)]
    # move the gradients into the arena the way the backward Functions leave them there (views of flat_g adopted as .grad)
    for p in list(net[0].parameters()) + list(net[1].parameters()):
        v = arena.grad_view(arena.slot(p))
        v.copy_(p.grad)
        p.grad = v
    assert arena.grads_in_arena()
    arena.flat_g[arena.offsets[arena.slot(net[2].weight)]] = 123.0      # stale junk in a never-used slot must not survive
    red = parallel.enable_overlap(net)
    red.hook([net[1].weight, net[1].bias])
    red.hook([net[0].weight])                      # a second early range, adjacent to nothing in flight
    red.hook([net[1].bias])                        # overlaps a range in flight: must be ignored
    assert len(red.pending) == 2
    assert all(w.algo == ('rs_ag' if algo == 'rs_ag' else 'allreduce') for _, _, w in red.pending)          # rs_ag: reduce-scatter + all-gather per range
    n = parallel.allreduce_gradients(net)
    q.put((rank, n, [g.numpy() for g in local], [p.grad.numpy().copy() for p in parallel.live_grads(net)], arena.flat_g.numpy().copy()))
    dist.destroy_process_group()


@pytest.mark.parametrize('algo', ['auto', 'rs_ag'])
def test_early_reducer_equals_single_allreduce(algo):
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker_early, args=(r, world, port, q, algo)) for r in range(world)]
    for p in ps:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    (_, n0, l0, r0, f0), (_, n1, l1, r1, f1) = res
    assert n0 == n1 == 4                                   # two early ranges + the gap between them + the tail
    for a, b, s0, s1 in zip(l0, l1, r0, r1):
        assert np.allclose(a + b, s0, atol=1e-6) and np.array_equal(s0, s1)
    assert np.array_equal(f0, f1) and not (f0 == 246.0).any() and not (f0 == 123.0).any()


def _worker_algos(rank, world, port, q):
    """reduce-scatter + all-gather == all-reduce on the arena buffer; accumulation (two backwards per step) with the early reducer
    armed only before the last backward == the plain SUM, identical on every rank; a mis-armed reducer refuses."""
    from echr_amd.arena import ParamArena
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(7, 5), torch.nn.Linear(5, 3), torch.nn.Linear(3, 3))
    arena = ParamArena(net)
    live = list(net[0].parameters()) + list(net[1].parameters())
    rs = np.random.RandomState(30 + rank)

    def backward_into_arena(first):
        x = torch.from_numpy(rs.standard_normal((4, 7)).astype(np.float32))
        gs = torch.autograd.grad(net[1](net[0](x)).pow(2).sum(), live)
        for p, g in zip(live, gs):
            if first:
                v = arena.grad_view(arena.slot(p))
                v.copy_(g)
                p.grad = v
            else:
                p.grad += g                       # what autograd does on the second backward: in place, into the arena view
        return [g.clone() for g in gs]

    # (1) rs_ag vs allreduce
    g1 = backward_into_arena(True)
    ref = arena.flat_g.clone()
    dist.all_reduce(ref)
    n = parallel.allreduce_gradients(net, algo='rs_ag')
    same_algo = bool(torch.equal(ref, arena.flat_g)) and n == 1
    # (2) accumulation with the early reducer: armed before the last backward only
    for p in live:
        p.grad = None
    arena.flat_g.zero_()
    red = parallel.enable_overlap(net, auto_arm=False, defer_first=False)
    ga = backward_into_arena(True)
    red.hook([net[1].weight, net[1].bias])          # not armed: must do nothing
    none_in_flight = len(red.pending) == 0
    gb = backward_into_arena(False)
    red.arm()
    red.hook([net[1].weight, net[1].bias])          # final gradients of the last backward: early range
    in_flight = len(red.pending) == 1
    refused = False
    try:
        red.check_no_backward_while_in_flight()
    except RuntimeError:
        refused = True
    parallel.allreduce_gradients(net)
    local = [a + b for a, b in zip(ga, gb)]
    reduced = [p.grad.numpy().copy() for p in live]
    # (3) defer_first (the default): the range that is final before the reverse recurrence is kept back until the next hook call
    red.disable()
    red2 = parallel.enable_overlap(net, auto_arm=True)
    keep = arena.flat_g.clone()
    red2.hook([net[1].weight, net[1].bias])
    held = red2.defer_first and len(red2.pending) == 0 and len(red2.deferred) == 1
    red2.hook([net[0].weight])
    both = len(red2.pending) == 2 and not red2.deferred
    parallel.allreduce_gradients(net)
    ref2 = keep.clone()
    dist.all_reduce(ref2)
    deferred_ok = bool(held and both and torch.equal(ref2, arena.flat_g))
    q.put((rank, same_algo, none_in_flight, in_flight, refused, not red.armed and deferred_ok, [t.numpy() for t in local], reduced))
    dist.destroy_process_group()


def test_rs_ag_and_accumulation_with_early_reducer():
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker_algos, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    for r in res:
        assert r[1] and r[2] and r[3] and r[4] and r[5], r[:6]
    (_, *_, l0, s0), (_, *_, l1, s1) = res
    for a, b, x, y in zip(l0, l1, s0, s1):
        assert np.allclose(a + b, x, atol=1e-6) and np.array_equal(x, y)          # plain SUM of both ranks' accumulated gradients


def test_shard_videos_partition():
    got = sorted(sum((parallel.shard_videos(11, r, 4) for r in range(4)), []))
    assert got == list(range(11))


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` without an outer torchrun must start the two ranks itself (before any GPU call), rank 0 prints ONE JSON
    line, the exit code is the ranks'.  ECHR_BENCH_DRYRUN stops each rank after rendezvous + one MAX all-reduce (no GPU here)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, ECHR_BENCH_BACKEND='gloo', ECHR_BENCH_DRYRUN='1')
    env.pop('WORLD_SIZE', None)
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0'], env=env, cwd=root,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out == {'dryrun': True, 'n_gpus': 2, 'max_rank_plus_1': 2.0}
    # a world size that does not match --gpus is refused with a message, not an assertion
    env2 = dict(env, WORLD_SIZE='1', RANK='0', LOCAL_RANK='0')
    env2.pop('ECHR_BENCH_DRYRUN')
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2'], env=env2, cwd=root, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and 'launches the ranks itself' in r.stderr


def test_collective_choice_by_world_size(monkeypatch):
    """parallel.choose_algo: one switch (ECHR_DP_ALGO) over all-reduce vs reduce-scatter + all-gather, default by world size."""
    from echr_amd import parallel
    monkeypatch.delenv('ECHR_DP_ALGO', raising=False)
    assert [parallel.choose_algo(w) for w in (1, 2, 4, 8)] == ['allreduce', 'allreduce', 'rs_ag', 'rs_ag']
    monkeypatch.setenv('ECHR_DP_ALGO', 'allreduce')
    assert parallel.choose_algo(8) == 'allreduce'
    monkeypatch.setenv('ECHR_DP_ALGO', 'rs_ag')
    assert parallel.choose_algo(2) == 'rs_ag' and parallel.choose_algo(2, 'allreduce') == 'allreduce'
    monkeypatch.setenv('ECHR_DP_ALGO', 'ring')
    with pytest.raises(ValueError):
        parallel.choose_algo(2)
