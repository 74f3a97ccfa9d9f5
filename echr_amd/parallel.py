"""Data parallelism over the 8 MI355X of a node: one process per GPU, videos sharded across ranks,
ONE collective per optimiser step.

Semantics pinned to the reference's gradient accumulation: `m_batch` videos are SUMMED without averaging,
the accumulated gradient is clamped, then Adam steps (train.py:281-283,313-317).  Data parallel over R ranks
is therefore reference-equivalent to m_batch = R on the same R videos: all-reduce SUM (no 1/R), clamp after
the reduce, identical fused Adam on every rank.  The collective runs on torch.distributed (backend "nccl" is
RCCL over xGMI on ROCm; "gloo" for the CPU tests).
"""
import torch
import torch.distributed as dist


def live_grads(module):
    """Parameters that actually receive gradients (core.fusion_layer / fusion_model.h2a_layer never do)."""
    return [p for p in module.parameters() if p.grad is not None]


def allreduce_gradients(module, group=None, bucket_bytes=64 << 20, force=False):
    """Sum the gradients over ranks in a few large flat buckets (xGMI is point-to-point: few large messages)."""
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size(group) == 1 and not force):
        return 0
    arena = getattr(module, '_echr_arena', None)
    if arena is not None and arena.grads_in_arena():
        # the flat gradient buffer IS the bucket: one collective, no pack/unpack copies
        arena.zero_unused_grads(keep=True)
        dist.all_reduce(arena.flat_g, op=dist.ReduceOp.SUM, group=group)
        return 1
    params = live_grads(module)
    buckets, cur, cur_bytes = [], [], 0
    for p in params:
        nb = p.grad.numel() * p.grad.element_size()
        if cur and cur_bytes + nb > bucket_bytes:
            buckets.append(cur)
            cur, cur_bytes = [], 0
        cur.append(p)
        cur_bytes += nb
    if cur:
        buckets.append(cur)
    works = []
    for b in buckets:
        flat = torch.cat([p.grad.reshape(-1) for p in b])
        works.append((dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group, async_op=True), flat, b))
    for w, flat, b in works:
        w.wait()
        off = 0
        for p in b:
            n = p.grad.numel()
            p.grad.copy_(flat[off:off + n].view_as(p.grad))
            off += n
    return len(buckets)


def shard_videos(n_videos, rank, world):
    """Indices of the videos rank `rank` processes (round-robin: independent units, no data-path collective)."""
    return list(range(rank, n_videos, world))
