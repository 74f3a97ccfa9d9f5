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


class EarlyReducer(object):
    """Overlaps the all-reduce of gradients that are final early in the backward pass with the rest of the backward pass.

    The late-fusion layer (logit.weight / logit.bias: 35 % of the gradient bytes) gets its gradient in the FIRST stage of the
    decoder backward (echr_decoder_bwd phase 1); the reverse recurrence and every other gradient follow.  The decoder Function
    calls `hook(params)` between the two stages; the hook starts an asynchronous SUM all-reduce on the arena range that holds
    those gradients (torch.distributed runs it on the collective stream, ordered after the work already queued on the current
    stream).  `allreduce_gradients` later reduces the remaining ranges and waits for the early one, so the result is the same
    SUM over ranks as the single-collective path."""

    def __init__(self, arena, group=None):
        self.arena, self.group = arena, group
        self.pending = None            # (lo, hi, work)
        self.early_ids = set()         # parameters whose gradients the pending collective covers
        arena.early_grad_hook = self.hook

    def hook(self, params):
        ar = self.arena
        if self.pending is not None or not (dist.is_available() and dist.is_initialized()):
            return
        slots = sorted(ar.slot(p) for p in params)
        if any(s is None for s in slots) or slots != list(range(slots[0], slots[-1] + 1)):
            return                      # not one contiguous arena range: leave everything to the final collective
        lo, hi = ar.span(slots)
        work = dist.all_reduce(ar.flat_g[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self.pending = (lo, hi, work)
        self.early_ids = {id(p) for p in params}

    def finish(self):
        """Reduce what the early collective did not cover, then wait for it.  Returns the number of collectives."""
        ar = self.arena
        if self.pending is None:
            dist.all_reduce(ar.flat_g, op=dist.ReduceOp.SUM, group=self.group)
            return 1
        lo, hi, work = self.pending
        self.pending = None
        n = 1
        for a, b in ((0, lo), (hi, ar.total)):
            if b > a:
                dist.all_reduce(ar.flat_g[a:b], op=dist.ReduceOp.SUM, group=self.group)
                n += 1
        work.wait()
        return n

    def disable(self):
        self.arena.early_grad_hook = None


def enable_overlap(module, group=None):
    """Install the early reducer on a module whose parameters live in a flat arena (CaptionGenerator.build_arena())."""
    arena = getattr(module, '_echr_arena', None)
    if arena is None:
        raise ValueError('enable_overlap needs the flat arena: call module.build_arena() first')
    red = EarlyReducer(arena, group)
    module._echr_early_reducer = red
    return red


def allreduce_gradients(module, group=None, bucket_bytes=64 << 20, force=False):
    """Sum the gradients over ranks in a few large flat buckets (xGMI is point-to-point: few large messages)."""
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size(group) == 1 and not force):
        return 0
    arena = getattr(module, '_echr_arena', None)
    if arena is not None and arena.grads_in_arena():
        # the flat gradient buffer IS the bucket: one collective (or the remainder of an overlapped one), no pack/unpack copies
        arena.zero_unused_grads(keep=True)
        red = getattr(module, '_echr_early_reducer', None)
        if red is not None:
            return red.finish()
        dist.all_reduce(arena.flat_g, op=dist.ReduceOp.SUM, group=group)
        return 1
    params = live_grads(module)
    red = getattr(module, '_echr_early_reducer', None)
    if red is not None and red.pending is not None:      # gradients left the arena after an early collective started: finish it and
        red.pending[2].wait()                             # keep its parameters out of the per-tensor buckets (already summed)
        red.pending = None
        params = [p for p in params if id(p) not in red.early_ids]
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
