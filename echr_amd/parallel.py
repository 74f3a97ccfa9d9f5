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

    The decoder backward runs in stages (echr_dec_grads.phase) and calls `hook(params)` after each stage with the parameters whose
    gradients just became final: the late-fusion layer (logit.weight / logit.bias, 35 % of the gradient bytes) before the reverse
    recurrence even starts, the three LSTM layers (39 %) right after it.  The hook starts an asynchronous SUM all-reduce on the arena
    range that holds those gradients (torch.distributed runs it on the collective stream, ordered after the work already queued on
    the current stream).  `allreduce_gradients` later reduces the ranges no early collective covered and waits for the early ones,
    so the result is the same SUM over ranks as the single-collective path."""

    def __init__(self, arena, group=None):
        self.arena, self.group = arena, group
        self.pending = []              # [(lo, hi, work)], disjoint arena ranges in flight
        self.early_ids = set()         # parameters whose gradients the pending collectives cover
        arena.early_grad_hook = self.hook

    def hook(self, params):
        ar = self.arena
        if not (dist.is_available() and dist.is_initialized()):
            return
        slots = sorted(ar.slot(p) for p in params)
        if any(s is None for s in slots) or slots != list(range(slots[0], slots[-1] + 1)):
            return                      # not one contiguous arena range: leave it to the final collective
        lo, hi = ar.span(slots)
        if any(lo < phi and plo < hi for plo, phi, _ in self.pending):
            return                      # overlaps a range already in flight (second backward in one step): final collective
        work = dist.all_reduce(ar.flat_g[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self.pending.append((lo, hi, work))
        self.early_ids.update(id(p) for p in params)

    def finish(self):
        """Reduce what the early collectives did not cover, then wait for them.  Returns the number of collectives."""
        ar = self.arena
        pend = sorted(self.pending, key=lambda t: t[0])
        self.pending, self.early_ids = [], set()
        n, pos = len(pend), 0
        for lo, hi, _ in pend + [(ar.total, ar.total, None)]:
            if lo > pos:
                dist.all_reduce(ar.flat_g[pos:lo], op=dist.ReduceOp.SUM, group=self.group)
                n += 1
            pos = max(pos, hi)
        for _, _, work in pend:
            work.wait()
        return n

    def wait_pending(self):
        """Finish the early collectives only (used when the gradients left the arena afterwards); returns the covered ids."""
        ids = self.early_ids
        for _, _, work in self.pending:
            work.wait()
        self.pending, self.early_ids = [], set()
        return ids

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
    if red is not None and red.pending:                   # gradients left the arena after early collectives started: finish them and
        done = red.wait_pending()                         # keep their parameters out of the per-tensor buckets (already summed)
        params = [p for p in params if id(p) not in done]
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
