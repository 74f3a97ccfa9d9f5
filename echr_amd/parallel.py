"""Data parallelism over the 8 MI355X of a node: one process per GPU, videos sharded across ranks,
ONE collective per optimiser step.

Semantics pinned to the reference's gradient accumulation: `m_batch` videos are SUMMED without averaging,
the accumulated gradient is clamped, then Adam steps (train.py:281-283,313-317).  Data parallel over R ranks
is therefore reference-equivalent to m_batch = R on the same R videos: all-reduce SUM (no 1/R), clamp after
the reduce, identical fused Adam on every rank.  The collective runs on torch.distributed (backend "nccl" is
RCCL over xGMI on ROCm; "gloo" for the CPU tests).
"""
import os

import torch
import torch.distributed as dist


def choose_algo(world, algo=None):
    """The ONE switch for the gradient collective: explicit `algo`, else ECHR_DP_ALGO, else by world size.  'auto': two ranks share one
    xGMI link whatever the algorithm -> 'allreduce'; from four ranks on a ring all-reduce is bound by ONE of the 7 links of a GPU while
    reduce-scatter + all-gather spreads 1/R of the buffer over R - 1 distinct links -> 'rs_ag'.  Unmeasured on
    hardware (this build never had more than one GPU): ECHR_DP_ALGO=allreduce / rs_ag overrides, bench.py prints the choice."""
    algo = algo or os.environ.get('ECHR_DP_ALGO', 'auto')
    if algo == 'auto':
        algo = 'rs_ag' if world >= 4 else 'allreduce'
    if algo not in ('allreduce', 'rs_ag'):
        raise ValueError('ECHR_DP_ALGO must be auto, allreduce or rs_ag (got %r)' % (algo,))
    return algo


class _Works(object):
    """The work handles of one logical SUM (all-reduce: one; reduce-scatter + all-gather: two, plus the shard they exchange through)."""

    def __init__(self, works, keep=None, then=None, algo='allreduce'):
        self.works, self.keep, self.then, self.algo = [w for w in works if w is not None], keep, then, algo

    def wait(self):
        for w in self.works:
            w.wait()
        if self.then is not None:          # (a second collective that could not be queued behind the first asynchronously: see reduce_sum_)
            self.then()
        self.keep = self.then = None


def reduce_sum_(flat, group=None, algo=None, async_op=False):
    """SUM over ranks of a flat fp32 buffer, in place.

    algo None: choose_algo (ECHR_DP_ALGO, default by world size: 'allreduce' for 2 ranks, 'rs_ag' from 4).
    algo 'allreduce': one dist.all_reduce -- RCCL picks ring / tree / direct itself.
    algo 'rs_ag' (or ECHR_DP_ALGO=rs_ag): reduce-scatter + all-gather on the same buffer (on a fully connected
    xGMI node each GPU then exchanges 1/R of the buffer with each of its R-1 peers over R-1 distinct links, instead of pushing the
    whole buffer around a ring whose every hop is bound by ONE link).  Needs numel % R == 0 (the arena and every range of it that the
    early reducers hand over are 64-float aligned, so R = 2, 4, 8 always qualify); otherwise it falls back to all_reduce.
    async_op: returns an object with .wait() (both collectives of rs_ag are queued asynchronously, in order, on the collective stream);
    otherwise None."""
    world = dist.get_world_size(group)
    algo = choose_algo(world, algo)
    if algo == 'rs_ag' and world > 1 and flat.numel() % world == 0 and flat.is_contiguous():
        n = flat.numel() // world
        shard = torch.empty(n, device=flat.device, dtype=flat.dtype)
        w1 = dist.reduce_scatter_tensor(shard, flat, op=dist.ReduceOp.SUM, group=group, async_op=async_op)
        if async_op and dist.get_backend(group) != 'nccl':
            # only RCCL orders two asynchronous collectives (its stream does); gloo runs them on worker threads, where the all-gather
            # could read the shard before the reduce-scatter has written it -> the all-gather follows at wait()
            return _Works([w1], shard, lambda: dist.all_gather_into_tensor(flat, shard, group=group), 'rs_ag')
        w2 = dist.all_gather_into_tensor(flat, shard, group=group, async_op=async_op)
        return _Works([w1, w2], shard, None, 'rs_ag') if async_op else None
    w = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group, async_op=async_op)
    return _Works([w]) if async_op else None


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

    def __init__(self, arena, group=None, auto_arm=True, defer_first=None, staged=None):
        self.arena, self.group = arena, group
        # staged = False (default since round 3, ECHR_DP_STAGED=1 for the old form): the decoder backward stays ONE library call with its
        # asynchronous tail (logit-layer / attention / embedding gradients on the helper stream) and hands over the twelve LSTM-layer gradients
        # (39 % of the bytes), which are final in stream order when the call returns; their all-reduce then overlaps the tail and the event
        # encoder's backward.  The three-stage form (True) makes 74 % of the bytes early-reducible but gives up the asynchronous tail and puts
        # the logit-layer gradient products back in front of the reverse recurrence: +0.2 ms of compute per iteration for ranges that
        # `defer_first` holds back until the recurrence is over anyway.
        self.staged = bool(int(os.environ.get('ECHR_DP_STAGED', '0'))) if staged is None else bool(staged)
        arena.early_staged = self.staged
        self.pending = []              # [(lo, hi, work)], disjoint arena ranges in flight
        # The first range of a backward pass (the late-fusion layer) becomes final BEFORE the reverse recurrence, which runs as a pair
        # of persistent kernels that need every CU of the device: a collective kernel resident beside them would not overlap with
        # them but serialise against them (and hold back the workgroups that found no CU).  defer_first (default on,
        # ECHR_DP_DEFER_FIRST=0 to disable) keeps that range back until the next hook call (after the recurrence), where it starts
        # together with the LSTM layers' range and overlaps with the launch-bound tail of the backward pass.
        self.defer_first = bool(int(os.environ.get('ECHR_DP_DEFER_FIRST', '1'))) if defer_first is None else bool(defer_first)
        self.deferred = []             # parameter lists kept back
        self.early_ids = set()         # parameters whose gradients the pending collectives cover
        # Early collectives are only legal during the LAST backward of an optimiser step: a later backward would add its gradients
        # into arena ranges that are already being reduced (allreduce(g1) + local g2, and a race with the in-flight collective).
        # auto_arm=True (one backward per step, the reference's m_batch = 1): every backward is the last one.  With gradient
        # accumulation pass auto_arm=False and call arm() right before the final backward of the step.
        self.auto_arm = bool(auto_arm)
        self.armed = self.auto_arm
        arena.early_grad_hook = self.hook
        arena.early_reducer = self

    def arm(self):
        """The next backward is the last one of this optimiser step: its final gradients may start their all-reduce early."""
        self.armed = True

    def check_no_backward_while_in_flight(self):
        """Called by the backward Functions when they are about to ACCUMULATE into existing gradients."""
        if self.pending:
            raise RuntimeError('a backward pass accumulates into gradients whose all-reduce is already in flight: with gradient '
                               'accumulation create the reducer with auto_arm=False and call arm() only before the last backward')

    def hook(self, params, after_recurrence=False):
        """after_recurrence: the persistent reverse recurrence of this backward pass is over (nothing to keep back for)."""
        if not self.armed or not (dist.is_available() and dist.is_initialized()):
            return
        if self.defer_first and not after_recurrence and not self.pending and not self.deferred:
            self.deferred.append(list(params))
            return
        held, self.deferred = self.deferred, []
        for ps in held + [list(params)]:
            self._start(ps)

    def _start(self, params):
        ar = self.arena
        slots = sorted(ar.slot(p) for p in params)
        if any(s is None for s in slots) or slots != list(range(slots[0], slots[-1] + 1)):
            return                      # not one contiguous arena range: leave it to the final collective
        lo, hi = ar.span(slots)
        if any(lo < phi and plo < hi for plo, phi, _ in self.pending):
            return                      # overlaps a range already in flight (second backward in one step): final collective
        work = reduce_sum_(ar.flat_g[lo:hi], self.group, None, async_op=True)          # choose_algo: rs_ag from 4 ranks (ranges are 64-float aligned)
        self.pending.append((lo, hi, work))
        self.early_ids.update(id(p) for p in params)

    def finish(self):
        """Reduce what the early collectives did not cover, then wait for them.  Returns the number of collectives."""
        ar = self.arena
        pend = sorted(self.pending, key=lambda t: t[0])
        self.pending, self.early_ids, self.deferred = [], set(), []        # a range still kept back is simply not covered early
        self.armed = self.auto_arm
        n, pos = len(pend), 0
        for lo, hi, _ in pend + [(ar.total, ar.total, None)]:
            if lo > pos:
                reduce_sum_(ar.flat_g[pos:lo], self.group)
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
        self.pending, self.early_ids, self.deferred = [], set(), []
        return ids

    def disable(self):
        self.arena.early_grad_hook = None


def enable_overlap(module, group=None, auto_arm=True, defer_first=None, staged=None):
    """Install the early reducer on a module whose parameters live in a flat arena (CaptionGenerator.build_arena()).
    auto_arm=False: gradient accumulation -- call the returned reducer's arm() before the last backward of every optimiser step.
    defer_first: see EarlyReducer (None = ECHR_DP_DEFER_FIRST, default on).  staged: three-stage decoder backward (None = ECHR_DP_STAGED,
    default off: one call + LSTM-layer gradients reduced early)."""
    arena = getattr(module, '_echr_arena', None)
    if arena is None:
        raise ValueError('enable_overlap needs the flat arena: call module.build_arena() first')
    red = EarlyReducer(arena, group, auto_arm, defer_first, staged)
    module._echr_early_reducer = red
    return red


def allreduce_gradients(module, group=None, bucket_bytes=64 << 20, force=False, algo=None):
    """Sum the gradients over ranks in a few large flat buckets (xGMI is point-to-point: few large messages).
    algo: None (choose_algo: ECHR_DP_ALGO, else 'allreduce' for 2 ranks and 'rs_ag' from 4) | 'allreduce' | 'rs_ag' -- see reduce_sum_."""
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size(group) == 1 and not force):
        return 0
    arena = getattr(module, '_echr_arena', None)
    if arena is not None and arena.grads_in_arena():
        # the flat gradient buffer IS the bucket: one collective (or the remainder of an overlapped one), no pack/unpack copies
        arena.zero_unused_grads(keep=True)
        red = getattr(module, '_echr_early_reducer', None)
        if red is not None:
            return red.finish()
        reduce_sum_(arena.flat_g, group, algo)
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
