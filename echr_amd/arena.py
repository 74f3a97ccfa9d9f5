"""Flat parameter / gradient arena.

Packs every parameter of a module into ONE fp32 device buffer (64-float aligned slots) and gives the backward
Functions matching views of ONE flat gradient buffer.  What it buys on MI355X:
  * a single fused clamp+Adam launch over the whole model (echr_clamp_adam streams 7 x 4 B per parameter once)
    instead of one launch per tensor,
  * one zero-fill per backward group instead of one per split-K product,
  * a single-bucket RCCL all-reduce straight on the gradient buffer (no pack/unpack copies).
Parameter tensors stay ordinary nn.Parameters (state_dict / load_state_dict unchanged); only their storage moves.
"""
import torch

ALIGN = 64


class ParamArena(object):
    def __init__(self, module):
        self.params = [p for p in module.parameters()]
        if not self.params:
            raise ValueError('module has no parameters')
        dev = self.params[0].device
        if any(p.device != dev or p.dtype != torch.float32 for p in self.params):
            raise ValueError('arena needs all parameters in fp32 on one device (move the module first, then build the arena)')
        self.offsets, off = [], 0
        for p in self.params:
            self.offsets.append(off)
            off += (p.numel() + ALIGN - 1) // ALIGN * ALIGN
        self.total = off
        self.flat_p = torch.zeros(off, device=dev, dtype=torch.float32)
        self.flat_g = torch.zeros(off, device=dev, dtype=torch.float32)
        self.index = {}
        self._zeroed = []
        # id of the autograd graph task (backward pass) in which a backward Function zero-filled the WHOLE arena, None otherwise.  The flag is
        # only trusted inside that very pass: an end-of-pass callback clears it, but callbacks do not run when a later node raises, so a
        # stale value must never make the next pass skip its zero fill (whole_zero_pass compares with the running pass's id)
        self._whole_zero_task = None
        self.deferred_clamp = None           # clip value of a clip_gradient call that was left to the fused step kernel (optim.ClampAdam)
        with torch.no_grad():
            for i, (p, o) in enumerate(zip(self.params, self.offsets)):
                v = self.flat_p[o:o + p.numel()].view(p.shape)
                v.copy_(p.data)
                p.data = v                                     # the Parameter object (and its name) is unchanged
                p.grad = None
                self.index[id(p)] = i
        module._echr_arena = self

    def slot(self, p):
        return self.index.get(id(p))

    def grad_view(self, i):
        """A FRESH view tensor of slot i (autograd adopts it as .grad without copying when .grad is None)."""
        p, o = self.params[i], self.offsets[i]
        return self.flat_g[o:o + p.numel()].view(p.shape)

    def span(self, idxs):
        """[start, end) float range from the first to the last of the given slots.  Slots in between that are not listed
        belong to parameters that never receive a gradient (core.fusion_layer); zero is the right content for them."""
        lo, hi = min(idxs), max(idxs)
        end = self.offsets[hi + 1] if hi + 1 < len(self.offsets) else self.total
        return self.offsets[lo], end

    def grads_in_arena(self):
        """True when every existing .grad aliases its arena slot (then flat_g IS the model's gradient)."""
        base = self.flat_g.data_ptr()
        ok = False
        for p, o in zip(self.params, self.offsets):
            if p.grad is None:
                continue
            if p.grad.data_ptr() != base + 4 * o or not p.grad.is_contiguous():
                return False
            ok = True
        return ok

    def params_in_arena(self):
        base = self.flat_p.data_ptr()
        return all(p.data.data_ptr() == base + 4 * o for p, o in zip(self.params, self.offsets))

    def flush_deferred_clamp(self):
        """Apply a clamp that `clip_gradient` deferred to the optimiser kernel: called when another backward is about to accumulate into
        the running gradient (the reference clamps after EVERY backward, train.py:313-317)."""
        clip, self.deferred_clamp = self.deferred_clamp, None
        if clip is not None and self.grads_in_arena():
            from . import functional as EF
            self.zero_unused_grads(keep=True)
            EF.clamp_(self.flat_g, clip)

    @staticmethod
    def _task_id():
        tid = torch._C._current_graph_task_id()
        return tid if tid >= 0 else None

    @property
    def whole_zero_pass(self):
        """True while the backward pass that zero-filled the whole arena is still the one running."""
        return self._whole_zero_task is not None and self._whole_zero_task == self._task_id()

    @whole_zero_pass.setter
    def whole_zero_pass(self, on):
        self._whole_zero_task = self._task_id() if on else None

    def end_backward_pass(self):
        self._whole_zero_task = None

    def note_zeroed(self, lo, hi):
        """A backward Function zero-filled flat_g[lo:hi] this step (GradSink.take)."""
        self._zeroed.append((lo, hi))

    def zero_unused_grads(self, keep=False):
        """Slots whose parameter received no gradient this step must not carry stale values into a flat update.  Slots inside a
        span that a backward Function zero-filled this step are already clean; adjacent remaining slots share one fill.
        keep=True leaves the record in place for a later call in the same step (gradient all-reduce, then the optimiser)."""
        todo = []
        for i, (p, o) in enumerate(zip(self.params, self.offsets)):
            if p.grad is not None:
                continue
            end = self.offsets[i + 1] if i + 1 < len(self.offsets) else self.total
            if any(lo <= o and end <= hi for lo, hi in self._zeroed):
                continue
            if todo and todo[-1][1] == o:
                todo[-1][1] = end
            else:
                todo.append([o, end])
        for lo, hi in todo:
            self.flat_g[lo:hi].zero_()
        if not keep:
            self._zeroed = []
            self._whole_zero_task = None      # (a pass that raised never ran its end-of-pass callback)
