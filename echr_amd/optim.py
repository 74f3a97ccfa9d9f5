"""Fused clamp + Adam optimiser (reference: misc/utils.py:107-111 clip_gradient + torch.optim.Adam as wired at
train.py:201-209,315-317: betas (optim_alpha, optim_beta), eps, weight_decay 0, no amsgrad)."""
import collections

import torch

from . import functional as EF


class ClampAdam(torch.optim.Optimizer):
    """`clip_gradient(optimizer, c); optimizer.step()` as one HIP kernel pass per parameter tensor.

    Parameters whose .grad is None are skipped like torch.optim.Adam does (the reference model has two
    never-used parameter groups: core.fusion_layer and fusion_model.h2a_layer)."""

    def __init__(self, params, lr=5e-5, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, grad_clip=None, arena=None):
        if weight_decay != 0:
            raise NotImplementedError('the ECHR recipe uses weight_decay=0 (opts.py:215)')
        super(ClampAdam, self).__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.grad_clip = grad_clip
        self.pending_clip = None
        self.arena = arena            # echr_amd.arena.ParamArena: one launch over the whole model when gradients live there
        self._flat = None
        self._same_params = None
        # True (default): with the flat arena `clip_gradient` does not run a clamp pass of its own -- the fused step kernel clamps on the
        # fly (one 174 MB pass less per step).  The reference clamps the RUNNING gradient after every backward (train.py:313-317), which
        # only differs from a single clamp when ANOTHER backward follows (m_batch > 1): the arena remembers the deferred clip value and
        # the next backward that accumulates applies it first (GradSink.usable -> arena.flush_deferred_clamp), so the trajectory is the
        # reference's clamp(clamp(g1) + g2) for any m_batch.  What deferral does change: `.grad` read between clip_gradient and step()
        # holds the unclamped values.  False: clip_gradient clamps in place right away (reference-visible .grad, one more pass).
        self.defer_clamp = True
        # Step counts against updates the device skipped (an aborted persistent launch: the optimiser kernels queued behind it return without
        # touching p / m / v, include/echr_hip.h echr_check_async).  The host counts a step when it QUEUES the update; the device counts it,
        # in this optimiser's own word, when it APPLIES it (echr_clamp_adam_counted; the first launch of a step() carries the word).  After
        # a -62 -- which every site surfaces through _lib.check -- the last (issued - applied) steps are wound back.  Per optimiser, so two
        # optimisers stepping in one process (the joint 'tap_cg' iteration) never see each other's skips.
        self._applied = None          # int32 device tensor [1]
        self._issued = 0
        self._step_log = collections.deque(maxlen=1024)       # per counted step(): the state dicts whose 'step' it incremented
        EF.L.ABORT_LISTENERS.add(self)

    def applied_counter(self, device):
        if self._applied is None:
            self._applied = torch.zeros(1, device=device, dtype=torch.int32)
        return self._applied

    def _count_step(self, states):
        """The host side of one queued update: `states` are the dicts whose 'step' was just incremented."""
        self._issued += 1
        self._step_log.append(list(states))

    def _on_async_abort(self):
        """Called by _lib.check when a library call returned -62 (the device is idle by then): wind the step counts back to what happened."""
        if self._applied is None or self._issued == 0:
            return
        applied = int(self._applied.item())
        lost = self._issued - applied
        for _ in range(max(0, min(lost, len(self._step_log)))):
            for st in self._step_log.pop():
                st['step'] = max(0, int(st['step']) - 1)
        self._issued = applied

    def _flat_step(self, clip):
        """Whole-model update in ONE kernel launch; valid when all parameters and all live gradients alias the arena."""
        ar = self.arena
        if ar is None or len(self.param_groups) != 1 or not ar.params_in_arena() or not ar.grads_in_arena():
            return False
        if self._same_params is None:          # the optimiser's parameter set == the arena's (checked once: both lists are fixed after construction)
            self._same_params = {id(p) for p in self.param_groups[0]['params']} == {id(p) for p in ar.params}
        if not self._same_params:
            return False
        group = self.param_groups[0]
        if self._flat is None:
            if any(self.state[p] for p in ar.params):
                return False                  # per-tensor state already exists (resumed run): keep the per-tensor path
            self._flat = dict(step=0, m=torch.zeros_like(ar.flat_p), v=torch.zeros_like(ar.flat_p))
        ar.zero_unused_grads()                # never-used parameters: g = 0 -> m = v = 0 -> no update (== Adam skipping them)
        st = self._flat
        st['step'] += 1
        self._count_step([st])
        b1, b2 = group['betas']
        EF.clamp_adam_(ar.flat_p, ar.flat_g, st['m'], st['v'], st['step'], group['lr'], b1, b2, group['eps'], clip,
                       applied=self.applied_counter(ar.flat_p.device))
        return True

    @torch.no_grad()
    def step_flat_raw(self, clip=None):
        """clip_gradient + step on the flat arena for callers that filled `arena.flat_g` through raw pointers (fused.JointTrainStep: the
        proposal encoder's backward writes its gradients there without autograd): the arena's gradient buffer IS this step's gradient, slots of
        parameters that received none are zero.  One launch, the same kernel and step accounting as step()."""
        ar = self.arena
        if ar is None or len(self.param_groups) != 1 or not ar.params_in_arena():
            raise RuntimeError('step_flat_raw needs the flat arena')
        if self._flat is None:
            if any(self.state[p] for p in ar.params):
                raise RuntimeError('per-tensor optimiser state exists: load it with load_state_dict on an arena optimiser first')
            self._flat = dict(step=0, m=torch.zeros_like(ar.flat_p), v=torch.zeros_like(ar.flat_p))
        group, st = self.param_groups[0], self._flat
        clip = self.grad_clip if clip is None else clip
        st['step'] += 1
        self._count_step([st])
        b1, b2 = group['betas']
        EF.clamp_adam_(ar.flat_p, ar.flat_g, st['m'], st['v'], st['step'], group['lr'], b1, b2, group['eps'], float('inf') if clip is None else float(clip),
                       applied=self.applied_counter(ar.flat_p.device))

    @torch.no_grad()
    def clamp_grads_(self, clip, fused_step_follows=False):
        """In-place element-wise clamp of every live gradient (misc/utils.py:107-111).  With the flat arena: one launch."""
        ar = self.arena
        if ar is not None and ar.grads_in_arena():
            if self.defer_clamp:
                ar.deferred_clamp = float(clip)     # the fused step kernel clamps; a further backward before it applies the clamp first
                return
            ar.zero_unused_grads(keep=True)
            EF.clamp_(ar.flat_g, clip)
            return
        for group in self.param_groups:
            for p in group['params']:
                if p.grad is not None:
                    EF.clamp_(p.grad.data if p.grad.is_contiguous() else p.grad.data.contiguous(), clip)

    # ---- checkpoint interop (train.py:214-216,456-461 save / restore cg_optimizer.state_dict()) --------------------------------
    def _export_flat(self):
        """Flat arena state -> torch.optim.Adam's per-parameter entries (step, exp_avg, exp_avg_sq); parameters that never received a
        gradient have no entry, exactly like torch.optim.Adam (their flat moments are identically zero)."""
        ar, st = self.arena, self._flat
        for p, o in zip(ar.params, ar.offsets):
            n = p.numel()
            m, v = st['m'][o:o + n], st['v'][o:o + n]
            if p.grad is None and not bool(m.any()) and not bool(v.any()):
                continue
            self.state[p] = dict(step=torch.tensor(float(st['step'])), exp_avg=m.view(p.shape).clone(), exp_avg_sq=v.view(p.shape).clone())

    def state_dict(self):
        """torch.optim.Adam's layout whichever path ran: a reference `cg_optimizer` blob and ours are interchangeable."""
        if self._flat is not None:
            saved = self.state
            self.state = type(saved)()
            try:
                self._export_flat()
                return super(ClampAdam, self).state_dict()
            finally:
                self.state = saved
        return super(ClampAdam, self).state_dict()

    def load_state_dict(self, state_dict):
        """Accepts torch.optim.Adam's layout (ours or the reference's).  With an arena the per-parameter moments are folded back into
        the flat buffers, so a resumed run stays on the single-launch path; all parameters must then share one step count (they do:
        every live parameter is updated at every step)."""
        super(ClampAdam, self).load_state_dict(state_dict)
        self._flat = None
        self._step_log.clear()                     # (the logged state dicts were just replaced)
        self._issued = 0
        if self._applied is not None:
            self._applied.zero_()
        ar = self.arena
        if ar is None or not self.state:
            return
        steps = {int(float(st['step'])) for st in self.state.values() if 'step' in st}
        if len(steps) != 1 or {id(p) for p in self.param_groups[0]['params']} != {id(p) for p in ar.params} or len(self.param_groups) != 1:
            return                                   # keep the per-tensor path (still correct, one launch per tensor)
        flat = dict(step=steps.pop(), m=torch.zeros_like(ar.flat_p), v=torch.zeros_like(ar.flat_p))
        for p, o in zip(ar.params, ar.offsets):
            st = self.state.get(p)
            if st:
                n = p.numel()
                flat['m'][o:o + n].copy_(st['exp_avg'].reshape(-1).to(flat['m']))
                flat['v'][o:o + n].copy_(st['exp_avg_sq'].reshape(-1).to(flat['v']))
        self.state.clear()
        self._flat = flat

    def zero_grad(self, set_to_none=True):
        if self.arena is not None:
            self.arena.deferred_clamp = None
            self.arena.end_backward_pass()
        return super(ClampAdam, self).zero_grad(set_to_none)

    @torch.no_grad()
    def step(self, closure=None):
        EF.L.check(EF.L.load().echr_check_async(), 'ClampAdam.step')      # an aborted persistent launch of this iteration surfaces here at the latest
        if self.arena is not None:
            self.arena.deferred_clamp = None
        clip = self.pending_clip if self.pending_clip is not None else self.grad_clip
        self.pending_clip = None
        clip = float('inf') if clip is None else float(clip)
        if self._flat_step(clip):
            return None
        if self._flat is not None:
            raise RuntimeError('ClampAdam: gradients left the flat arena after flat optimiser state was created')
        stepped = []
        for group in self.param_groups:
            b1, b2 = group['betas']
            for p in group['params']:
                if p.grad is None:
                    continue
                st = self.state[p]
                if not st:
                    st['step'] = 0
                    st['exp_avg'] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    st['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                st['step'] += 1
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                # (the abort word is sticky until the host acknowledges it, so the launches of one step() are applied or skipped together:
                # the first one carries the optimiser's applied-update word)
                EF.clamp_adam_(p.data, g, st['exp_avg'], st['exp_avg_sq'], st['step'], group['lr'], b1, b2, group['eps'], clip,
                               applied=None if stepped or not p.is_cuda else self.applied_counter(p.device))
                stepped.append(st)
        if stepped:
            self._count_step(stepped)
        return None
