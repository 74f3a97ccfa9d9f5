"""Fused clamp + Adam optimiser (reference: misc/utils.py:107-111 clip_gradient + torch.optim.Adam as wired at
train.py:201-209,315-317: betas (optim_alpha, optim_beta), eps, weight_decay 0, no amsgrad)."""
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

    def _flat_step(self, clip):
        """Whole-model update in ONE kernel launch; valid when all parameters and all live gradients alias the arena."""
        ar = self.arena
        if ar is None or len(self.param_groups) != 1 or not ar.params_in_arena() or not ar.grads_in_arena():
            return False
        if {id(p) for p in self.param_groups[0]['params']} != {id(p) for p in ar.params}:
            return False
        group = self.param_groups[0]
        if self._flat is None:
            if any(self.state[p] for p in ar.params):
                return False                  # per-tensor state already exists (resumed run): keep the per-tensor path
            self._flat = dict(step=0, m=torch.zeros_like(ar.flat_p), v=torch.zeros_like(ar.flat_p))
        ar.zero_unused_grads()                # never-used parameters: g = 0 -> m = v = 0 -> no update (== Adam skipping them)
        st = self._flat
        st['step'] += 1
        b1, b2 = group['betas']
        EF.clamp_adam_(ar.flat_p, ar.flat_g, st['m'], st['v'], st['step'], group['lr'], b1, b2, group['eps'], clip)
        return True

    @torch.no_grad()
    def step(self, closure=None):
        clip = self.pending_clip if self.pending_clip is not None else self.grad_clip
        self.pending_clip = None
        clip = float('inf') if clip is None else float(clip)
        if self._flat_step(clip):
            return None
        if self._flat is not None:
            raise RuntimeError('ClampAdam: gradients left the flat arena after flat optimiser state was created')
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
                EF.clamp_adam_(p.data, g, st['exp_avg'], st['exp_avg_sq'], st['step'], group['lr'], b1, b2, group['eps'], clip)
        return None
