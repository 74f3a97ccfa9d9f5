"""Fused clamp + Adam optimiser (reference: misc/utils.py:107-111 clip_gradient + torch.optim.Adam as wired at
train.py:201-209,315-317: betas (optim_alpha, optim_beta), eps, weight_decay 0, no amsgrad)."""
import torch

from . import functional as EF


class ClampAdam(torch.optim.Optimizer):
    """`clip_gradient(optimizer, c); optimizer.step()` as one HIP kernel pass per parameter tensor.

    Parameters whose .grad is None are skipped like torch.optim.Adam does (the reference model has two
    never-used parameter groups: core.fusion_layer and fusion_model.h2a_layer)."""

    def __init__(self, params, lr=5e-5, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, grad_clip=None):
        if weight_decay != 0:
            raise NotImplementedError('the ECHR recipe uses weight_decay=0 (opts.py:215)')
        super(ClampAdam, self).__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.grad_clip = grad_clip
        self.pending_clip = None

    @torch.no_grad()
    def step(self, closure=None):
        clip = self.pending_clip if self.pending_clip is not None else self.grad_clip
        self.pending_clip = None
        clip = float('inf') if clip is None else float(clip)
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
