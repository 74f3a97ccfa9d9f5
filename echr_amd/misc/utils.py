"""Criteria and optimiser helpers with the reference's names (misc/utils.py:15-122)."""
import torch
import torch.nn as nn

from .. import functional as EF


def if_use_att(caption_model):
    return not (caption_model in ('show_tell', 'all_img', 'fc') or 'allimg' in caption_model)


def decode_sequence(ix_to_word, seq):
    """Index rows -> strings; 0 terminates a row (misc/utils.py:24-38)."""
    rows = seq.tolist() if hasattr(seq, 'tolist') else seq
    out = []
    for row in rows:
        words = []
        for ix in row:
            if ix <= 0:
                break
            words.append(ix_to_word[str(int(ix))])
        out.append(' '.join(words))
    return out


class LanguageModelCriterion(nn.Module):
    """Masked NLL over log-probs [N,S,V+1] (misc/utils.py:62-75), evaluated by echr_nll_loss_fwd."""

    def forward(self, input, target, mask):
        # when `input` comes straight from the native decoder, its backward takes the criterion's gradient in fused form
        # (echr_dec_grads.nll_*: softmax - one-hot in one pass) instead of a dense [N,S,V+1] tensor; see MaskedNLL.backward
        node = input.grad_fn if (EF.FUSED_NLL[0] and type(input.grad_fn).__name__ == 'DecoderFunctionBackward') else None
        return EF.MaskedNLL.apply(input, target.to(input.device), mask.to(input.device), node)


class TAPModelCriterion(nn.Module):
    """Weighted BCE of the proposal head (misc/utils.py:78-99), evaluated by echr_tap_bce_fwd/bwd."""

    def forward(self, scores, masks, labels, w1):
        return EF.TapBCE.apply(scores, masks.to(scores.device), labels.to(scores.device), w1.to(scores.device))


def set_lr(optimizer, lr):
    for group in optimizer.param_groups:
        group['lr'] = lr


def clip_gradient(optimizer, grad_clip):
    """Element-wise clamp of every gradient to +-grad_clip (misc/utils.py:107-111).

    With echr_amd.optim.ClampAdam the clamp is folded into the fused step kernel (it is recorded here and
    applied inside `step()`); for any other optimiser the clamp runs as its own HIP kernel per tensor."""
    from ..optim import ClampAdam
    if isinstance(optimizer, ClampAdam):
        # The reference clamps the RUNNING gradient after every backward (train.py:313-317): with m_batch > 1 it computes
        # clamp(clamp(g1) + g2).  Per-tensor gradients are clamped in place here.  With the flat arena (ClampAdam.defer_clamp, default on)
        # the clamp is left to the fused step kernel and, should another backward accumulate before the step, applied right before that
        # accumulation -- the same trajectory for any m_batch; only `.grad` read between this call and step() shows unclamped values
        # (set optimizer.defer_clamp = False to clamp in place here, one more 174 MB pass).
        optimizer.clamp_grads_(float(grad_clip))
        optimizer.pending_clip = float(grad_clip)
        return
    for group in optimizer.param_groups:
        for p in group['params']:
            if p.grad is not None:
                EF.clamp_(p.grad.data, grad_clip)


def fix_model_parameters(model):
    for p in model.parameters():
        p.requires_grad = False


def unfix_model_parameters(model):
    for p in model.parameters():
        p.requires_grad = True
