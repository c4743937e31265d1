"""``AULoss`` - pos-weighted BCE-with-logits over the 12 action units (reference models/loss.py:63-103),
computed by one HIP kernel (forward value and the logits gradient in the same launch)."""
from __future__ import annotations

import torch
from torch import nn

from . import ops

# reference models/loss.py:73
AU_POS_WEIGHT = (1., 1., 1., 1., 1., 1., 1., 3., 3., 3., 1., 2.)


class _AULossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y_pred, y_true, pos_weight, ignore):
        if y_pred.stride(-1) != 1:
            y_pred = y_pred.contiguous()
        if y_true.stride(-1) != 1 or y_true.dtype != torch.float32:
            y_true = y_true.to(torch.float32).contiguous()
        loss, grad = ops.au_loss(y_pred.to(torch.float32), y_true, pos_weight, ignore)
        ctx.save_for_backward(grad)
        return loss

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return grad * g, None, None, None


class _AULossSumFn(torch.autograd.Function):
    """(sum over kept rows of the row-mean BCE, kept rows): numerator and denominator of loss.py:85-102, kept apart so
    that a data-parallel wrapper can reduce both over the ranks before dividing"""

    @staticmethod
    def forward(ctx, y_pred, y_true, pos_weight, ignore):
        if y_pred.stride(-1) != 1:
            y_pred = y_pred.contiguous()
        if y_true.stride(-1) != 1 or y_true.dtype != torch.float32:
            y_true = y_true.to(torch.float32).contiguous()
        sc, grad = ops.au_loss_sum(y_pred.to(torch.float32), y_true, pos_weight, ignore)
        ctx.save_for_backward(grad)
        ctx.mark_non_differentiable(sc[1])
        return sc[0], sc[1]

    @staticmethod
    def backward(ctx, g, _gk):
        (grad,) = ctx.saved_tensors
        return grad * g, None, None, None


class _AULossRowsFn(torch.autograd.Function):
    """AULoss on slots 0..11 of the model's [B, width] output rows, the gradient returned in that layout by the loss kernel itself
    (avf_au_loss_wide): ``loss(out[:, :12], y)`` costs autograd a fill and a copy for the slice on top of the loss's own launches"""

    @staticmethod
    def forward(ctx, out, y_true, pos_weight, ignore, sum_mode):
        if y_true.stride(-1) != 1 or y_true.dtype != torch.float32:
            y_true = y_true.to(torch.float32).contiguous()
        res, grad = ops.au_loss_wide(out, y_true, pos_weight, ignore, sum_mode)
        ctx.save_for_backward(grad)
        if sum_mode:
            ctx.mark_non_differentiable(res[1])
            return res[0], res[1]
        return res

    @staticmethod
    def backward(ctx, g, *_):
        (grad,) = ctx.saved_tensors
        return grad * g, None, None, None, None


class AULoss(nn.Module):
    """Rows whose FIRST label equals ``ignore`` are dropped (loss.py:85-88); the loss is the mean of
    ``BCEWithLogits(reduction='none', pos_weight=[1,1,1,1,1,1,1,3,3,3,1,2])`` over kept rows x 12.
    With every row dropped the result is NaN, exactly like the reference's mean over an empty tensor.
    Unlike the reference ctor (loss.py:73) this does not need a current CUDA device at construction:
    ``pos_weight`` is a buffer and follows ``.to(device)``."""

    def __init__(self, ignore=-1):
        super().__init__()
        self.ignore = ignore
        self.register_buffer("pos_weight", torch.tensor(AU_POS_WEIGHT, dtype=torch.float32), persistent=False)
        # set by dp.DataParallel: callable (local_sum, local_kept) -> global mean whose backward, AVERAGED over the ranks,
        # is the gradient of that global mean (ranks may hold different numbers of ignored rows).  It is a COLLECTIVE: it
        # is taken only for a training-mode loss with gradients enabled - the one call every rank makes once per step -
        # or when ``reduce_eval`` is set (then EVERY rank must call the loss the same number of times); a validation
        # loss on one rank, or on ranks with different batch counts, stays local and cannot deadlock.
        self.global_mean = None
        self.reduce_eval = False

    def forward(self, y_pred, y_true):
        if not y_pred.is_cuda:
            raise RuntimeError("AULoss (HIP) needs its inputs on the MI355X; there is no CPU fallback")
        pw = self.pos_weight if self.pos_weight.device == y_pred.device else self.pos_weight.to(y_pred.device)
        if self.global_mean is not None and ((self.training and torch.is_grad_enabled()) or self.reduce_eval):
            s, k = _AULossSumFn.apply(y_pred, y_true, pw, float(self.ignore))
            return self.global_mean(s, k)
        return _AULossFn.apply(y_pred, y_true, pw, float(self.ignore))

    def forward_rows(self, out, y_true):
        """``self(out[:, :y_true.shape[1]], y_true)`` for a contiguous fp32 [B, width] output of the model, without the slice:
        same value, same gradient (zero in the other slots), two launches fewer in backward (the task models' get_au_loss)"""
        if not (out.is_cuda and out.dim() == 2 and out.dtype == torch.float32 and out.is_contiguous() and y_true.dim() == 2
                and out.shape[0] == y_true.shape[0] and out.shape[1] >= y_true.shape[1]):
            return self(out[:, :y_true.shape[1]], y_true)
        pw = self.pos_weight if self.pos_weight.device == out.device else self.pos_weight.to(out.device)
        if self.global_mean is not None and ((self.training and torch.is_grad_enabled()) or self.reduce_eval):
            s, k = _AULossRowsFn.apply(out, y_true, pw, float(self.ignore), True)
            return self.global_mean(s, k)
        return _AULossRowsFn.apply(out, y_true, pw, float(self.ignore), False)
