"""``DiceLoss`` / ``DiceCELoss`` with MONAI's call surface (reference/train.py:88,116,142), computed by
single-pass HIP reductions: per-sample sums of sigmoid(z)*t, sigmoid(z), t and the BCE-with-logits
terms in one kernel, the gradient w.r.t. the logits in another.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import _lib


class _DiceCEFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target, smooth_nr, smooth_dr, lambda_dice, lambda_ce, fold2):
        lib = _lib.load()
        nb = logits.shape[0]
        npix = logits[0].numel()
        dev = logits.device
        sums = torch.empty(nb * 4, dtype=torch.float64, device=dev)
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        st = _lib.current_stream(dev)
        _lib.check(lib.mi_dice_ce_loss_fwd(logits.data_ptr(), target.data_ptr(), _lib.label_dtype_code(target),
                                           1 if fold2 else 0, nb, npix, smooth_nr, smooth_dr, lambda_dice, lambda_ce,
                                           sums.data_ptr(), loss.data_ptr(), 0, st), "mi_dice_ce_loss_fwd")
        ctx.save_for_backward(logits, target, sums)
        ctx.cfg = (smooth_nr, smooth_dr, lambda_dice, lambda_ce, fold2)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, gout):
        logits, target, sums = ctx.saved_tensors
        smooth_nr, smooth_dr, lambda_dice, lambda_ce, fold2 = ctx.cfg
        lib = _lib.load()
        nb = logits.shape[0]
        npix = logits[0].numel()
        dz = torch.empty_like(logits)
        g = gout.contiguous().float()
        _lib.check(lib.mi_dice_ce_loss_bwd(logits.data_ptr(), target.data_ptr(), _lib.label_dtype_code(target),
                                           1 if fold2 else 0, nb, npix, smooth_nr, smooth_dr, lambda_dice, lambda_ce,
                                           sums.data_ptr(), g.data_ptr(), 1.0, dz.data_ptr(),
                                           _lib.current_stream(logits.device)), "mi_dice_ce_loss_bwd")
        return dz, None, None, None, None, None, None


def _prepare(input: torch.Tensor, target: torch.Tensor):
    if input.dim() != target.dim():
        raise ValueError("the number of dimensions for input and target should be the same, "
                         f"got shape {input.shape} (nb dims: {len(input.shape)}) and {target.shape} "
                         f"(nb dims: {len(target.shape)}).")
    if target.shape != input.shape:
        raise AssertionError(f"ground truth has different shape ({target.shape}) from input ({input.shape})")
    if input.shape[1] != 1:
        raise NotImplementedError("unet_amd losses implement the one-channel sigmoid case (reference/train.py:88)")
    if not input.is_cuda:
        raise RuntimeError("unet_amd losses run on an AMD GPU only (no CPU or eager fallback)")
    x = input.as_subclass(torch.Tensor) if type(input) is not torch.Tensor else input
    t = target.as_subclass(torch.Tensor) if type(target) is not torch.Tensor else target
    x = x.contiguous()
    if x.dtype != torch.float32:
        x = x.float()
    t = t.contiguous()
    if _lib.label_dtype_code(t) < 0:
        t = t.float()
    return x, t


class DiceCELoss(nn.Module):
    """``monai.losses.DiceCELoss`` for one-channel logits: ``lambda_dice * Dice(sigmoid(x), t) +
    lambda_ce * BCEWithLogits(x, t)`` (MONAI >= 1.3 semantics).  ``lambda_ce=0`` gives the plain Dice loss.

    ``fold_label_two=True`` treats label value 2 as 1 inside the kernels, i.e. fuses
    ``labels[labels == 2] = 1`` (reference/train.py:112-113) into the loss read.
    """

    def __init__(self, include_background=True, to_onehot_y=False, sigmoid=False, softmax=False, other_act=None,
                 squared_pred=False, jaccard=False, reduction="mean", smooth_nr=1e-5, smooth_dr=1e-5, batch=False,
                 weight=None, lambda_dice=1.0, lambda_ce=1.0, fold_label_two=False):
        super().__init__()
        bad = []
        if not sigmoid: bad.append("sigmoid=True is required")
        if softmax or other_act is not None: bad.append("softmax / other_act")
        if to_onehot_y or squared_pred or jaccard or batch or weight is not None: bad.append("to_onehot_y / squared_pred / jaccard / batch / weight")
        if not include_background: bad.append("include_background=False")
        if reduction != "mean": bad.append("reduction must be 'mean'")
        if bad:
            raise NotImplementedError("unet_amd.DiceCELoss: unsupported options: " + "; ".join(bad))
        if lambda_dice < 0.0 or lambda_ce < 0.0:
            raise ValueError("lambda_dice and lambda_ce should be no less than 0.0.")
        self.smooth_nr, self.smooth_dr = float(smooth_nr), float(smooth_dr)
        self.lambda_dice, self.lambda_ce = float(lambda_dice), float(lambda_ce)
        self.fold_label_two = bool(fold_label_two)

    def forward(self, input: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
        x, t = _prepare(input, target)
        return _DiceCEFunction.apply(x, t, self.smooth_nr, self.smooth_dr, self.lambda_dice, self.lambda_ce,
                                     self.fold_label_two)


class DiceLoss(DiceCELoss):
    """``monai.losses.DiceLoss(sigmoid=True)``: the Dice term alone."""

    def __init__(self, include_background=True, to_onehot_y=False, sigmoid=False, softmax=False, other_act=None,
                 squared_pred=False, jaccard=False, reduction="mean", smooth_nr=1e-5, smooth_dr=1e-5, batch=False,
                 fold_label_two=False):
        super().__init__(include_background=include_background, to_onehot_y=to_onehot_y, sigmoid=sigmoid,
                         softmax=softmax, other_act=other_act, squared_pred=squared_pred, jaccard=jaccard,
                         reduction=reduction, smooth_nr=smooth_nr, smooth_dr=smooth_dr, batch=batch,
                         lambda_dice=1.0, lambda_ce=0.0, fold_label_two=fold_label_two)
