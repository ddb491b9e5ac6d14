"""Oracle (test infrastructure): loss, metric and post-transform restatements.

Call sites followed: reference/train.py:88 (``DiceCELoss(to_onehot_y=False, sigmoid=True)``),
train.py:89,145 (``Compose([Activations(sigmoid=True), AsDiscrete(threshold=0.5)])``),
train.py:92,132,147,151-152 (``DiceMetric(include_background=False, reduction="mean")``),
train.py:141,145 (``decollate_batch``), train.py:24 (``set_determinism``) and the legacy helper
reference/utils/utils.py:11-44 (``calculate_dice_score``).  Algorithms restated from MONAI 1.3
(``monai/losses/dice.py``, ``monai/metrics/meandice.py``, ``monai/metrics/utils.py``,
``monai/transforms/post/array.py``).  Parity unpinned (see package docstring).
"""
from __future__ import annotations

import random

import numpy as np
import torch
import torch.nn as nn


class DiceLoss(nn.Module):
    """``monai.losses.DiceLoss`` restricted to the options the reference path uses."""

    def __init__(self, include_background=True, to_onehot_y=False, sigmoid=False, squared_pred=False,
                 jaccard=False, reduction="mean", smooth_nr=1e-5, smooth_dr=1e-5, batch=False):
        super().__init__()
        if to_onehot_y or squared_pred or jaccard or not include_background:
            raise NotImplementedError("oracle restates the options used at reference/train.py:88 only")
        self.sigmoid, self.reduction = sigmoid, reduction
        self.smooth_nr, self.smooth_dr, self.batch = float(smooth_nr), float(smooth_dr), batch

    def forward(self, input: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
        if self.sigmoid:
            input = torch.sigmoid(input)
        if target.shape != input.shape:
            raise AssertionError(f"ground truth has different shape ({target.shape}) from input ({input.shape})")
        reduce_axis = list(range(2, input.dim()))
        if self.batch:
            reduce_axis = [0] + reduce_axis
        intersection = torch.sum(target * input, dim=reduce_axis)
        ground_o = torch.sum(target, dim=reduce_axis)
        pred_o = torch.sum(input, dim=reduce_axis)
        denominator = ground_o + pred_o
        f = 1.0 - (2.0 * intersection + self.smooth_nr) / (denominator + self.smooth_dr)
        if self.reduction == "mean":
            return torch.mean(f)
        if self.reduction == "sum":
            return torch.sum(f)
        if self.reduction == "none":
            return f.view(list(f.shape) + [1] * (input.dim() - 2))
        raise ValueError(f"Unsupported reduction: {self.reduction}")


class DiceCELoss(nn.Module):
    """``monai.losses.DiceCELoss``; for one output channel the CE term is BCE-with-logits (MONAI >= 1.3)."""

    def __init__(self, include_background=True, to_onehot_y=False, sigmoid=False, softmax=False, reduction="mean",
                 smooth_nr=1e-5, smooth_dr=1e-5, batch=False, lambda_dice=1.0, lambda_ce=1.0):
        super().__init__()
        if softmax:
            raise NotImplementedError
        self.dice = DiceLoss(include_background=include_background, to_onehot_y=to_onehot_y, sigmoid=sigmoid,
                             reduction=reduction, smooth_nr=smooth_nr, smooth_dr=smooth_dr, batch=batch)
        self.cross_entropy = nn.CrossEntropyLoss(reduction=reduction)
        self.binary_cross_entropy = nn.BCEWithLogitsLoss(reduction=reduction)
        if lambda_dice < 0.0 or lambda_ce < 0.0:
            raise ValueError("lambda_dice and lambda_ce should be no less than 0.0.")
        self.lambda_dice, self.lambda_ce = lambda_dice, lambda_ce

    def forward(self, input: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
        if input.dim() != target.dim():
            raise ValueError("the number of dimensions for input and target should be the same, "
                             f"got shape {input.shape} (nb dims: {len(input.shape)}) and {target.shape} "
                             f"(nb dims: {len(target.shape)}).")
        if target.shape[1] != 1 and target.shape[1] != input.shape[1]:
            raise ValueError("number of channels for target is neither 1 nor the same as input")
        dice_loss = self.dice(input, target)
        if input.shape[1] != 1:
            n_pred_ch, n_target_ch = input.shape[1], target.shape[1]
            tgt = torch.squeeze(target, dim=1).long() if n_pred_ch != n_target_ch and n_target_ch == 1 else target
            ce_loss = self.cross_entropy(input, tgt)
        else:
            tgt = target if torch.is_floating_point(target) else target.to(dtype=input.dtype)
            ce_loss = self.binary_cross_entropy(input, tgt)
        return self.lambda_dice * dice_loss + self.lambda_ce * ce_loss


class Activations:
    def __init__(self, sigmoid=False):
        self.sigmoid = sigmoid

    def __call__(self, img):
        return torch.sigmoid(img) if self.sigmoid else img


class AsDiscrete:
    """``monai.transforms.AsDiscrete(threshold=t)``: ``img >= t`` cast back to the input dtype."""

    def __init__(self, threshold=None):
        self.threshold = threshold

    def __call__(self, img):
        if self.threshold is not None:
            return (img >= self.threshold).to(img.dtype)
        return img


class Compose:
    def __init__(self, transforms):
        self.transforms = list(transforms)

    def __call__(self, x):
        for t in self.transforms:
            x = t(x)
        return x


def decollate_batch(batch: torch.Tensor):
    """For a plain tensor: list of per-sample tensors (``monai.data.decollate_batch``)."""
    return [batch[i] for i in range(batch.shape[0])]


class DiceMetric:
    """``monai.metrics.DiceMetric(include_background=False, reduction="mean")`` for binarised inputs.

    One-channel data: class 1 is evaluated (``y_pred == 1`` vs ``y == 1``); a sample with an empty
    ground truth yields NaN (``ignore_empty=True``); ``aggregate`` is the NaN-skipping mean.
    """

    def __init__(self, include_background=True, reduction="mean", ignore_empty=True):
        self.include_background, self.reduction, self.ignore_empty = include_background, reduction, ignore_empty
        self._buffers = []

    @staticmethod
    def _channel(y_pred, y, ignore_empty):
        y_o = torch.sum(y)
        if y_o > 0:
            return (2.0 * torch.sum(torch.masked_select(y, y_pred))) / (y_o + torch.sum(y_pred))
        if ignore_empty:
            return torch.tensor(float("nan"), device=y_o.device)
        denorm = y_o + torch.sum(y_pred)
        return torch.tensor(1.0 if denorm <= 0 else 0.0, device=y_o.device)

    def _compute(self, y_pred, y):
        n_pred_ch = y_pred.shape[1]
        first_ch = 0 if self.include_background else 1
        data = []
        for b in range(y_pred.shape[0]):
            c_list = []
            for c in (range(first_ch, n_pred_ch) if n_pred_ch > 1 else [1]):
                x_pred = (y_pred[b, 0] == c) if y_pred.shape[1] == 1 else y_pred[b, c].bool()
                x = (y[b, 0] == c) if y.shape[1] == 1 else y[b, c]
                c_list.append(self._channel(x_pred, x.to(torch.float32) if x.dtype == torch.bool else x.float(),
                                            self.ignore_empty))
            data.append(torch.stack(c_list))
        return torch.stack(data, dim=0).contiguous()

    def __call__(self, y_pred, y):
        if isinstance(y_pred, (list, tuple)):
            y_pred = torch.stack(list(y_pred), dim=0)
        if isinstance(y, (list, tuple)):
            y = torch.stack(list(y), dim=0)
        ret = self._compute(y_pred, y)
        self._buffers.append(ret)
        return ret

    def aggregate(self):
        f = torch.cat(self._buffers, dim=0).clone().float()
        nans = torch.isnan(f)
        not_nans = (~nans).float()
        f[nans] = 0
        if self.reduction == "mean":
            nn_c = not_nans.sum(dim=1)
            f = torch.where(nn_c > 0, f.sum(dim=1) / nn_c, torch.zeros(1, dtype=torch.float))
            nn_b = (nn_c > 0).float().sum(dim=0)
            return torch.where(nn_b > 0, f.sum(dim=0) / nn_b, torch.zeros(1, dtype=torch.float)).squeeze()
        raise NotImplementedError

    def reset(self):
        self._buffers = []


def calculate_dice_score(y_pred, y):
    """Restatement of reference/utils/utils.py:11-44 (dice, sensitivity, specificity on binary tensors)."""
    predict = y_pred.contiguous().view(1, -1)
    target = y.contiguous().view(1, -1)
    tp = torch.sum(torch.mul(predict, target))
    fn = torch.sum(torch.mul(predict != 1, target))
    fp = torch.sum(torch.mul(predict, target != 1))
    tn = torch.sum(torch.mul(predict != 1, target != 1))
    dice = 2 * tp / (torch.sum(predict) + torch.sum(target))
    sensitivity = tp / (tp + fn)
    specificity = tn / (tn + fp)
    return dice, sensitivity, specificity


def set_determinism(seed=None):
    """``monai.utils.set_determinism`` as used at reference/train.py:24."""
    if seed is None:
        return
    seed = int(seed) % (2**32)
    torch.manual_seed(seed)
    random.seed(seed)
    np.random.seed(seed)
    torch.backends.cudnn.deterministic = True
    torch.backends.cudnn.benchmark = False
