"""Dice metric (reference/train.py:92,132,147,151-152) and the legacy confusion-count helper
(reference/utils/utils.py:11-44) on one fused HIP counting kernel: tp / fp / fn / tn per sample as
64-bit integers in a single pass over prediction and label.
"""
from __future__ import annotations

import torch

from . import _lib


def confusion_counts(y_pred: torch.Tensor, y: torch.Tensor, *, binary: bool = True, threshold: float = 0.5,
                     fold_label_two: bool = False) -> torch.Tensor:
    """[B,4] int64 counts (tp, fp, fn, tn); the positive class is value 1.

    ``binary=True``: ``y_pred`` holds {0,1}; otherwise ``y_pred >= threshold`` is the prediction (pass logits
    with ``threshold=0`` to fuse sigmoid + AsDiscrete(0.5) of reference/train.py:89,145).
    """
    if not y_pred.is_cuda:
        raise RuntimeError("unet_amd metrics run on an AMD GPU only (no CPU or eager fallback)")
    lib = _lib.load()
    p = y_pred.as_subclass(torch.Tensor) if type(y_pred) is not torch.Tensor else y_pred
    t = y.as_subclass(torch.Tensor) if type(y) is not torch.Tensor else y
    p = p.contiguous()
    if p.dtype != torch.float32:
        p = p.float()
    t = t.contiguous()
    if _lib.label_dtype_code(t) < 0:
        t = t.float()
    if p.numel() != t.numel():
        raise ValueError(f"y_pred and y should have the same number of elements, got {tuple(p.shape)} and {tuple(t.shape)}")
    nb = p.shape[0]
    npix = p[0].numel()
    counts = torch.empty(nb, 4, dtype=torch.int64, device=p.device)
    _lib.check(lib.mi_dice_counts(p.data_ptr(), 1 if binary else 0, float(threshold), t.data_ptr(),
                                  _lib.label_dtype_code(t), 1 if fold_label_two else 0, nb, npix, counts.data_ptr(),
                                  _lib.current_stream(p.device)), "mi_dice_counts")
    return counts


class DiceMetric:
    """``monai.metrics.DiceMetric`` for binarised one-channel data.

    Per sample ``2|P∧G| / (|P|+|G|)``, NaN when the ground truth is empty (``ignore_empty=True``);
    ``aggregate()`` is the NaN-skipping mean over everything seen since ``reset()``.
    ``include_background`` is accepted for signature compatibility (one channel: class 1 is scored).
    """

    def __init__(self, include_background: bool = True, reduction: str = "mean", get_not_nans: bool = False,
                 ignore_empty: bool = True, num_classes=None, return_with_label=False):
        if reduction != "mean":
            raise NotImplementedError("unet_amd.DiceMetric implements reduction='mean'")
        self.include_background, self.reduction = include_background, reduction
        self.get_not_nans, self.ignore_empty = get_not_nans, ignore_empty
        self._buffers = []

    def __call__(self, y_pred, y):
        if isinstance(y_pred, (list, tuple)):
            y_pred = torch.stack(list(y_pred), dim=0)
        if isinstance(y, (list, tuple)):
            y = torch.stack(list(y), dim=0)
        if y_pred.shape[1] != 1:
            raise NotImplementedError("unet_amd.DiceMetric implements one-channel predictions")
        c = confusion_counts(y_pred, y, binary=True).to(torch.float32)
        tp, fp, fn = c[:, 0], c[:, 1], c[:, 2]
        g = tp + fn
        pr = tp + fp
        dice = 2.0 * tp / (g + pr)
        if self.ignore_empty:
            dice = torch.where(g > 0, dice, torch.full_like(dice, float("nan")))
        else:
            dice = torch.where(g > 0, dice, torch.where(pr > 0, torch.zeros_like(dice), torch.ones_like(dice)))
        dice = dice.reshape(-1, 1)
        self._buffers.append(dice)
        return dice

    def get_buffer(self):
        return torch.cat(self._buffers, dim=0) if self._buffers else None

    def aggregate(self, reduction=None):
        if not self._buffers:
            raise ValueError("the data to aggregate must be PyTorch Tensor.")
        f = torch.cat(self._buffers, dim=0)
        valid = ~torch.isnan(f)
        n = valid.sum()
        total = torch.where(valid, f, torch.zeros_like(f)).sum()
        out = torch.where(n > 0, total / n.clamp(min=1), torch.zeros_like(total))
        return (out, n.float()) if self.get_not_nans else out

    def reset(self):
        self._buffers = []


def calculate_dice_score(y_pred: torch.Tensor, y: torch.Tensor):
    """(dice, sensitivity, specificity) of reference/utils/utils.py:11-44 on flattened binary tensors."""
    c = confusion_counts(y_pred.reshape(1, -1), y.reshape(1, -1), binary=True).to(torch.float32)[0]
    tp, fp, fn, tn = c[0], c[1], c[2], c[3]
    dice = 2 * tp / ((tp + fp) + (tp + fn))
    sensitivity = tp / (tp + fn)
    specificity = tn / (tn + fp)
    return dice, sensitivity, specificity
