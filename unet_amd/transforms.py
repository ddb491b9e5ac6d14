"""Post-processing surface used at reference/train.py:89,141,145: ``Compose([Activations(sigmoid=True),
AsDiscrete(threshold=0.5)])`` and ``decollate_batch``.  These are per-sample host-side glue over
single element-wise tensor expressions; the fused GPU form (threshold the logits at 0 inside the
counting kernel) is ``unet_amd.metrics.confusion_counts(logits, y, binary=False, threshold=0.0)``.
"""
from __future__ import annotations

import torch


class Activations:
    def __init__(self, sigmoid: bool = False, softmax: bool = False, other=None):
        if softmax or other is not None:
            raise NotImplementedError("unet_amd.Activations implements sigmoid only")
        self.sigmoid = sigmoid

    def __call__(self, img: torch.Tensor) -> torch.Tensor:
        return torch.sigmoid(img) if self.sigmoid else img


class AsDiscrete:
    """``threshold=t``: ``img >= t`` in the input dtype (MONAI semantics)."""

    def __init__(self, argmax: bool = False, to_onehot=None, threshold=None, rounding=None):
        if argmax or to_onehot is not None or rounding is not None:
            raise NotImplementedError("unet_amd.AsDiscrete implements thresholding only")
        self.threshold = threshold

    def __call__(self, img: torch.Tensor) -> torch.Tensor:
        if self.threshold is None:
            return img
        return (img >= self.threshold).to(img.dtype)


class Compose:
    def __init__(self, transforms=None):
        self.transforms = list(transforms or [])

    def __call__(self, x):
        for t in self.transforms:
            x = t(x)
        return x


def decollate_batch(batch):
    """A batched tensor becomes the list of its samples (``monai.data.decollate_batch`` on a tensor)."""
    if isinstance(batch, torch.Tensor):
        return [batch[i] for i in range(batch.shape[0])]
    raise NotImplementedError("unet_amd.decollate_batch handles plain tensors")
