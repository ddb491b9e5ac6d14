"""Full-volume evaluation: sliding-window inference over the slices of a set of volumes, hard Dice per
slice, mean Dice per volume (reference/train.py:128-153 for the loop, reference/utils/utils.py:190-224
for the per-volume table: the slices of one patient, named ``{organ}_{case}_{slice}``
(reference/create_2d_slices.py:35), are averaged).

Slices are independent, so they are sharded over the ranks (one process per GPU); the only exchange is
one all-reduce of the per-volume (sum of slice Dice, number of scored slices) pairs.
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist

from .inferers import shard_range, sliding_window_inference
from .metrics import confusion_counts
from .parallel import all_reduce_dice_partials
from .utils import volume_id


class VolumeEvaluator:
    """``evaluate(slices, labels, names)`` -> ``(per_volume: {volume id: mean Dice}, mean_dice)``.

    ``slices`` / ``labels``: ``[S, 1, H, W]`` tensors on the GPU (labels may hold 2, folded to 1 like
    reference/train.py:137-138); ``names``: one ``{organ}_{case}_{slice}`` string per slice.  Every rank passes the
    same full list and scores its own contiguous shard.  ``mean_dice`` is the NaN-skipping mean over all
    slices, i.e. what ``DiceMetric.aggregate()`` returns in the reference's validation loop.
    """

    def __init__(self, model: Callable[[torch.Tensor], torch.Tensor], roi_size, sw_batch_size: int = 4,
                 overlap: float = 0.25, slices_per_call: int = 8, group=None):
        self.model, self.roi_size, self.sw_batch_size, self.overlap = model, roi_size, int(sw_batch_size), overlap
        self.slices_per_call = max(1, int(slices_per_call))
        self.group = group

    def _world(self) -> Tuple[int, int]:
        if dist.is_available() and dist.is_initialized():
            return dist.get_rank(self.group), dist.get_world_size(self.group)
        return 0, 1

    @torch.no_grad()
    def slice_dice(self, slices: torch.Tensor, labels: torch.Tensor) -> torch.Tensor:
        """Hard Dice of every slice ([S] fp32, NaN where the ground truth is empty)."""
        out = []
        for s in range(0, slices.shape[0], self.slices_per_call):
            x = slices[s:s + self.slices_per_call]
            logits = sliding_window_inference(x, self.roi_size, self.sw_batch_size, self.model, overlap=self.overlap)
            # sigmoid(z) >= 0.5  <=>  z >= 0: activation + threshold fused into the counting kernel
            c = confusion_counts(logits, labels[s:s + self.slices_per_call], binary=False, threshold=0.0,
                                 fold_label_two=True).to(torch.float32)
            tp, fp, fn = c[:, 0], c[:, 1], c[:, 2]
            g = tp + fn
            d = 2.0 * tp / (g + tp + fp)
            out.append(torch.where(g > 0, d, torch.full_like(d, float("nan"))))
        if not out:
            return torch.empty(0, dtype=torch.float32, device=slices.device)
        return torch.cat(out)

    @torch.no_grad()
    def evaluate(self, slices: torch.Tensor, labels: torch.Tensor, names: Sequence[str]):
        if slices.shape[0] != len(names) or labels.shape[0] != len(names):
            raise ValueError("slices, labels and names must describe the same number of slices")
        was_training = getattr(self.model, "training", False)
        if hasattr(self.model, "eval"):
            self.model.eval()
        rank, world = self._world()
        vols: List[str] = sorted({volume_id(n) for n in names})
        index = {v: i for i, v in enumerate(vols)}
        lo, hi = shard_range(len(names), rank, world)
        dice = self.slice_dice(slices[lo:hi], labels[lo:hi])
        part = torch.zeros(len(vols) + 1, 2, dtype=torch.float64, device=slices.device)
        if hi > lo:
            vid = torch.tensor([index[volume_id(n)] for n in names[lo:hi]], dtype=torch.int64, device=slices.device)
            ok = ~torch.isnan(dice)
            d64 = torch.where(ok, dice, torch.zeros_like(dice)).double()
            part[:, 0].index_add_(0, vid, d64)
            part[:, 1].index_add_(0, vid, ok.double())
            part[len(vols), 0] = d64.sum()
            part[len(vols), 1] = ok.double().sum()
        all_reduce_dice_partials(part, self.group)
        host = part.cpu()
        table: Dict[str, float] = {}
        for v, i in index.items():
            if host[i, 1] > 0:
                table[v] = float(host[i, 0] / host[i, 1])
        mean = float(host[len(vols), 0] / host[len(vols), 1]) if host[len(vols), 1] > 0 else 0.0
        if was_training and hasattr(self.model, "train"):
            self.model.train()
        return table, mean
