"""Small host-side helpers with the reference's names: seeding (reference/train.py:24), checkpoint
file naming (reference/utils/utils.py:172-187, train.py:157) and the per-volume Dice grouping
(reference/utils/utils.py:190-224 with slice names from reference/create_2d_slices.py:35).
"""
from __future__ import annotations

import os
import random
import re
from collections import defaultdict
from typing import Dict, Iterable, Tuple

import numpy as np
import torch


def set_determinism(seed=None, model=None) -> None:
    """Seeds python / numpy / torch like ``monai.utils.set_determinism``; ``model`` (a ``unet_amd.UNet``)
    also gets its stateless dropout stream re-seeded."""
    if seed is None:
        return
    seed = int(seed) % (2 ** 32)
    torch.manual_seed(seed)
    random.seed(seed)
    np.random.seed(seed)
    torch.backends.cudnn.deterministic = True
    torch.backends.cudnn.benchmark = False
    if model is not None:
        model.seed_dropout(seed)


def checkpoint_name(organ: str, train_data: str = "real", img_format: str = "2d", sam_prompt: str = "box",
                    n_samples=None) -> str:
    """Legacy file name ``unet_{organ}_{traindata}_{2D|3D}.pth`` (reference/utils/utils.py:172-187)."""
    organ_key = organ.split("_")[1].lower() if "_" in organ else organ.lower()
    task = organ.split("_")[0].lower() if "_" in organ else "task"
    tag = f"{sam_prompt}_pseudolabels" if train_data != "real" else "real"
    n = f"_{n_samples}" if n_samples else ""
    return f"unet_{task}_{organ_key}_{tag}{n}_{img_format.upper()}.pth"


def save_checkpoint(model: torch.nn.Module, path: str) -> str:
    """``torch.save(model.state_dict(), path)`` with CPU fp32 tensors in MONAI's key layout (train.py:157)."""
    sd = {k: v.detach().to("cpu", torch.float32).clone() for k, v in model.state_dict().items()}
    d = os.path.dirname(path)
    if d:
        os.makedirs(d, exist_ok=True)
    torch.save(sd, path)
    return path


_SLICE_RE = re.compile(r"^(?P<organ>[A-Za-z]+)_(?P<case>\d+)_(?P<idx>\d+)")


def volume_id(slice_name: str) -> str:
    """``{organ}_{case}`` of a slice file named ``{organ}_{case}_{slice}.nii.gz`` (create_2d_slices.py:35)."""
    m = _SLICE_RE.match(os.path.basename(slice_name))
    if not m:
        raise ValueError(f"not a slice name: {slice_name}")
    return f"{m.group('organ')}_{m.group('case')}"


def per_volume_dice(slice_names: Iterable[str], slice_dices: Iterable[float]) -> Dict[str, float]:
    """Per-volume Dice = mean of the slice Dices of one patient (reference/utils/utils.py:196,216-224)."""
    acc: Dict[str, Tuple[float, int]] = defaultdict(lambda: (0.0, 0))
    for n, d in zip(slice_names, slice_dices):
        if d != d:  # NaN: empty ground truth, skipped like DiceMetric.aggregate
            continue
        s, c = acc[volume_id(n)]
        acc[volume_id(n)] = (s + float(d), c + 1)
    return {k: s / c for k, (s, c) in acc.items() if c}
