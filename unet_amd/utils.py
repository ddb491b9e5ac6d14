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


TRAIN_DATA_LIST = ("all", "n_random", "n_worst")     # reference/config.py:15


def checkpoint_name(organ: str, train_data: str = "all", img_format: str = "2d", use_pseudo_labels: bool = False,
                    sam_prompt: str = "box", n_train_samples: int = 1000) -> str:
    """File name of ``saveCheckpoint`` (reference/utils/utils.py:172-187), from the legacy flags of
    reference/config.py: ``ORGAN`` (:12), ``TRAIN_DATA`` (:15-16), ``N_TRAIN_SAMPLES`` (:17), ``USE_PSEUDO_LABELS``
    (:22), ``SAM_PROMPT`` (:26), ``IMG_FORMAT`` (:5).

    2-D: ``unet_{organ.lower()}_{tag}_2D.pth`` with tag ``{N}_{random|worst}`` when ``train_data`` is a subset
    (the pseudo-label flag is not looked at in that branch), else ``{prompt}_pseudolabels`` or ``all``.
    3-D: ``unet_{organ.lower()}_3D.pth``."""
    if img_format == "3d":
        return f"unet_{organ.lower()}_3D.pth"
    if img_format != "2d":
        raise ValueError(f"img_format must be '2d' or '3d', got {img_format!r}")
    if train_data != "all":
        tag = f"{n_train_samples}_{train_data.split('_')[1]}"
    elif use_pseudo_labels:
        tag = f"{sam_prompt}_pseudolabels"
    else:
        tag = f"{train_data}"
    return f"unet_{organ.lower()}_{tag}_2D.pth"


def unet_config_from_state_dict(state) -> dict:
    """Constructor arguments (spatial_dims, channels, strides, num_res_units) read off the tensor shapes of a
    MONAI-layout ``state_dict``: the reference trains 32-512 channels (reference/train.py:82), the benchmark
    configurations 16-256, and either file must load without the caller knowing which it is."""
    channels, prefix = [], "model"
    while f"{prefix}.0.conv.unit0.conv.weight" in state:
        channels.append(int(state[f"{prefix}.0.conv.unit0.conv.weight"].shape[0]))
        prefix += ".1.submodule"
    bottom = f"{prefix}.conv.unit0.conv.weight"
    if not channels or bottom not in state:
        raise ValueError("not a residual U-Net state_dict in MONAI's key layout")
    channels.append(int(state[bottom].shape[0]))
    units = 0
    while f"model.0.conv.unit{units}.conv.weight" in state:
        units += 1
    return dict(spatial_dims=state[bottom].dim() - 2, channels=tuple(channels), strides=(2,) * (len(channels) - 1),
                num_res_units=units)


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
