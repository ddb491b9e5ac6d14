"""Oracle (test infrastructure): ``monai.inferers.sliding_window_inference`` restated.

Call site: reference/train.py:140 ``sliding_window_inference(val_inputs, (roi, roi), 4, model)`` with
MONAI defaults overlap=0.25, mode="constant", padding_mode="constant", cval=0.  Restated from MONAI 1.3
``monai/inferers/utils.py`` and ``monai/data/utils.py`` (``dense_patch_slices``).  Parity unpinned.
"""
from __future__ import annotations

import math
from typing import Callable, Sequence

import numpy as np
import torch
import torch.nn.functional as F


def _scan_interval(image_size, roi_size, overlap):
    out = []
    for i in range(len(roi_size)):
        if roi_size[i] == image_size[i]:
            out.append(int(roi_size[i]))
        else:
            interval = int(roi_size[i] * (1 - overlap))
            out.append(interval if interval > 0 else 1)
    return tuple(out)


def dense_patch_starts(image_size, patch_size, scan_interval):
    """Window start coordinates in MONAI's order (first spatial dim slowest)."""
    nd = len(image_size)
    starts = []
    for i in range(nd):
        if scan_interval[i] == 0:
            scan_num = 1
        else:
            num = int(math.ceil(float(image_size[i]) / scan_interval[i]))
            scan_dim = next((d for d in range(num) if d * scan_interval[i] + patch_size[i] >= image_size[i]), None)
            scan_num = scan_dim + 1 if scan_dim is not None else 1
        dim_starts = []
        for idx in range(scan_num):
            s = idx * scan_interval[i]
            s -= max(s + patch_size[i] - image_size[i], 0)
            dim_starts.append(s)
        starts.append(dim_starts)
    out = np.asarray([x.flatten() for x in np.meshgrid(*starts, indexing="ij")]).T
    return [tuple(int(v) for v in row) for row in out]


def sliding_window_inference(inputs: torch.Tensor, roi_size: Sequence[int] | int, sw_batch_size: int,
                             predictor: Callable[[torch.Tensor], torch.Tensor], overlap: float = 0.25,
                             mode: str = "constant", padding_mode: str = "constant", cval: float = 0.0):
    if mode != "constant":
        raise NotImplementedError("oracle restates constant blending only")
    nd = inputs.dim() - 2
    batch_size = inputs.shape[0]
    image_size_ = list(inputs.shape[2:])
    if isinstance(roi_size, int):
        roi_size = (roi_size,) * nd
    roi_size = tuple(int(r) if r and r > 0 else int(s) for r, s in zip(roi_size, image_size_))
    image_size = tuple(max(image_size_[i], roi_size[i]) for i in range(nd))
    pad_size = []
    for k in range(inputs.dim() - 1, 1, -1):
        diff = max(roi_size[k - 2] - inputs.shape[k], 0)
        half = diff // 2
        pad_size.extend([half, diff - half])
    if any(pad_size):
        inputs = F.pad(inputs, pad=pad_size, mode=padding_mode, value=cval)
    interval = _scan_interval(image_size, roi_size, overlap)
    starts = dense_patch_starts(image_size, roi_size, interval)
    num_win = len(starts)
    total = num_win * batch_size
    output, count = None, None
    for g in range(0, total, sw_batch_size):
        idxs = list(range(g, min(g + sw_batch_size, total)))
        wins = []
        for idx in idxs:
            b, w = idx // num_win, idx % num_win
            sl = (slice(b, b + 1), slice(None)) + tuple(slice(s, s + r) for s, r in zip(starts[w], roi_size))
            wins.append(inputs[sl])
        prob = predictor(torch.cat(wins, dim=0))
        if output is None:
            output = torch.zeros((batch_size, prob.shape[1]) + image_size, dtype=prob.dtype, device=prob.device)
            count = torch.zeros((1, 1) + image_size, dtype=prob.dtype, device=prob.device)
            # count map is shared over the batch: every sample sees the same windows
            seen = set()
        for j, idx in enumerate(idxs):
            b, w = idx // num_win, idx % num_win
            sl_sp = tuple(slice(s, s + r) for s, r in zip(starts[w], roi_size))
            output[(slice(b, b + 1), slice(None)) + sl_sp] += prob[j:j + 1]
            if w not in seen:
                count[(slice(None), slice(None)) + sl_sp] += 1.0
                seen.add(w)
    output = output / count
    if any(pad_size):
        crop = [slice(None), slice(None)]
        for sp in range(nd):
            lo = pad_size[(nd - 1 - sp) * 2]
            crop.append(slice(lo, lo + image_size_[sp]))
        output = output[tuple(crop)]
    return output
