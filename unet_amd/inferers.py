"""``sliding_window_inference`` with MONAI's call surface (reference/train.py:140): window placement is
host logic, the crop and the overlap-average are HIP kernels, the predictor is any callable (the
``unet_amd.UNet``).  Constant blending, constant padding; one-channel fp32 images, 2-D or 3-D.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Callable, Sequence

import torch

from . import _lib


def _scan_interval(image_size, roi_size, overlap):
    out = []
    for i in range(len(roi_size)):
        if roi_size[i] == image_size[i]:
            out.append(int(roi_size[i]))
        else:
            interval = int(roi_size[i] * (1 - overlap))
            out.append(interval if interval > 0 else 1)
    return tuple(out)


def window_starts(image_size: Sequence[int], roi_size: Sequence[int], overlap: float = 0.25):
    """Window origins in MONAI's order (first spatial dim slowest); the last window of a dim is pulled
    back so that it ends at the image border."""
    interval = _scan_interval(image_size, roi_size, overlap)
    per_dim = []
    for i in range(len(image_size)):
        num = int(math.ceil(float(image_size[i]) / interval[i]))
        scan_dim = next((d for d in range(num) if d * interval[i] + roi_size[i] >= image_size[i]), None)
        scan_num = scan_dim + 1 if scan_dim is not None else 1
        starts = []
        for idx in range(scan_num):
            s = idx * interval[i]
            s -= max(s + roi_size[i] - image_size[i], 0)
            starts.append(s)
        per_dim.append(starts)
    out = [()]
    for starts in per_dim:
        out = [o + (s,) for o in out for s in starts]
    return out


def shard_range(n_items: int, rank: int, world: int):
    """Contiguous chunk [lo, hi) of ``n_items`` independent units (slices, windows) owned by ``rank``."""
    per = (n_items + world - 1) // world
    lo = min(n_items, rank * per)
    return lo, min(n_items, lo + per)


SW_GROUP_WINDOWS = 288   # windows per model call when the predictor is known to treat samples independently


def _windows_per_call(predictor, sw_batch_size: int, total: int, window_elems: int):
    """(windows per model call, pad the last call to that size).

    MONAI's ``sw_batch_size`` only bounds memory.  For this package's U-Net in eval mode every window is an
    independent sample (instance norm, dropout off), so a larger group gives the same numbers in fewer, fuller
    launches.  The group size is then a power of two (or the maximum), and a short last group is padded with
    zero windows, whose outputs are dropped, up to the next power of two (not to the full group: a tail of one
    window costs one window): whatever the image size, the U-Net sees only the batch sizes 1, 2, 4 ... and the
    maximum, and builds at most that many engine plans (``UNet.MAX_PLANS`` covers the set).  Any other predictor gets exactly the group size it was asked
    for and an unpadded tail (``UNET_AMD_SW_EXACT=1`` forces that for the U-Net too)."""
    import os

    from .nets import UNet

    n = max(1, int(sw_batch_size))
    if isinstance(predictor, UNet) and not predictor.training and os.environ.get("UNET_AMD_SW_EXACT", "0") != "1":
        cap = max(1, (1 << 28) // max(1, window_elems))       # keep one group's input below 1 GiB of fp32
        top = max(n, min(SW_GROUP_WINDOWS, cap))
        g = 1
        while g < total and g < top:
            g *= 2
        return min(g, top), True
    return n, False


_STARTS_CACHE: dict = {}


def _starts_on_device(st3, dev):
    """Window origins as an int32 tensor on ``dev``, kept per (origins, device): a validation loop asks for the same
    few placements again and again, and a pageable host-to-device copy per call would stall the host behind the GPU."""
    key = (st3, dev.type, dev.index)
    t = _STARTS_CACHE.get(key)
    if t is None:
        if len(_STARTS_CACHE) >= 64:
            _STARTS_CACHE.clear()
        t = torch.tensor(st3, dtype=torch.int32).to(dev)
        _STARTS_CACHE[key] = t
    return t


def sliding_window_inference(inputs: torch.Tensor, roi_size, sw_batch_size: int,
                             predictor: Callable[[torch.Tensor], torch.Tensor], overlap: float = 0.25,
                             mode: str = "constant", sigma_scale=0.125, padding_mode: str = "constant",
                             cval: float = 0.0, sw_device=None, device=None, progress=False, **kwargs):
    if str(mode).lower() != "constant" or str(padding_mode).lower() != "constant":
        raise NotImplementedError("unet_amd.sliding_window_inference implements constant blending and padding")
    if not inputs.is_cuda:
        raise RuntimeError("unet_amd.sliding_window_inference runs on an AMD GPU only (no CPU or eager fallback)")
    if inputs.shape[1] != 1:
        raise NotImplementedError("one-channel inputs only")
    lib = _lib.load()
    x = inputs.as_subclass(torch.Tensor) if type(inputs) is not torch.Tensor else inputs
    x = x.contiguous()
    if x.dtype != torch.float32:
        x = x.float()
    nd = x.dim() - 2
    nb = x.shape[0]
    img = [int(s) for s in x.shape[2:]]
    if isinstance(roi_size, int):
        roi_size = (roi_size,) * nd
    roi = [int(r) if r and r > 0 else img[i] for i, r in enumerate(roi_size)]
    padded = [max(img[i], roi[i]) for i in range(nd)]
    pad_lo = [max(roi[i] - img[i], 0) // 2 for i in range(nd)]
    starts = window_starts(padded, roi, overlap)
    nwin = len(starts)
    total = nwin * nb
    dev = x.device
    st3 = tuple((0,) * (3 - nd) + tuple(s) for s in starts)
    starts_dev = _starts_on_device(st3, dev)
    I3 = C.c_int32 * 3
    img3 = I3(*([1] * (3 - nd) + img))
    roi3 = I3(*([1] * (3 - nd) + roi))
    pad3 = I3(*([0] * (3 - nd) + pad_lo))
    rpix = 1
    for r in roi:
        rpix *= r
    win_out = None
    stream = _lib.current_stream(dev)
    group, pad_tail = _windows_per_call(predictor, sw_batch_size, total, rpix)
    for g in range(0, total, group):
        n = min(group, total - g)
        if n < group and pad_tail and n & (n - 1):
            wins = torch.zeros((min(group, 1 << n.bit_length()), 1) + tuple(roi), dtype=torch.float32, device=dev)
        else:
            wins = torch.empty((n, 1) + tuple(roi), dtype=torch.float32, device=dev)
        _lib.check(lib.mi_sw_gather(x.data_ptr(), wins.data_ptr(), starts_dev.data_ptr(), nwin, g, n, nb, img3, roi3,
                                    pad3, float(cval), stream), "mi_sw_gather")
        pred = predictor(wins)
        if isinstance(pred, (tuple, list)):
            pred = pred[0]
        if pred.shape != wins.shape:
            raise NotImplementedError("the predictor must keep the window shape (one output channel)")
        if n == total and pred.dtype == torch.float32 and pred.is_contiguous():
            win_out = pred                  # one group holds every window: blend straight from the predictor's output
        else:
            if win_out is None:
                win_out = torch.empty((total, 1) + tuple(roi), dtype=torch.float32, device=dev)
            win_out[g:g + n].copy_(pred[:n])
    out = torch.empty((nb, 1) + tuple(img), dtype=torch.float32, device=dev)
    _lib.check(lib.mi_sw_blend(win_out.data_ptr(), out.data_ptr(), starts_dev.data_ptr(), nwin, nb, img3, roi3, pad3,
                               stream), "mi_sw_blend")
    return out
