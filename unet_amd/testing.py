"""The legacy test run (reference/README.md:22-28: ``python3 unetTest.py --checkpoint_name ...``; the script
itself is absent from the reference tree): load a ``.pth`` state dict, evaluate it over a directory of
2-D slice files named ``{organ}_{case}_{slice}.nii.gz`` (reference/create_2d_slices.py:35-40) and report
the Dice per volume and its average / maximum / minimum (reference/utils/utils.py:216-224).

Validation preprocessing is ``val_transforms_2d`` (reference/unetDataset.py:117-129): intensity window
[-175, 250] -> [0, 1] with clipping, nothing else.
"""
from __future__ import annotations

import argparse
import os
from typing import Dict, List, Sequence, Tuple

import numpy as np
import torch

from .evaluator import VolumeEvaluator
from .nets import UNet
from .nifti import read_nifti
from .utils import unet_config_from_state_dict, volume_id


def list_slices(images_dir: str, labels_dir: str) -> List[Tuple[str, str, str]]:
    """(name, image path, label path) for every slice present in both directories, ordered by volume then slice."""
    def stem(f):
        return f[:-7] if f.endswith(".nii.gz") else f[:-4] if f.endswith(".nii") else None
    imgs = {stem(f): os.path.join(images_dir, f) for f in os.listdir(images_dir) if stem(f)}
    lbls = {stem(f): os.path.join(labels_dir, f) for f in os.listdir(labels_dir) if stem(f)}
    names = sorted(set(imgs) & set(lbls), key=lambda n: (volume_id(n), int(n.rsplit("_", 1)[1])))
    return [(n, imgs[n], lbls[n]) for n in names]


def load_slices(items: Sequence[Tuple[str, str, str]], a_min=-175.0, a_max=250.0, b_min=0.0, b_max=1.0, device="cuda"):
    """Reads the files, applies the intensity window, returns (images [S,1,X,Y] fp32, labels [S,1,X,Y] fp32, names)."""
    xs, ys = [], []
    for _, ip, lp in items:
        x, _ = read_nifti(ip)
        y, _ = read_nifti(lp)
        x = np.squeeze(x).astype(np.float32)
        y = np.squeeze(y).astype(np.float32)
        if x.ndim != 2 or x.shape != y.shape:
            raise ValueError(f"{ip}: expected matching 2-D image and label, got {x.shape} and {y.shape}")
        x = (x - np.float32(a_min)) / (np.float32(a_max) - np.float32(a_min))
        x = np.clip(x * (np.float32(b_max) - np.float32(b_min)) + np.float32(b_min), b_min, b_max).astype(np.float32)
        xs.append(x); ys.append(y)
    shapes = {a.shape for a in xs}
    if len(shapes) != 1:
        raise ValueError(f"slices of different sizes in one run: {sorted(shapes)}")
    X = torch.from_numpy(np.stack(xs))[:, None].to(device)
    Y = torch.from_numpy(np.stack(ys))[:, None].to(device)
    return X, Y, [n for n, _, _ in items]


def run_test(checkpoint_path: str, images_dir: str, labels_dir: str, channels=None, strides=None,
             num_res_units=None, roi: int = 256, sw_batch_size: int = 4, compute_dtype=torch.bfloat16, device="cuda"):
    """Returns (per_volume {volume: dice}, summary {average, max, min, mean_slice_dice}).  ``channels`` /
    ``num_res_units`` default to what the checkpoint's tensor shapes say (the reference trains 32-512 channels,
    reference/train.py:82; the benchmark configurations 16-256)."""
    state = torch.load(checkpoint_path, map_location="cpu", weights_only=True)
    cfg = unet_config_from_state_dict(state)
    if cfg["spatial_dims"] != 2:
        raise ValueError("the legacy test run evaluates 2-D slices; this checkpoint holds a 3-D network")
    channels = tuple(channels) if channels else cfg["channels"]
    strides = tuple(strides) if strides else (2,) * (len(channels) - 1)
    model = UNet(2, 1, 1, channels, strides, num_res_units=num_res_units or cfg["num_res_units"], compute_dtype=compute_dtype)
    model.load_state_dict(state)
    model.to(device).eval()
    X, Y, names = load_slices(list_slices(images_dir, labels_dir), device=device)
    table, mean = VolumeEvaluator(model, (roi, roi), sw_batch_size=sw_batch_size).evaluate(X, Y, names)
    vals = list(table.values())
    summary = {"average": sum(vals) / len(vals) if vals else float("nan"), "max": max(vals) if vals else float("nan"),
               "min": min(vals) if vals else float("nan"), "mean_slice_dice": mean}
    return table, summary


def main(argv=None):
    ap = argparse.ArgumentParser(description="Evaluate a U-Net checkpoint on a directory of 2-D NIfTI slices")
    ap.add_argument("--checkpoint_name", required=True, help="file name of the .pth state dict, e.g. unet_task03_liver_box_pseudolabels_2D.pth")
    ap.add_argument("--model_dir", default="output/unet/pretrained")
    ap.add_argument("--images_dir", required=True)
    ap.add_argument("--labels_dir", required=True)
    ap.add_argument("--channels", default="", help="comma-separated; default: read from the checkpoint")
    ap.add_argument("--roi", type=int, default=256)
    ap.add_argument("--sw_batch_size", type=int, default=4)
    ap.add_argument("--dtype", default="bf16", choices=("bf16", "fp32"))
    a = ap.parse_args(argv)
    channels = tuple(int(c) for c in a.channels.split(",")) if a.channels else None
    table, summary = run_test(os.path.join(a.model_dir, a.checkpoint_name), a.images_dir, a.labels_dir, channels,
                              None, roi=a.roi, sw_batch_size=a.sw_batch_size,
                              compute_dtype=torch.bfloat16 if a.dtype == "bf16" else torch.float32)
    for v, d in sorted(table.items()):
        print(f"{v:24s} {d:.4f}")
    print(f"{'Average':24s} {summary['average']:.4f}   Max {summary['max']:.4f}   Min {summary['min']:.4f}   "
          f"(mean over slices {summary['mean_slice_dice']:.4f})")
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
