"""CPU oracle (numpy) for the 2-D training transforms of reference/unetDataset.py:68-93, restating the
MONAI 1.3 transforms it composes.  TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED: MONAI is not installed and
the reference holds no fixtures for its data pipeline; the random draws are taken from an explicit
array of uniform numbers (MONAI draws them from a NumPy RandomState).

    ScaleIntensityRanged -> CropForegroundd(select image > 0) -> SpatialPadd(symmetric, constant 0)
    -> RandCropByPosNegLabeld(pos, neg, num_samples, image_threshold) -> RandRotate90d(prob, axes (0,1))
"""
from __future__ import annotations

import numpy as np


def scale_intensity_range(img, a_min, a_max, b_min, b_max, clip):
    # monai.transforms.ScaleIntensityRange.__call__
    img = (img.astype(np.float32) - np.float32(a_min)) / (np.float32(a_max) - np.float32(a_min))
    img = img * (np.float32(b_max) - np.float32(b_min)) + np.float32(b_min)
    if clip:
        img = np.clip(img, np.float32(b_min), np.float32(b_max))
    return img.astype(np.float32)


def foreground_box(img):
    # monai.transforms.utils.generate_spatial_bounding_box with select_fn = (x > 0), margin 0;
    # an image without foreground keeps its full extent here
    m = img > 0
    if not m.any():
        return (0, img.shape[0], 0, img.shape[1])
    ys, xs = np.where(m.any(1))[0], np.where(m.any(0))[0]
    return (int(ys[0]), int(ys[-1]) + 1, int(xs[0]), int(xs[-1]) + 1)


def spatial_pad(a, size):
    # monai.transforms.SpatialPad(method="symmetric", mode="constant")
    pads = []
    for d in range(2):
        w = max(size[d] - a.shape[d], 0)
        pads.append((w // 2, w - w // 2))
    return np.pad(a, pads, mode="constant")


def correct_crop_centers(center, roi, shape):
    # monai.transforms.utils.correct_crop_centers
    out = []
    for c, r, s in zip(center, roi, shape):
        valid_start = r // 2
        valid_end = int(np.floor(s + 1 - r / 2))
        if valid_end == valid_start:
            valid_end += 1
        out.append(min(max(int(c), valid_start), valid_end - 1))
    return out


def train_transforms_2d(image, label, uniforms, a_min=-175.0, a_max=250.0, b_min=0.0, b_max=1.0, clip=True,
                        spatial_size=(256, 256), pos=2.0, neg=1.0, num_samples=4, image_threshold=0.0, rot_prob=0.5, max_k=3):
    """image, label: [H, W] arrays; uniforms: [num_samples, 4].  Returns lists of num_samples (image, label) crops."""
    img = scale_intensity_range(np.asarray(image), a_min, a_max, b_min, b_max, clip)
    lbl = np.asarray(label).astype(np.float32)
    y0, y1, x0, x1 = foreground_box(img)
    img, lbl = img[y0:y1, x0:x1], lbl[y0:y1, x0:x1]
    img, lbl = spatial_pad(img, spatial_size), spatial_pad(lbl, spatial_size)
    # monai.transforms.utils.map_binary_to_indices
    fg = np.nonzero((lbl > 0).ravel())[0]
    bg = np.nonzero(((lbl <= 0) & (img > image_threshold)).ravel())[0]
    pos_ratio = pos / (pos + neg)
    outs_i, outs_l = [], []
    for s in range(num_samples):
        u = uniforms[s]
        # monai.transforms.utils.generate_pos_neg_label_crop_centers
        if len(fg) == 0 and len(bg) == 0:
            center = [img.shape[0] // 2 - (y1 - y0 < spatial_size[0]) * 0, img.shape[1] // 2]
            # degenerate slice: the centre of the unpadded box (shifted by the padding)
            ph, pw = max(spatial_size[0] - (y1 - y0), 0) // 2, max(spatial_size[1] - (x1 - x0), 0) // 2
            center = [(y1 - y0) // 2 + ph, (x1 - x0) // 2 + pw]
        else:
            if len(fg) == 0:
                idx = bg
            elif len(bg) == 0:
                idx = fg
            else:
                idx = fg if np.float32(u[0]) < np.float32(pos_ratio) else bg
            k = min(int(np.float64(np.float32(u[1])) * len(idx)), len(idx) - 1)
            center = list(np.unravel_index(idx[k], img.shape))
        center = correct_crop_centers(center, spatial_size, img.shape)
        sy, sx = center[0] - spatial_size[0] // 2, center[1] - spatial_size[1] // 2
        ci = img[sy:sy + spatial_size[0], sx:sx + spatial_size[1]]
        cl = lbl[sy:sy + spatial_size[0], sx:sx + spatial_size[1]]
        # monai.transforms.RandRotate90
        if u[2] < rot_prob:
            kk = min(int(np.float64(np.float32(u[3])) * max_k), max_k - 1) + 1
            ci, cl = np.rot90(ci, kk, (0, 1)), np.rot90(cl, kk, (0, 1))
        outs_i.append(np.ascontiguousarray(ci))
        outs_l.append(np.ascontiguousarray(cl))
    return outs_i, outs_l
