"""GPU-resident input pipeline for the 2-D training crops: the arithmetic of ``train_transforms_2d``
(reference/unetDataset.py:68-93) on slices that already sit in HBM.

    ScaleIntensityRanged(a_min, a_max, b_min, b_max, clip)  ->  CropForegroundd(source = image > 0)
    ->  SpatialPadd(spatial_size)  ->  RandCropByPosNegLabeld(spatial_size, pos, neg, num_samples,
    image_threshold)  ->  RandRotate90d(prob, spatial_axes=(0, 1))

Three kernels do the work (foreground bounding box, crop-centre selection, crop + scale + rotate); the
few integers per output sample in between are torch tensor arithmetic on the device, so nothing
synchronises with the host.  Randomness: four uniform numbers per output sample (class choice, pixel
rank, rotate or not, number of quarter turns) from a ``torch.Generator``; MONAI draws from a NumPy
``RandomState`` instead, so identical seeds do not give identical crops - identical uniforms do.
"""
from __future__ import annotations

from typing import Optional, Sequence, Tuple

import torch

from . import _lib

_IMG_DT = {torch.float32: 0, torch.int16: 1}


class SliceCropPipeline:
    def __init__(self, a_min: float = -175.0, a_max: float = 250.0, b_min: float = 0.0, b_max: float = 1.0, clip: bool = True,
                 spatial_size: Sequence[int] = (256, 256), pos: float = 2.0, neg: float = 1.0, num_samples: int = 4,
                 image_threshold: float = 0.0, rot_prob: float = 0.5, max_k: int = 3):
        if pos < 0 or neg < 0 or pos + neg == 0:
            raise ValueError("pos and neg must be nonnegative, got pos=%s neg=%s." % (pos, neg))
        if spatial_size[0] != spatial_size[1]:
            raise NotImplementedError("square crops only (quarter-turn rotations)")
        if b_max <= b_min or a_max <= a_min:
            raise ValueError("a_max must exceed a_min and b_max must exceed b_min")
        self.a_min, self.a_max, self.b_min, self.b_max, self.clip = float(a_min), float(a_max), float(b_min), float(b_max), bool(clip)
        self.roi = int(spatial_size[0])
        self.pos_ratio = float(pos) / float(pos + neg)
        self.num_samples, self.rot_prob, self.max_k = int(num_samples), float(rot_prob), int(max_k)
        # "scaled image > t"  <=>  "raw image > raw_thr(t)" because the intensity map is increasing
        scale = (self.a_max - self.a_min) / (self.b_max - self.b_min)
        self.fg_thr = self.a_min + (0.0 - self.b_min) * scale
        self.bg_thr = self.a_min + (float(image_threshold) - self.b_min) * scale
        if self.clip and (float(image_threshold) < self.b_min or 0.0 < self.b_min):
            # below b_min every clipped value equals b_min > threshold: everything passes
            self.fg_thr = float("-inf") if 0.0 < self.b_min else self.fg_thr
            self.bg_thr = float("-inf") if float(image_threshold) < self.b_min else self.bg_thr

    def descriptors(self, images: torch.Tensor, labels: torch.Tensor, uniforms: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """(desc [B,4] int32 = slice, oy, ox, k ; bbox [S,4] int32), all on the device."""
        lib = _lib.load()
        S, H, W = images.shape
        dev = images.device
        st = _lib.current_stream(dev)
        B = S * self.num_samples
        bbox = torch.empty(S, 4, dtype=torch.int32, device=dev)
        _lib.check(lib.mi_fg_bbox(images.data_ptr(), _IMG_DT[images.dtype], self.fg_thr, S, H, W, bbox.data_ptr(), st), "mi_fg_bbox")
        sample_slice = torch.arange(S, dtype=torch.int32, device=dev).repeat_interleave(self.num_samples)
        u = uniforms.to(device=dev, dtype=torch.float32).contiguous()
        u01 = u[:, :2].contiguous()
        centers = torch.empty(B, 2, dtype=torch.int32, device=dev)
        _lib.check(lib.mi_pick_centers(images.data_ptr(), _IMG_DT[images.dtype], labels.data_ptr(), _lib.label_dtype_code(labels),
                                       self.bg_thr, bbox.data_ptr(), H, W, sample_slice.data_ptr(), u01.data_ptr(),
                                       self.pos_ratio, B, centers.data_ptr(), st), "mi_pick_centers")
        roi = self.roi
        bb = bbox[sample_slice.long()].long()                       # [B,4]
        size = torch.stack([bb[:, 1] - bb[:, 0], bb[:, 3] - bb[:, 2]], 1)
        pad_lo = torch.clamp(roi - size, min=0) // 2                # SpatialPad, symmetric
        shape = torch.clamp(size, min=roi)                          # padded extent
        c = centers.long() + pad_lo
        valid_start = roi // 2
        valid_end = shape + 1 - roi // 2 - int(roi % 2 > 0)       # floor(shape + 1 - roi / 2)
        valid_end = torch.where(valid_end == valid_start, valid_end + 1, valid_end)
        c = torch.minimum(torch.clamp(c, min=valid_start), valid_end - 1)
        start = c - roi // 2                                        # crop origin in padded box coordinates
        origin = torch.stack([bb[:, 0], bb[:, 2]], 1) - pad_lo + start
        k = torch.where(u[:, 2] < self.rot_prob, torch.clamp((u[:, 3].double() * self.max_k).long(), max=self.max_k - 1) + 1,
                        torch.zeros(B, dtype=torch.long, device=dev))
        desc = torch.stack([sample_slice.long(), origin[:, 0], origin[:, 1], k], 1).to(torch.int32).contiguous()
        return desc, bbox

    def __call__(self, images: torch.Tensor, labels: torch.Tensor, generator: Optional[torch.Generator] = None,
                 uniforms: Optional[torch.Tensor] = None):
        """images [S,H,W] (fp32 or int16, raw intensities), labels [S,H,W] -> (image, label) [S*num_samples,1,roi,roi] fp32."""
        if not images.is_cuda:
            raise RuntimeError("unet_amd.SliceCropPipeline runs on an AMD GPU only (no CPU or eager fallback)")
        if images.dim() != 3 or labels.shape != images.shape:
            raise ValueError("images and labels must both be [S, H, W]")
        if images.dtype not in _IMG_DT:
            raise TypeError("images must be float32 or int16")
        images = images.contiguous()
        labels = labels.contiguous()
        if _lib.label_dtype_code(labels) < 0:
            labels = labels.float()
        S, H, W = images.shape
        B = S * self.num_samples
        if uniforms is None:
            uniforms = torch.rand(B, 4, generator=generator, device=images.device if generator is None or generator.device.type == "cuda" else "cpu")
        if tuple(uniforms.shape) != (B, 4):
            raise ValueError(f"uniforms must be [{B}, 4]")
        desc, bbox = self.descriptors(images, labels, uniforms)
        lib = _lib.load()
        out_i = torch.empty(B, 1, self.roi, self.roi, dtype=torch.float32, device=images.device)
        out_l = torch.empty_like(out_i)
        _lib.check(lib.mi_crop_scale_rot(images.data_ptr(), _IMG_DT[images.dtype], labels.data_ptr(), _lib.label_dtype_code(labels),
                                         H, W, desc.data_ptr(), bbox.data_ptr(), B, self.a_min, self.a_max, self.b_min, self.b_max,
                                         1 if self.clip else 0, self.roi, self.roi, out_i.data_ptr(), out_l.data_ptr(),
                                         _lib.current_stream(images.device)), "mi_crop_scale_rot")
        return out_i, out_l
