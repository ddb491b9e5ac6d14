"""Minimal NIfTI-1 single-file (.nii / .nii.gz) reader and writer, enough for the reference's slice files:
2-D/3-D images written by SimpleITK (reference/create_2d_slices.py:35-40) and float32 pseudo-masks with an
identity affine written through nibabel (reference/sam_eval.py:71-75).  No nibabel / SimpleITK needed.

``read_nifti(path)`` returns ``(array, affine)`` with the array indexed ``[x, y(, z)]`` as nibabel's
``get_fdata()`` / MONAI's ``LoadImage`` present it (the file stores x fastest), scaled by
``scl_slope`` / ``scl_inter`` when the header asks for it.
"""
from __future__ import annotations

import gzip
import struct
from typing import Tuple

import numpy as np

_DTYPES = {2: np.uint8, 4: np.int16, 8: np.int32, 16: np.float32, 64: np.float64, 256: np.int8, 512: np.uint16,
           768: np.uint32, 1024: np.int64, 1280: np.uint64}
_CODES = {np.dtype(v).name: k for k, v in _DTYPES.items()}


def _open(path, mode):
    return gzip.open(path, mode) if str(path).endswith(".gz") else open(path, mode)


def read_nifti(path) -> Tuple[np.ndarray, np.ndarray]:
    with _open(path, "rb") as f:
        raw = f.read()
    if len(raw) < 352:
        raise ValueError(f"{path}: too short for a NIfTI-1 file")
    end = "<"
    if struct.unpack("<i", raw[:4])[0] != 348:
        end = ">"
        if struct.unpack(">i", raw[:4])[0] != 348:
            raise ValueError(f"{path}: not a NIfTI-1 header (sizeof_hdr != 348)")
    if raw[344:347] not in (b"n+1", b"ni1"):
        raise ValueError(f"{path}: bad magic {raw[344:348]!r}")
    if raw[344:347] == b"ni1":
        raise NotImplementedError("two-file NIfTI (.hdr/.img) is not supported")
    dim = struct.unpack(end + "8h", raw[40:56])
    datatype, bitpix = struct.unpack(end + "2h", raw[70:74])
    pixdim = struct.unpack(end + "8f", raw[76:108])
    vox_offset, slope, inter = struct.unpack(end + "3f", raw[108:120])
    qform_code, sform_code = struct.unpack(end + "2h", raw[252:256])
    if datatype not in _DTYPES:
        raise NotImplementedError(f"{path}: NIfTI datatype {datatype}")
    nd = dim[0]
    if not 1 <= nd <= 7:
        raise ValueError(f"{path}: dim[0] = {nd}")
    shape = [int(d) for d in dim[1:1 + nd]]
    while len(shape) > 1 and shape[-1] == 1:
        shape.pop()
    dt = np.dtype(_DTYPES[datatype]).newbyteorder(end)
    n = int(np.prod(shape))
    off = int(vox_offset) if vox_offset >= 352 else 352
    data = np.frombuffer(raw, dtype=dt, count=n, offset=off).reshape(shape, order="F")
    data = data.astype(dt.newbyteorder("="))
    if slope != 0 and not (slope == 1 and inter == 0) and np.isfinite(slope) and np.isfinite(inter):
        data = data.astype(np.float64) * slope + inter
    affine = np.eye(4)
    if sform_code > 0:
        affine[:3, :] = np.array(struct.unpack(end + "12f", raw[280:328])).reshape(3, 4)
    elif qform_code > 0:
        b, c, d = struct.unpack(end + "3f", raw[256:268])
        ox, oy, oz = struct.unpack(end + "3f", raw[268:280])
        a = np.sqrt(max(0.0, 1.0 - (b * b + c * c + d * d)))
        R = np.array([[a * a + b * b - c * c - d * d, 2 * (b * c - a * d), 2 * (b * d + a * c)],
                      [2 * (b * c + a * d), a * a + c * c - b * b - d * d, 2 * (c * d - a * b)],
                      [2 * (b * d - a * c), 2 * (c * d + a * b), a * a + d * d - b * b - c * c]])
        qfac = -1.0 if pixdim[0] < 0 else 1.0
        affine[:3, :3] = R * np.array([pixdim[1], pixdim[2], qfac * pixdim[3]])
        affine[:3, 3] = (ox, oy, oz)
    else:
        affine[:3, :3] = np.diag(pixdim[1:4])
    return data, affine


def write_nifti(path, array: np.ndarray, affine=None) -> None:
    """Writes ``array`` (indexed [x, y(, z)]) as a little-endian NIfTI-1 file with the given affine as sform
    (identity by default), compressed when the name ends in .gz."""
    a = np.asarray(array)
    if a.dtype.name not in _CODES:
        raise TypeError(f"unsupported dtype {a.dtype}")
    if not 1 <= a.ndim <= 7:
        raise ValueError("1 to 7 dimensions")
    aff = np.eye(4) if affine is None else np.asarray(affine, dtype=np.float64)
    hdr = bytearray(348)
    struct.pack_into("<i", hdr, 0, 348)
    dim = [a.ndim] + list(a.shape) + [1] * (7 - a.ndim)
    struct.pack_into("<8h", hdr, 40, *dim)
    struct.pack_into("<2h", hdr, 70, _CODES[a.dtype.name], a.dtype.itemsize * 8)
    zooms = np.sqrt((aff[:3, :3] ** 2).sum(0))
    struct.pack_into("<8f", hdr, 76, 1.0, *([float(z) for z in zooms] + [1.0] * 4))
    struct.pack_into("<3f", hdr, 108, 352.0, 1.0, 0.0)      # vox_offset, scl_slope, scl_inter
    hdr[123] = 2                                             # xyzt_units: millimetres
    struct.pack_into("<2h", hdr, 252, 0, 2)                  # qform unknown, sform aligned
    struct.pack_into("<12f", hdr, 280, *[float(v) for v in aff[:3, :].reshape(-1)])
    hdr[344:348] = b"n+1\0"
    body = np.asfortranarray(a.astype(a.dtype.newbyteorder("<"))).tobytes(order="F")
    with _open(path, "wb") as f:
        f.write(bytes(hdr) + b"\0\0\0\0" + body)
