"""NIfTI-1 reader / writer: round trips, a header laid out by hand from the NIfTI-1 field table
(big-endian, int16, scl_slope / scl_inter), error cases.  No nibabel in this image: the checks are
against the published byte offsets, not against another implementation."""
import gzip
import struct

import numpy as np
import pytest

from unet_amd.nifti import read_nifti, write_nifti


def test_round_trip_float32_identity_affine(tmp_path):
    a = np.arange(12, dtype=np.float32).reshape(3, 4)          # [x, y]
    p = tmp_path / "liver_3_17.nii.gz"
    write_nifti(p, a)
    b, aff = read_nifti(p)
    assert b.dtype == np.float32 and np.array_equal(a, b) and np.array_equal(aff, np.eye(4))
    raw = gzip.open(p, "rb").read()
    assert struct.unpack("<i", raw[:4])[0] == 348 and raw[344:348] == b"n+1\0"
    assert struct.unpack("<8h", raw[40:56])[:3] == (2, 3, 4)
    assert struct.unpack("<2h", raw[70:74]) == (16, 32)
    # x runs fastest in the file
    assert np.array_equal(np.frombuffer(raw, "<f4", 12, 352), a.reshape(-1, order="F"))


def test_round_trip_3d_uint8_uncompressed(tmp_path):
    a = (np.random.RandomState(0).rand(5, 6, 7) > 0.5).astype(np.uint8)
    aff = np.diag([1.5, 1.5, 2.0, 1.0]); aff[:3, 3] = (-10, 20, 5)
    write_nifti(tmp_path / "v.nii", a, aff)
    b, aff2 = read_nifti(tmp_path / "v.nii")
    assert np.array_equal(a, b) and np.allclose(aff, aff2)


def test_hand_made_big_endian_int16_with_scaling(tmp_path):
    hdr = bytearray(348)
    struct.pack_into(">i", hdr, 0, 348)
    struct.pack_into(">8h", hdr, 40, 3, 2, 3, 1, 1, 1, 1, 1)   # a 2 x 3 x 1 volume
    struct.pack_into(">2h", hdr, 70, 4, 16)                     # int16
    struct.pack_into(">8f", hdr, 76, 1, 0.8, 0.8, 5, 1, 1, 1, 1)
    struct.pack_into(">3f", hdr, 108, 352, 2.0, -1024.0)        # vox_offset, slope, intercept
    hdr[344:348] = b"n+1\0"
    vals = np.array([0, 1, 2, 3, 4, 5], dtype=">i2")
    (tmp_path / "be.nii").write_bytes(bytes(hdr) + b"\0\0\0\0" + vals.tobytes())
    data, aff = read_nifti(tmp_path / "be.nii")
    assert data.shape == (2, 3)
    assert np.array_equal(data, np.array([[0, 2, 4], [1, 3, 5]]) * 2.0 - 1024.0)
    assert np.allclose(np.diag(aff)[:3], (0.8, 0.8, 5))


def test_rejects_foreign_files(tmp_path):
    (tmp_path / "x.nii").write_bytes(b"\0" * 400)
    with pytest.raises(ValueError):
        read_nifti(tmp_path / "x.nii")
    with pytest.raises(TypeError):
        write_nifti(tmp_path / "y.nii", np.zeros((2, 2), dtype=np.complex64))
