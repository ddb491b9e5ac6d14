"""Input pipeline: CPU checks of the numpy oracle on hand-made cases, and exact GPU parity of the HIP
pipeline against it on the same uniform numbers."""
import numpy as np
import pytest
import torch

from oracle import pipeline_ref as ref


def test_oracle_intensity_window_and_box():
    x = np.array([[-1000.0, -175.0, 37.5], [250.0, 3000.0, -174.0]], dtype=np.float32)
    y = ref.scale_intensity_range(x, -175, 250, 0, 1, True)
    assert np.allclose(y, [[0, 0, 0.5], [1, 1, 1 / 425]], atol=1e-6)
    assert ref.foreground_box(y) == (0, 2, 0, 3)
    assert ref.foreground_box(np.zeros((4, 5), np.float32)) == (0, 4, 0, 5)
    z = np.zeros((6, 7), np.float32); z[2:4, 3:6] = 1
    assert ref.foreground_box(z) == (2, 4, 3, 6)


def test_oracle_pad_and_center_correction():
    a = np.ones((3, 6), np.float32)
    p = ref.spatial_pad(a, (8, 4))
    assert p.shape == (8, 6) and p[:2].sum() == 0 and p[2:5].sum() == 18 and p[5:].sum() == 0   # 5 rows of padding: 2 before, 3 after
    assert ref.correct_crop_centers([0, 500], (256, 256), (256, 300)) == [128, 172]
    assert ref.correct_crop_centers([130, 10], (256, 256), (300, 300)) == [130, 128]


def test_oracle_crops_contain_the_chosen_pixel_and_rotate():
    rng = np.random.RandomState(0)
    img = rng.uniform(-300, 400, (40, 48)).astype(np.float32)
    lbl = np.zeros((40, 48), np.float32); lbl[10:14, 20:25] = 1
    u = np.array([[0.1, 0.0, 0.9, 0.0], [0.9, 0.5, 0.1, 0.99]], np.float32)   # fg first pixel, no rotation; bg, 3 quarter turns
    ci, cl = ref.train_transforms_2d(img, lbl, u, spatial_size=(16, 16), num_samples=2)
    assert ci[0].shape == (16, 16) and cl[0].sum() > 0      # the foreground crop holds foreground
    base_i, _ = ref.train_transforms_2d(img, lbl, np.array([[0.9, 0.5, 0.9, 0.0]], np.float32), spatial_size=(16, 16), num_samples=1)
    assert np.array_equal(ci[1], np.rot90(base_i[0], 3))


@pytest.mark.gpu
@pytest.mark.parametrize("img_dtype", [torch.float32, torch.int16])
def test_gpu_pipeline_matches_oracle_exactly(dev, img_dtype):
    from unet_amd.pipeline import SliceCropPipeline
    rng = np.random.RandomState(3)
    S, H, W, roi, ns = 5, 96, 80, 64, 4
    raw = rng.uniform(-400, 500, (S, H, W)).astype(np.float32)
    raw[:, :7] = -1000; raw[:, :, :5] = -1000; raw[:, :, 70:] = -1000        # air frame: the foreground box is smaller than the slice
    raw[3, 30:, :] = -1000                                                      # a slice shorter than the crop: padding
    if img_dtype == torch.int16:
        raw = np.round(raw)
    lbl = np.zeros((S, H, W), np.float32)
    lbl[0, 40:50, 30:45] = 1; lbl[1, 20:22, 10:12] = 2; lbl[3, 10:20, 20:40] = 1; lbl[4, 60:90, 8:60] = 1   # slice 2: no foreground
    u = rng.uniform(0, 1, (S * ns, 4)).astype(np.float32)
    pipe = SliceCropPipeline(spatial_size=(roi, roi), num_samples=ns)
    gi, gl = pipe(torch.from_numpy(raw).to(img_dtype).to(dev), torch.from_numpy(lbl).to(dev), uniforms=torch.from_numpy(u))
    torch.cuda.synchronize()
    gi, gl = gi.cpu().numpy(), gl.cpu().numpy()
    for s in range(S):
        ri, rl = ref.train_transforms_2d(raw[s], lbl[s], u[s * ns:(s + 1) * ns], spatial_size=(roi, roi), num_samples=ns)
        for j in range(ns):
            assert np.array_equal(gl[s * ns + j, 0], rl[j]), (s, j)
            assert np.array_equal(gi[s * ns + j, 0], ri[j]), (s, j)
    # labels given as uint8 give the same crops
    gi2, gl2 = pipe(torch.from_numpy(raw).to(img_dtype).to(dev), torch.from_numpy(lbl).to(torch.uint8).to(dev), uniforms=torch.from_numpy(u))
    assert torch.equal(gi2.cpu(), torch.from_numpy(gi)) and torch.equal(gl2.cpu(), torch.from_numpy(gl))
