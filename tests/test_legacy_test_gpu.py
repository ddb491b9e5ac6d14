"""The legacy test run end to end on the GPU: NIfTI slice files + a saved checkpoint -> per-volume Dice
table, compared with the same evaluation done directly on tensors."""
import numpy as np
import pytest
import torch

import unet_amd
from unet_amd.nifti import write_nifti
from unet_amd.testing import list_slices, run_test
from test_unet_gpu import disks

pytestmark = pytest.mark.gpu


def test_run_test_on_nifti_directory(dev, tmp_path):
    torch.manual_seed(11)
    channels = (16, 32, 64)
    model = unet_amd.UNet(2, 1, 1, channels, (2, 2), num_res_units=2, compute_dtype=torch.float32)
    ck = unet_amd.save_checkpoint(model, str(tmp_path / "models" / unet_amd.checkpoint_name("Task03_Liver", use_pseudo_labels=True, sam_prompt="box")))
    assert ck.endswith("unet_task03_liver_box_pseudolabels_2D.pth")       # the file reference/README.md:28 tests
    rng = np.random.RandomState(1)
    (tmp_path / "img").mkdir(); (tmp_path / "lbl").mkdir()
    names = ["liver_2_10", "liver_2_9", "liver_15_0", "liver_15_1", "liver_15_2"]
    lab = disks(len(names), (48, 48), 4).numpy()[:, 0]
    raws = []
    for i, n in enumerate(names):
        raw = rng.uniform(-300, 400, (48, 48)).astype(np.float32)
        raws.append(raw)
        write_nifti(tmp_path / "img" / f"{n}.nii.gz", raw)
        write_nifti(tmp_path / "lbl" / f"{n}.nii.gz", lab[i].astype(np.float32))
    items = list_slices(str(tmp_path / "img"), str(tmp_path / "lbl"))
    assert [n for n, _, _ in items] == ["liver_15_0", "liver_15_1", "liver_15_2", "liver_2_9", "liver_2_10"]
    # channels and residual units are read from the checkpoint
    table, summary = run_test(ck, str(tmp_path / "img"), str(tmp_path / "lbl"), roi=32, compute_dtype=torch.float32)
    # the same evaluation straight from the arrays
    order = [names.index(n) for n, _, _ in items]
    x = np.clip((np.stack(raws)[order] + 175.0) / 425.0, 0, 1).astype(np.float32)
    m = model.to(dev).eval()
    ref, mean = unet_amd.VolumeEvaluator(m, (32, 32)).evaluate(torch.from_numpy(x)[:, None].to(dev),
                                                                   torch.from_numpy(lab[order])[:, None].to(dev),
                                                                   [n for n, _, _ in items])
    assert set(table) == {"liver_2", "liver_15"}
    for k in table:
        assert abs(table[k] - ref[k]) < 1e-4
    assert abs(summary["mean_slice_dice"] - mean) < 1e-4
    assert summary["min"] <= summary["average"] <= summary["max"]
