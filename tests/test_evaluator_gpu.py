"""Sharded full-volume evaluator on the GPU against the oracle's validation loop (sliding-window inference,
sigmoid + threshold, per-slice Dice, per-volume means)."""
import math

import pytest
import torch

import oracle
import unet_amd
from unet_amd.evaluator import VolumeEvaluator
from test_unet_gpu import disks

pytestmark = pytest.mark.gpu


def test_per_volume_dice_matches_oracle(dev):
    torch.manual_seed(3)
    channels = (16, 32, 64)
    o = oracle.UNet(2, 1, 1, channels, (2, 2), num_res_units=2).eval()
    m = unet_amd.UNet(2, 1, 1, channels, (2, 2), num_res_units=2, compute_dtype=torch.float32)
    m.load_state_dict(o.state_dict())
    m.to(dev).eval()
    S, H = 7, 48
    g = torch.Generator().manual_seed(5)
    x = torch.rand(S, 1, H, H, generator=g)
    t = disks(S, (H, H), 9, with_two=True)
    t[3] = 0                                                     # a slice without ground truth: skipped by the mean
    names = [f"liver_{c}_{s}" for c, s in ((7, 0), (7, 1), (7, 2), (12, 0), (12, 1), (12, 2), (12, 3))]
    # oracle: reference/train.py:128-153
    metric = oracle.DiceMetric(include_background=False, reduction="mean")
    post = oracle.Compose([oracle.Activations(sigmoid=True), oracle.AsDiscrete(threshold=0.5)])
    tl = t.clone(); tl[tl == 2] = 1
    ref_slice = []
    with torch.no_grad():
        for s in range(S):
            out = oracle.sliding_window_inference(x[s:s + 1], (32, 32), 4, o)
            pred = [post(i) for i in oracle.decollate_batch(out)]
            d = metric(y_pred=pred, y=tl[s:s + 1])
            ref_slice.append(float(d.reshape(-1)[0]))
    ref_mean = float(metric.aggregate())
    ref_table = unet_amd.per_volume_dice(names, ref_slice)

    ev = VolumeEvaluator(m, (32, 32), sw_batch_size=4, slices_per_call=3)
    table, mean = ev.evaluate(x.to(dev), t.to(dev), names)
    assert set(table) == set(ref_table) == {"liver_7", "liver_12"}
    for k in table:
        assert abs(table[k] - ref_table[k]) < 2e-3, (k, table[k], ref_table[k])
    assert abs(mean - ref_mean) < 2e-3
    d = ev.slice_dice(x.to(dev), t.to(dev)).cpu()
    assert math.isnan(float(d[3])) and math.isnan(ref_slice[3])
    # a larger window batch changes nothing: windows are independent samples
    table2, mean2 = VolumeEvaluator(m, (32, 32), sw_batch_size=16, slices_per_call=7).evaluate(x.to(dev), t.to(dev), names)
    assert abs(mean2 - mean) < 1e-6 and all(abs(table2[k] - table[k]) < 1e-6 for k in table)
    assert m.training is False


def test_window_grouping_keeps_results_and_respects_other_predictors(dev, monkeypatch):
    """The package's U-Net in eval mode may be fed more windows per call than sw_batch_size (they are independent
    samples); any other predictor gets exactly the group size it was asked for."""
    torch.manual_seed(4)
    m = unet_amd.UNet(2, 1, 1, (16, 32, 64), (2, 2), num_res_units=2, compute_dtype=torch.float32).to(dev).eval()
    x = torch.rand(3, 1, 80, 80, generator=torch.Generator().manual_seed(6)).to(dev)
    grouped = unet_amd.sliding_window_inference(x, (32, 32), 4, m)
    monkeypatch.setenv("UNET_AMD_SW_EXACT", "1")
    exact = unet_amd.sliding_window_inference(x, (32, 32), 4, m)
    monkeypatch.delenv("UNET_AMD_SW_EXACT")
    assert float((grouped - exact).abs().max()) < 1e-5 * float(exact.abs().max())
    sizes = []
    unet_amd.sliding_window_inference(x, (32, 32), 4, lambda w: (sizes.append(w.shape[0]), w)[1])
    nwin = len(unet_amd.window_starts([80, 80], [32, 32], 0.25)) * 3
    assert max(sizes) == 4 and sum(sizes) == nwin
    m.train()
    sizes.clear()
    from unet_amd.inferers import _windows_per_call
    assert _windows_per_call(m, 4, nwin, 32 * 32) == (4, False)  # dropout masks depend on the position in the batch
    m.eval()
    assert _windows_per_call(m, 4, nwin, 32 * 32) == (32, True)  # 27 windows: one padded group of the next power of two
    assert _windows_per_call(m, 4, 1000, 32 * 32) == (288, True)
    assert _windows_per_call(m, 4, 3, 32 * 32) == (4, True)
