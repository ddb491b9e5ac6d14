"""CPU tests of the oracle: known-answer parameter counts, state_dict layout, closed-form loss / metric
values, sliding-window identities and the committed golden vectors (tests/golden, written by
``python -m oracle.make_golden``)."""
import math
import os

import numpy as np
import pytest
import torch

import oracle
from oracle.make_golden import KINK_MARGIN, grad_summary, inputs, run, sample_index, weight_checksum

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def nparams(m):
    return sum(p.numel() for p in m.parameters())


def test_known_answer_parameter_counts():
    # MONAI's spleen-tutorial network (3-D, 1 -> 2, batch norm): the published count
    assert nparams(oracle.UNet(3, 1, 2, (16, 32, 64, 128, 256), (2, 2, 2, 2), num_res_units=2, norm="BATCH")) == 4_808_917
    # instance norm has no affine parameters: 2 * 1218 fewer
    assert nparams(oracle.UNet(3, 1, 2, (16, 32, 64, 128, 256), (2, 2, 2, 2), num_res_units=2)) == 4_806_481
    # the reference's 2-D networks (reference/train.py:78-86)
    assert nparams(oracle.UNet(2, 1, 1, (16, 32, 64, 128, 256), (2, 2, 2, 2), num_res_units=2)) == 1_624_844
    assert nparams(oracle.UNet(2, 1, 1, (32, 64, 128, 256, 512), (2, 2, 2, 2), num_res_units=2)) == 6_494_716
    assert nparams(oracle.UNet(3, 1, 1, (16, 32, 64, 128, 256), (2, 2, 2, 2), num_res_units=2)) == 4_805_534


def test_state_dict_layout():
    sd = oracle.UNet(2, 1, 1, (16, 32, 64, 128, 256), (2, 2, 2, 2), num_res_units=2).state_dict()
    keys = list(sd.keys())
    assert len(keys) == 63
    assert keys[0] == "model.0.conv.unit0.conv.weight" and tuple(sd[keys[0]].shape) == (16, 1, 3, 3)
    assert "model.0.residual.weight" in sd and tuple(sd["model.0.residual.weight"].shape) == (16, 1, 3, 3)
    assert tuple(sd["model.1.submodule.1.submodule.1.submodule.1.submodule.residual.weight"].shape) == (256, 128, 1, 1)
    assert tuple(sd["model.2.0.conv.weight"].shape) == (32, 1, 3, 3)          # transposed conv: [in, out, k, k]
    assert tuple(sd["model.2.1.conv.unit0.conv.weight"].shape) == (1, 1, 3, 3)
    assert "model.2.1.conv.unit0.adn.A.weight" not in sd                        # last unit is conv only
    assert float(sd["model.0.conv.unit0.adn.A.weight"]) == 0.25


def test_forward_shape_and_divisibility():
    m = oracle.UNet(2, 1, 1, (16, 32, 64), (2, 2), num_res_units=2).eval()
    assert m(torch.zeros(1, 1, 32, 48)).shape == (1, 1, 32, 48)
    m3 = oracle.UNet(3, 1, 1, (16, 32), (2,), num_res_units=2).eval()
    assert m3(torch.zeros(1, 1, 8, 8, 8)).shape == (1, 1, 8, 8, 8)


def test_loss_closed_forms():
    f = np.load(os.path.join(GOLD, "closed_forms.npz"))
    t = torch.from_numpy(f["target"])
    loss = oracle.DiceCELoss(sigmoid=True)
    # zero logits: p = 0.5 everywhere, half the pixels foreground: dice = 1 - (2*16+s)/(32+32+s), bce = ln 2
    s = 1e-5
    expect = 1 - (2 * 16 + s) / (32 + 32 + s) + math.log(2.0)
    assert abs(float(loss(torch.zeros_like(t), t)) - expect) < 1e-6
    assert abs(float(f["loss_zero_logits"]) - expect) < 1e-6
    # saturated correct logits: both terms vanish
    assert float(loss((2 * t - 1) * 30.0, t)) < 1e-6 and float(f["loss_perfect"]) < 1e-6
    # saturated wrong logits: dice -> 1 - s/(32+s) ~ 1, bce = 30
    assert abs(float(f["loss_inverted"]) - (1.0 + 30.0)) < 1e-3
    # empty target and empty prediction: dice term 1 - s/s = 0
    z = torch.zeros(1, 1, 4, 4)
    assert float(oracle.DiceLoss(sigmoid=True)(z - 40.0, z)) < 1e-6
    # error behaviour follows MONAI
    with pytest.raises(ValueError):
        loss(torch.zeros(1, 1, 4, 4), torch.zeros(1, 4, 4))
    with pytest.raises(AssertionError):
        loss(torch.zeros(1, 1, 4, 4), torch.zeros(1, 1, 4, 5))


def test_metric_closed_forms():
    m = oracle.DiceMetric(include_background=False, reduction="mean")
    y = torch.zeros(3, 1, 4, 4)
    p = torch.zeros(3, 1, 4, 4)
    y[0, 0, :2] = 1; p[0, 0, :2] = 1            # perfect: 1
    y[1, 0, :2] = 1; p[1, 0, 1:3] = 1           # half overlap: 2*4/(8+8) = 0.5
    p[2, 0, 0] = 1                               # empty ground truth: NaN, ignored by the mean
    m(y_pred=p, y=y)
    assert abs(float(m.aggregate()) - 0.75) < 1e-7
    m.reset()
    m(y_pred=[p[2]], y=y[2:3])
    assert float(m.aggregate()) == 0.0               # MONAI's nan-mean of an all-NaN buffer is 0
    d, se, sp = oracle.calculate_dice_score(p[1], y[1])
    assert abs(float(d) - 0.5) < 1e-6 and abs(float(se) - 0.5) < 1e-6 and abs(float(sp) - 0.5) < 1e-6


def test_post_transform_threshold_is_inclusive():
    post = oracle.Compose([oracle.Activations(sigmoid=True), oracle.AsDiscrete(threshold=0.5)])
    out = post(torch.tensor([[-1.0, 0.0, 1.0]]))
    assert out.tolist() == [[0.0, 1.0, 1.0]]


def test_sliding_window_identities():
    x = torch.rand(2, 1, 40, 56)
    ident = oracle.sliding_window_inference(x, (16, 16), 4, lambda w: w)
    assert torch.allclose(ident, x, atol=1e-6)
    # a window as large as the image is a single call
    calls = []
    oracle.sliding_window_inference(x, (64, 64), 4, lambda w: (calls.append(tuple(w.shape)), w)[1])
    assert calls == [(2, 1, 64, 64)]                 # image is padded up to the window
    # 512 with roi 256 and overlap 0.25: starts 0, 128(?), 256 -> 3 per axis (9 windows per slice)
    assert oracle.dense_patch_starts((512, 512), (256, 256), (192, 192)).__len__() == 9


@pytest.mark.parametrize("fname,shape", [("unet2d_16_256.npz", (2, 1, 64, 64)), ("unet3d_small.npz", (1, 1, 32, 32, 32))])
def test_oracle_reproduces_seed_built_golden(fname, shape):
    f = np.load(os.path.join(GOLD, fname))
    dims, channels = int(f["dims"]), tuple(int(c) for c in f["channels"])
    torch.manual_seed(0)
    m = oracle.UNet(dims, 1, 1, channels, (2,) * (len(channels) - 1), num_res_units=2, dropout=0.0)
    assert abs(weight_checksum(m) - float(f["weight_checksum"])) < 1e-9, "torch RNG stream or construction order changed"
    x, t = inputs(shape, int(f["input_seed"]))
    assert np.array_equal(x.numpy(), f["x"]) and np.array_equal(t.numpy(), f["t"])
    # the fixture was written by the fp64 oracle on an input that keeps every PReLU input away from 0
    m64 = oracle.UNet(dims, 1, 1, channels, (2,) * (len(channels) - 1), num_res_units=2, dropout=0.0).double()
    m64.load_state_dict({k: v.double() for k, v in m.state_dict().items()})
    m64.eval()
    from oracle.inject import record_preactivations
    margin = min(float(d.abs().min()) for d in record_preactivations(m64, x.double()))
    assert margin > KINK_MARGIN and abs(margin - float(f["kink_margin"])) < 1e-9
    y, loss = run(m64, x.double(), t.double())
    assert np.allclose(y.numpy(), f["logits"], rtol=0, atol=1e-6 * np.abs(f["logits"]).max())
    assert abs(loss - float(f["loss"])) < 1e-9
    names, sums, norms, firsts = grad_summary(m64)
    assert list(names) == list(f["grad_names"])
    assert np.allclose(norms, f["grad_norms"], rtol=1e-9, atol=1e-12)
    for i, (_, p) in enumerate(m64.named_parameters()):
        g = p.grad.flatten().numpy()
        idx = sample_index(i, g.size)
        assert np.allclose(g[idx], f["grad_samples"][i, :idx.size], rtol=1e-9, atol=1e-12)
    # the fp32 oracle (what the CPU baseline and the other parity tests run) agrees with it
    y32, loss32 = run(m, x, t)
    assert np.allclose(y32.numpy(), f["logits"], rtol=0, atol=1e-5 * np.abs(f["logits"]).max())
    assert abs(loss32 - float(f["loss"])) < 1e-6


def test_oracle_reproduces_stored_weight_golden():
    f = np.load(os.path.join(GOLD, "unet2d_small.npz"))
    m = oracle.UNet(2, 1, 1, (16, 32, 64), (2, 2), num_res_units=2, dropout=0.0)
    m.load_state_dict({k[6:]: torch.from_numpy(f[k]) for k in f.files if k.startswith("param:")})
    y, loss = run(m, torch.from_numpy(f["x"]), torch.from_numpy(f["t"]))
    assert np.allclose(y.numpy(), f["logits"], rtol=0, atol=1e-5 * np.abs(f["logits"]).max())
    assert abs(loss - float(f["loss"])) < 1e-6
    for n, p in m.named_parameters():
        g = f["grad:" + n]
        assert np.allclose(p.grad.numpy(), g, rtol=0, atol=1e-4 * max(np.abs(g).max(), 1e-6)), n
