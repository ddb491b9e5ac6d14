"""Properties that hold at any size, checked at the benchmark's full size (32 slices of 512 x 512, channels
16-256), where the CPU oracle would take minutes:

* instance norm is per sample, so permuting the batch permutes the logits bit for bit and leaves every
  parameter gradient unchanged up to summation order;
* the loss of a batch is the mean of the losses of its halves (Dice and BCE are means over samples / pixels);
* a sliding window as large as the image is the plain forward pass;
* the directional derivative of the loss along a random direction matches <gradient, direction> (fp32);
* one fused trainer step equals forward + loss + backward + AdamW done through autograd;
* the 3-D benchmark shape (4 patches of 96^3): batch permutation and directional derivative.
"""
import pytest
import torch

import unet_amd
from test_unet_gpu import disks

pytestmark = pytest.mark.gpu
CH = (16, 32, 64, 128, 256)


def _model(dtype, dev, seed=0):
    torch.manual_seed(seed)
    return unet_amd.UNet(2, 1, 1, CH, (2, 2, 2, 2), num_res_units=2, compute_dtype=dtype).to(dev)


def _batch(nb, size, dev, seed=3):
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(nb, 1, size, size, generator=g)
    t = disks(nb, (size, size), seed + 1, with_two=True)
    return x.to(dev), t.to(dev)


def test_batch_permutation_equivariance_full_size(dev):
    m = _model(torch.bfloat16, dev).train()
    x, t = _batch(32, 512, dev)
    perm = torch.randperm(32, generator=torch.Generator().manual_seed(1)).to(dev)
    loss = unet_amd.DiceCELoss(sigmoid=True, fold_label_two=True)
    y = m(x)
    l = loss(y, t); l.backward()
    g = [p.grad.clone() for p in m.parameters()]
    m.zero_grad()
    yp = m(x[perm])
    lp = loss(yp, t[perm]); lp.backward()
    torch.cuda.synchronize()
    assert torch.equal(yp, y[perm])                                   # same samples, same bits
    assert abs(float(l.detach()) - float(lp.detach())) < 1e-6
    gn = torch.sqrt(sum((a.float() ** 2).sum() for a in g))
    dn = torch.sqrt(sum(((a - p.grad).float() ** 2).sum() for a, p in zip(g, m.parameters())))
    assert float(dn / gn) < 1e-3                                      # only the summation order over the batch changes


def test_loss_of_batch_is_mean_of_halves_full_size(dev):
    m = _model(torch.bfloat16, dev).eval()
    x, t = _batch(32, 512, dev)
    loss = unet_amd.DiceCELoss(sigmoid=True, fold_label_two=True)
    with torch.no_grad():
        y = m(x)
        whole = float(loss(y, t))
        halves = 0.5 * (float(loss(y[:16], t[:16])) + float(loss(y[16:], t[16:])))
    assert abs(whole - halves) < 1e-6


def test_window_as_large_as_the_image_is_the_forward_pass(dev):
    m = _model(torch.bfloat16, dev).eval()
    x, _ = _batch(4, 512, dev)
    with torch.no_grad():
        a = m(x)
        b = unet_amd.sliding_window_inference(x, (512, 512), 4, m)
    assert torch.equal(a, b)


def test_nine_window_sliding_inference_full_size(dev):
    """roi 256 on 512 x 512 (reference/train.py:140 with the legacy crop size; BASELINE.json configs[4]): 3 x 3
    windows per slice at starts 0, 192, 256.  Properties that hold for any predictor, then equality with the
    per-window forward passes blended by hand."""
    x, _ = _batch(4, 512, dev, seed=11)
    starts = unet_amd.window_starts((512, 512), (256, 256), 0.25)
    assert starts == [(a, b) for a in (0, 192, 256) for b in (0, 192, 256)]
    # identity predictor: overlap-averaging copies of the same pixels gives the image back
    assert torch.equal(unet_amd.sliding_window_inference(x, (256, 256), 4, lambda w: w), x)
    # constant predictor: every pixel is an average of equal constants
    out = unet_amd.sliding_window_inference(x, (256, 256), 4, lambda w: torch.full_like(w, 3.25))
    assert torch.equal(out, torch.full_like(x, 3.25))
    # per-window forward of the U-Net (one window per call), accumulated and divided by the cover count in torch;
    # fp32 storage: a batch of one window takes other kernel variants (contraction split) than a batch of 36, and
    # only fp32 keeps that change of summation order at rounding level
    m = _model(torch.float32, dev).eval()
    with torch.no_grad():
        got = unet_amd.sliding_window_inference(x, (256, 256), 4, m)
        acc = torch.zeros_like(x, dtype=torch.float64)
        cnt = torch.zeros_like(x, dtype=torch.float64)
        for (sy, sx) in starts:
            for b in range(x.shape[0]):
                y = m(x[b:b + 1, :, sy:sy + 256, sx:sx + 256].contiguous())
                acc[b:b + 1, :, sy:sy + 256, sx:sx + 256] += y.double()
                cnt[b:b + 1, :, sy:sy + 256, sx:sx + 256] += 1
    assert float(cnt.min()) == 1 and float(cnt.max()) == 4          # starts 0, 192, 256: at most two windows per axis
    ref = (acc / cnt).float()
    assert float((got - ref).abs().max()) <= 2e-5 * float(ref.abs().max())


def test_directional_derivative_matches_gradient_fp32(dev):
    m = _model(torch.float32, dev).eval()            # eval: no dropout; norm statistics are per sample anyway
    x, t = _batch(2, 256, dev)
    loss = unet_amd.DiceCELoss(sigmoid=True, fold_label_two=True)
    l0 = loss(m(x), t); l0.backward()
    # parameters are views of the flat buffer, in a different order than parameters(): build the direction per tensor
    gen = torch.Generator().manual_seed(9)
    dirs = [torch.randn(p.shape, generator=gen).to(dev) * p.detach().abs().mean().clamp(min=1e-3) for p in m.parameters()]
    gd = sum(float((p.grad.double() * d.double()).sum()) for p, d in zip(m.parameters(), dirs))
    eps = 1e-2
    with torch.no_grad():
        for p, d in zip(m.parameters(), dirs):
            p.add_(eps * d)
        lp = float(loss(m(x), t))
        for p, d in zip(m.parameters(), dirs):
            p.sub_(2 * eps * d)
        lm = float(loss(m(x), t))
    fd = (lp - lm) / (2 * eps)
    assert abs(fd - gd) < 2e-2 * max(abs(gd), 1e-3), (fd, gd)


def test_fused_step_equals_autograd_step_fp32(dev):
    x, t = _batch(4, 128, dev)
    ma, mb = _model(torch.float32, dev, seed=4).train(), _model(torch.float32, dev, seed=4).train()
    loss = unet_amd.DiceCELoss(sigmoid=True, fold_label_two=True)
    oa = unet_amd.AdamW(ma.parameters(), lr=1e-3, weight_decay=1e-5, model=ma)
    ob = unet_amd.AdamW(mb.parameters(), lr=1e-3, weight_decay=1e-5, model=mb)
    tr = unet_amd.Trainer(ma, loss, oa)
    tr.step(x, t)
    ob.zero_grad(); loss(mb(x), t).backward()
    torch.cuda.synchronize()
    # same kernels, same order: the gradient the trainer fed to AdamW equals the autograd gradient
    ga = tr._grad
    for (name, shape, n, o, _), p in zip(mb._ptable_ordered, mb.parameters()):
        d = float((ga[o:o + n] - p.grad.flatten()).abs().max())
        assert d <= 1e-6 * max(float(p.grad.abs().max()), 1e-6), name
    ob.step()
    torch.cuda.synchronize()
    # AdamW turns rounding-level gradients into +-lr steps: one step of lr = 1e-3
    assert float((ma.flat_parameters() - mb.flat_parameters()).abs().max()) < 2.5e-3


@pytest.mark.parametrize("dt,drop,nb,size", [(torch.float32, 0.0, 4, 128), (torch.bfloat16, 0.1, 4, 128), (torch.bfloat16, 0.1, 16, 512)],
                         ids=["fp32", "bf16_dropout", "bf16_dropout_large_batch"])
def test_one_call_step_equals_the_split_calls_bit_for_bit(dev, dt, drop, nb, size):
    """mi_unet_train_step (one library call per step) against forward + loss + backward + AdamW as separate calls:
    same kernels in the same order, so parameters, gradients, moments and the loss sum agree exactly over three steps.
    The one-call step clears the loss sums and backward statistics beside the forward pass, lets the loss backward
    launch write the loss value and, on a large batch, runs AdamW bucket by bucket behind each bucket's reduction."""
    x, t = _batch(nb, size, dev)

    def run(split):
        torch.manual_seed(4)
        m = unet_amd.UNet(2, 1, 1, CH, (2, 2, 2, 2), num_res_units=2, dropout=drop, compute_dtype=dt).to(dev).train()
        m.seed_dropout(31)
        o = unet_amd.AdamW(m.parameters(), lr=1e-3, weight_decay=1e-5, model=m)
        sched = torch.optim.lr_scheduler.StepLR(o, step_size=1, gamma=0.5)
        tr = unet_amd.Trainer(m, unet_amd.DiceCELoss(sigmoid=True, fold_label_two=True), o, split_calls=split)
        for _ in range(3):
            tr.step(x, t)
            sched.step()                       # the learning rate of the next step is read from the optimizer each time
        torch.cuda.synchronize()
        return m.flat_parameters().clone(), tr._grad.clone(), o._m.clone(), o._v.clone(), tr.loss_sum(), o._step

    a, b = run(False), run(True)
    for u, v in zip(a[:4], b[:4]):
        assert torch.equal(u, v)
    assert a[4] == b[4] and a[4] > 0 and a[5] == b[5] == 3


def test_3d_patch_permutation_and_direction_full_size(dev):
    """The 3-D benchmark shape (4 patches of 96^3, channels 16-256): permuting the batch permutes the logits bit
    for bit (bf16), and the fp32 gradient agrees with a central difference of the loss along a random direction."""
    torch.manual_seed(0)
    mk = lambda dt: unet_amd.UNet(3, 1, 1, CH, (2, 2, 2, 2), num_res_units=2, compute_dtype=dt).to(dev)
    g = torch.Generator().manual_seed(5)
    x = torch.rand(4, 1, 96, 96, 96, generator=g).to(dev)
    t = disks(4, (96, 96, 96), 6, with_two=True).to(dev)
    loss = unet_amd.DiceCELoss(sigmoid=True, fold_label_two=True)
    m = mk(torch.bfloat16).train()
    perm = torch.tensor([2, 0, 3, 1], device=dev)
    with torch.no_grad():
        y = m(x); yp = m(x[perm])
    assert torch.equal(yp, y[perm])
    del m, y, yp
    m = mk(torch.float32).eval()
    l0 = loss(m(x), t); l0.backward()
    ps = list(m.parameters())
    dirs = [torch.randn(p.shape, generator=g).to(dev) * p.detach().abs().mean().clamp(min=1e-3) for p in ps]
    dot = sum(float((p.grad.double() * d.double()).sum()) for p, d in zip(ps, dirs))
    eps = 1e-2
    vals = []
    with torch.no_grad():
        for sgn in (1.0, -1.0):
            for p, d in zip(ps, dirs): p.add_(sgn * eps * d)
            vals.append(float(loss(m(x), t).double()))
            for p, d in zip(ps, dirs): p.sub_(sgn * eps * d)
    fd = (vals[0] - vals[1]) / (2 * eps)
    assert abs(fd - dot) < 2e-2 * max(abs(dot), 1e-3), (fd, dot)
