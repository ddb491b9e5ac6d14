"""Element-wise gradient parity at the reference's PReLU slopes, in eval and in train mode (dropout 0.1 as
reference/train.py:85), fp32 HIP engine against the fp64 oracle.

A PReLU kink and a dropout mask are decisions, not arithmetic: two correct implementations may round an
activation to different sides of 0, and no hand-written kernel can reproduce torch's Philox stream.  The oracle
is therefore run with the decisions of the implementation under test (oracle/inject.py):

  * dropout: the HIP path's mask is a pure function of (seed, layer, image, element index); it is restated in
    numpy (oracle/dropout_ref.py) and multiplied in where ``nn.Dropout`` would draw;
  * PReLU: the side each activation took is recomputed from the raw convolution outputs and fp64 statistics
    the engine leaves in its workspace (``mi_unet_norm_info``).

The test then checks (1) that those branch decisions differ from the oracle's own only where the oracle's PReLU
input is within 1e-4 of zero, and how many such elements there are; (2) logits and loss at 1e-4 / 1e-5; (3)
EVERY element of all 63 gradient tensors within 1e-4 of its tensor's largest element (biases in front of an
instance norm, whose true gradient is 0, within 1e-6 of the global gradient norm; a PReLU slope gradient, one
scalar that is a signed sum over a whole tensor, within 1e-4 of the larger of its value and 1 % of the absolute
sum of its terms).  Nothing is masked out.
"""
import copy
import ctypes as C
import math

import numpy as np
import pytest
import torch

import oracle
import unet_amd
from oracle.dropout_ref import keep_mask
from oracle.inject import inject, record_preactivations
from test_unet_gpu import disks, make_pair
from unet_amd import _lib

pytestmark = pytest.mark.gpu


def hip_step(m, x, t, dev, training, seed):
    """Forward, DiceCE loss and backward through the plan API with a workspace the test keeps."""
    xg, tg = x.to(dev), t.to(dev)
    plan = m._plan_for(xg, True)
    flat = m._ensure_flat(dev)
    ws = torch.empty(plan.workspace_bytes, dtype=torch.uint8, device=dev)
    logits = torch.empty_like(xg)
    plan.forward(flat, xg, logits, ws, training, seed)
    lg = logits.clone().requires_grad_()
    loss = unet_amd.DiceCELoss(sigmoid=True)(lg, tg)
    loss.backward()
    g = torch.zeros_like(flat)
    plan.backward(flat, xg, lg.grad.contiguous(), g, ws, training, seed)
    torch.cuda.synchronize()
    grads = {n: g[o:o + k].view(s).cpu() for (n, s, k, o, _) in m._ptable}
    return logits.cpu(), float(loss), grads, ws, plan


def hip_decisions(plan, ws, shapes, keep):
    """True where the HIP forward took the identity side of the PReLU, per norm block, as [N, C, *spatial]."""
    lib = plan.lib
    n = lib.mi_unet_num_norms(plan.h)
    assert n == len(shapes)
    out = []
    for k in range(n):
        info = _lib.MiNormInfo()
        _lib.check(lib.mi_unet_norm_info(plan.h, k, C.byref(info)))
        nb, c = shapes[k][0], shapes[k][1]
        assert info.c == c and info.npix == math.prod(shapes[k][2:]) and info.layer == k
        esz, tdt = (2, torch.bfloat16) if info.dtype else (4, torch.float32)
        raw = ws[info.raw_offset:info.raw_offset + nb * info.npix * info.pitch * esz].view(tdt)
        raw = raw.view(nb, info.npix, info.pitch)[:, :, info.coff:info.coff + c].float()
        st = ws[info.stats_offset:info.stats_offset + nb * 8 * c * 16].view(torch.float64).view(nb, 8, c, 2)
        s1 = torch.zeros(nb, c, dtype=torch.float64, device=ws.device)
        for sl in range(8):                         # the kernels add the slots in this order
            s1 = s1 + st[:, sl, :, 0]
        mean = (s1 / float(info.npix)).float()      # (x - mean) * rstd > 0  <=>  x > mean
        pos = (raw > mean[:, None, :]).permute(0, 2, 1).contiguous().reshape(shapes[k]).cpu()      # channels first, dense
        if keep is not None:
            pos &= keep[k]                          # a dropped element is 0: the slope side
        out.append(pos)
    return out


CASES = [
    # dims, channels, batch, spatial, slopes, dropout
    (2, (16, 32, 64, 128, 256), 2, (64, 64), (0.25, 0.25), 0.0),
    (2, (16, 32, 64, 128, 256), 2, (64, 64), (0.1, 0.4), 0.0),
    (2, (16, 32, 64, 128, 256), 2, (64, 64), (0.25, 0.25), 0.1),
    (3, (16, 32, 64, 128, 256), 1, (32, 32, 32), (0.25, 0.25), 0.0),
    (3, (16, 32, 64, 128, 256), 1, (32, 32, 32), (0.1, 0.4), 0.1),
    (2, (16, 32, 64), 3, (48, 80), (0.25, 0.25), 0.1),           # ragged tiles, odd batch
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "%dd_%s_b%d_%s_slope%s_drop%s" % (
    c[0], "-".join(map(str, c[1])), c[2], "x".join(map(str, c[3])), c[4][0] if c[4][0] == c[4][1] else "rand", c[5]))
def test_every_gradient_element_matches_the_fp64_oracle(dev, case):
    dims, channels, nb, sp, slopes, p = case
    training = p > 0
    o, m = make_pair(dims, channels, 2, p, torch.float32, dev, seed=3, slopes=slopes)
    g = torch.Generator().manual_seed(4321)
    x = torch.rand(nb, 1, *sp, generator=g)
    t = disks(nb, sp, 78)
    seed = 0x1234ABCD5678
    m.train(training)
    ym, loss_m, grads_m, ws, plan = hip_step(m, x, t, dev, training, seed)

    o64 = copy.deepcopy(o).double()
    o64.train(training)
    x64, t64 = x.double(), t.double()
    keep = None
    # shapes of the PReLU inputs, then the masks of this (seed, layer, image)
    o64.eval()
    shapes = [tuple(d.shape) for d in record_preactivations(o64, x64)]
    o64.train(training)
    if training:
        keep = [torch.from_numpy(keep_mask(seed, k, s, p)) for k, s in enumerate(shapes)]
        rate = sum(int((~k).sum()) for k in keep) / sum(k.numel() for k in keep)
        assert abs(rate - p) < 0.01, rate
    with inject(o64, x64, keep=keep, p=p):
        pre = record_preactivations(o64, x64)                   # the oracle's own PReLU inputs under this mask
    dec = hip_decisions(plan, ws, shapes, keep)
    flips, total = 0, 0
    for k, (d, h) in enumerate(zip(pre, dec)):
        diff = h != (d > 0)
        flips += int(diff.sum()); total += d.numel()
        if diff.any():
            assert float(d[diff].abs().max()) < 1e-4, f"block {k}: the HIP path took another PReLU branch at |input| = {float(d[diff].abs().max()):.3e}"
    assert flips <= max(4, total // 20000), (flips, total)

    mass = {}
    with inject(o64, x64, decisions=dec, keep=keep, p=p, slope_mass=mass):
        o64.zero_grad()
        yo = o64(x64)
        loss_o = oracle.DiceCELoss(sigmoid=True)(yo, t64)
        loss_o.backward()
    assert float((ym.double() - yo.detach()).abs().max() / yo.detach().abs().max()) < 1e-4, "logits"
    assert abs(loss_m - float(loss_o)) < 1e-5 * max(1.0, abs(float(loss_o))), "loss"

    gnorm = math.sqrt(sum(float(q.grad.pow(2).sum()) for q in o64.parameters()))
    names = {n for n, _ in o64.named_parameters()}
    worst = ("", 0.0)
    slope_of = {id(q): mod for mod in o64.modules() if isinstance(mod, torch.nn.PReLU) for q in mod.parameters()}
    for n, q in o64.named_parameters():
        go, gm = q.grad, grads_m[n].double()
        cancelled = n.endswith("conv.bias") and n.replace("conv.bias", "adn.A.weight") in names
        scale = 1e-2 * gnorm if cancelled else max(float(go.abs().max()), 1e-30)
        if id(q) in slope_of:
            # one scalar = a signed sum over a whole tensor: where the terms cancel, fp32 rounding of the terms is read
            # against 1 % of their absolute sum (an absolute error of 1e-6 of that sum) rather than against the result
            scale = max(scale, 1e-2 * mass[slope_of[id(q)]])
        err = float((gm - go).abs().max()) / scale
        if err > worst[1]:
            worst = (n, err)
        assert err < 1e-4, (n, err)
    print(f"PReLU decisions differing from the oracle's own: {flips} of {total}; worst gradient element {worst[1]:.2e} ({worst[0]})")
