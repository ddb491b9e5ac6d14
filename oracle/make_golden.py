"""Writes the golden vectors under tests/golden/ from the CPU oracle.

    python -m oracle.make_golden

The reference holds no fixtures for this path and MONAI is not installed, so these vectors are
outputs of the oracle itself (the restatement in this package), not of the reference: they pin the
HIP path and the oracle against regressions of either side.

Files (each < 1 MB):
  unet2d_small.npz   2-D (16,32,64) network: every parameter, input, labels, logits, DiceCE loss, every gradient
  unet2d_16_256.npz  2-D (16,32,64,128,256) network (reference/train.py:78-86) built from seed 0: input, labels,
                     logits, loss, a checksum of the weights, and for every gradient tensor its sum, norm, maximum and
                     up to 256 elements at fixed pseudo-random positions (fp64 oracle).  The input is the first of a
                     seed sequence for which no PReLU input of the fp64 oracle lies within 2e-5 of zero, so an fp32
                     implementation takes the same side of every kink and single gradient elements can be compared
  unet3d_small.npz   3-D (16,32,64) network from seed 0: same content as the 16-256 file
  closed_forms.npz   loss / metric / sliding-window cases with analytically known answers
"""
from __future__ import annotations

import os

import numpy as np
import torch

import oracle

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def inputs(shape, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(*shape, generator=g)
    t = (torch.rand(*shape, generator=g) > 0.7).float()
    return x, t


def run(model, x, t):
    model.eval()
    model.zero_grad()
    y = model(x)
    loss = oracle.DiceCELoss(to_onehot_y=False, sigmoid=True)(y, t)
    loss.backward()
    return y.detach(), float(loss.detach())


def weight_checksum(model):
    """Order-sensitive fp64 checksum of all parameters: detects a change of torch's RNG streams or
    of the construction order, either of which would make the seed-built fixtures meaningless."""
    acc = 0.0
    for i, (_, p) in enumerate(model.named_parameters()):
        v = p.detach().double().flatten()
        w = torch.arange(1, v.numel() + 1, dtype=torch.float64)
        acc += float((v * torch.cos(w * 0.001 + i)).sum())
    return acc


def grad_summary(model):
    names, sums, norms, firsts = [], [], [], []
    for n, p in model.named_parameters():
        g = p.grad.detach().double().flatten()
        names.append(n)
        sums.append(float(g.sum()))
        norms.append(float(g.norm()))
        firsts.append(g[:4].numpy() if g.numel() >= 4 else np.pad(g.numpy(), (0, 4 - g.numel())))
    return np.array(names), np.array(sums), np.array(norms), np.stack(firsts)


KINK_MARGIN = 2e-5      # smallest |PReLU input| accepted for a seed-built fixture (normalised units)
SAMPLES = 256           # gradient elements stored per tensor


def sample_index(i, numel):
    """Positions of the stored elements of gradient tensor number i (shared with the test through this rule:
    numpy RandomState(1000 + i), a sorted draw without replacement)."""
    if numel <= SAMPLES:
        return np.arange(numel)
    return np.sort(np.random.RandomState(1000 + i).choice(numel, SAMPLES, replace=False))


def seeded(dims, channels, shape, fname):
    from oracle.inject import record_preactivations

    torch.manual_seed(0)
    m = oracle.UNet(dims, 1, 1, channels, (2,) * (len(channels) - 1), num_res_units=2, dropout=0.0)
    m.eval()
    m64 = oracle.UNet(dims, 1, 1, channels, (2,) * (len(channels) - 1), num_res_units=2, dropout=0.0).double()
    m64.load_state_dict({k: v.double() for k, v in m.state_dict().items()})
    m64.eval()
    for seed in range(1, 2000):
        x, t = inputs(shape, seed)
        margin = min(float(d.abs().min()) for d in record_preactivations(m64, x.double()))
        if margin > KINK_MARGIN:
            break
    else:
        raise RuntimeError("no kink-safe input found")
    y, loss = run(m64, x.double(), t.double())
    names, sums, norms, firsts = grad_summary(m64)
    samples = np.zeros((len(names), SAMPLES))
    maxima = np.zeros(len(names))
    for i, (_, p) in enumerate(m64.named_parameters()):
        g = p.grad.detach().flatten().numpy()
        idx = sample_index(i, g.size)
        samples[i, :idx.size] = g[idx]
        maxima[i] = np.abs(g).max()
    print(f"{fname}: input seed {seed}, kink margin {margin:.3e}")
    np.savez_compressed(os.path.join(OUT, fname), x=x.numpy(), t=t.numpy(), logits=y.float().numpy(), loss=np.float64(loss),
                        grad_names=names, grad_sums=sums, grad_norms=norms, grad_firsts=firsts, grad_samples=samples,
                        grad_max=maxima, input_seed=np.int64(seed), kink_margin=np.float64(margin),
                        weight_checksum=np.float64(weight_checksum(m)), channels=np.array(channels), dims=np.int64(dims))


def main():
    os.makedirs(OUT, exist_ok=True)
    # ---- small 2-D network with everything stored ------------------------------------------------
    torch.manual_seed(0)
    m = oracle.UNet(2, 1, 1, (16, 32, 64), (2, 2), num_res_units=2, dropout=0.0)
    x, t = inputs((2, 1, 32, 32), 1)
    y, loss = run(m, x, t)
    blob = {"x": x.numpy(), "t": t.numpy(), "logits": y.numpy(), "loss": np.float64(loss)}
    for n, p in m.named_parameters():
        blob["param:" + n] = p.detach().numpy()
        blob["grad:" + n] = p.grad.detach().numpy()
    np.savez_compressed(os.path.join(OUT, "unet2d_small.npz"), **blob)
    # ---- seed-built networks ----------------------------------------------------------------------
    seeded(2, (16, 32, 64, 128, 256), (2, 1, 64, 64), "unet2d_16_256.npz")
    seeded(3, (16, 32, 64), (1, 1, 32, 32, 32), "unet3d_small.npz")
    # ---- closed forms -------------------------------------------------------------------------------
    big = 30.0
    tt = torch.zeros(1, 1, 8, 8)
    tt[..., :4] = 1
    perfect = (2 * tt - 1) * big
    l_perfect = float(oracle.DiceCELoss(sigmoid=True)(perfect, tt))
    l_zero = float(oracle.DiceCELoss(sigmoid=True)(torch.zeros_like(tt), tt))
    l_wrong = float(oracle.DiceCELoss(sigmoid=True)(-perfect, tt))
    np.savez_compressed(os.path.join(OUT, "closed_forms.npz"), target=tt.numpy(), logit_scale=np.float64(big),
                        loss_perfect=np.float64(l_perfect), loss_zero_logits=np.float64(l_zero),
                        loss_inverted=np.float64(l_wrong))
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()
