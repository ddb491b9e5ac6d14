"""Element-wise oracle parity at the benchmark's own sizes (BASELINE.json configs[1]: 512 x 512 slices; configs[3]:
96^3 patches), reference/train.py:115-118.  Multi-tile walks per workgroup, the eight statistics slots, the 192 / 256
workgroup weight-gradient split and the slab reduction only take their production shape at these sizes.

fp32 engine: logits 1e-4, loss 1e-5 and EVERY element of all 63 gradient tensors within 1e-4 of its tensor's maximum
against the fp64 CPU oracle, with the engine's PReLU decisions and dropout masks injected into the oracle
(tests/test_grad_parity_gpu.py explains the method and asserts that the injected decisions differ from the oracle's
own only within rounding distance of the kink).  bf16 engine: the bounds of test_bf16_close_to_oracle."""
import pytest
import torch

from helpers import rel_l2
from test_grad_parity_gpu import test_every_gradient_element_matches_the_fp64_oracle as every_element
from test_unet_gpu import disks, make_pair, run_both

pytestmark = pytest.mark.gpu
CH = (16, 32, 64, 128, 256)

FULL = [
    (2, CH, 8, (512, 512), (0.25, 0.25), 0.1),       # train mode as reference/train.py:85
    (3, CH, 2, (96, 96, 96), (0.1, 0.4), 0.1),
]


@pytest.mark.parametrize("case", FULL, ids=["2d_512x512_b8", "3d_96x96x96_b2"])
def test_fp32_every_gradient_element_at_benchmark_size(dev, case):
    every_element(dev, case)


@pytest.mark.parametrize("dims,nb,sp", [(2, 8, (512, 512)), (3, 2, (96, 96, 96))], ids=["2d_512x512_b8", "3d_96x96x96_b2"])
def test_bf16_close_to_oracle_at_benchmark_size(dev, dims, nb, sp):
    o, m = make_pair(dims, CH, 2, 0.0, torch.bfloat16, dev)
    o.train(); m.train()
    g = torch.Generator().manual_seed(1234)
    x = torch.rand(nb, 1, *sp, generator=g)
    t = disks(nb, sp, 77)
    yo, lo, ym, lm = run_both(o, m, x, t, dev)
    assert rel_l2(ym, yo) < 3e-2, "logits"
    assert abs(float(lm) - float(lo)) < 2e-2, "loss"
    go = torch.cat([p.grad.flatten() for p in o.parameters()])
    gm = torch.cat([p.grad.flatten().cpu() for p in m.parameters()])
    assert rel_l2(gm, go) < 0.15, "gradients"
