"""Runs single operators repeatedly on identical inputs and reports run-to-run differences (debug aid)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from unet_amd import ops
from helpers import to_cl

dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(0)
nb, c, sp = 2, 16, (32, 32)
x = torch.randn(nb, c, *sp, generator=g); dy = torch.randn(nb, c, *sp, generator=g)
w = torch.randn(c, c, 3, 3, generator=g) * 0.1
for dt in (torch.float32, torch.bfloat16):
    xc, dyc = to_cl(x, dt, dev), to_cl(dy, dt, dev)
    s1 = x.flatten(2).sum(-1); s2 = (x * x).flatten(2).sum(-1)
    stats = torch.zeros(nb, 8, c, 2, dtype=torch.float64); stats[:, 0, :, 0], stats[:, 0, :, 1] = s1, s2
    stats = stats.to(dev); alpha = torch.tensor([0.25], device=dev)
    ref = None; nbad = 0; worst = 0.0
    for r in range(100):
        dx, da = ops.instnorm_prelu_bwd(dyc, xc, stats, alpha)
        dx = dx.float()
        if ref is None: ref = dx.clone()
        d = float((dx - ref).abs().max())
        worst = max(worst, d); nbad += d > 1e-4
    print(dt, "norm bwd   : worst run-to-run diff", worst, "bad", nbad)
    ref = None; nbad = 0; worst = 0.0
    for r in range(100):
        y, st = ops.conv_forward(xc, w.to(dev), None, want_stats=True)
        v = torch.cat([y.float().flatten(), st.sum(1).flatten()])
        if ref is None: ref = v.clone()
        d = float(((v - ref).abs() / (ref.abs() + 1)).max())
        worst = max(worst, d); nbad += d > 1e-4
    print(dt, "conv fwd   : worst", worst, "bad", nbad)
    ref = None; nbad = 0; worst = 0.0
    for r in range(100):
        dw, db = ops.conv_wgrad(xc, dyc, (c, c, 3, 3))
        v = torch.cat([dw.flatten(), db.flatten()])
        if ref is None: ref = v.clone()
        d = float(((v - ref).abs() / (ref.abs() + 1)).max())
        worst = max(worst, d); nbad += d > 1e-4
    print(dt, "conv wgrad : worst", worst, "bad", nbad)
    ref = None; nbad = 0; worst = 0.0
    for r in range(100):
        dx = ops.conv_dgrad(dyc, w.to(dev), xc.shape, addend=xc).float()
        if ref is None: ref = dx.clone()
        d = float((dx - ref).abs().max())
        worst = max(worst, d); nbad += d > 1e-4
    print(dt, "conv dgrad : worst", worst, "bad", nbad)
