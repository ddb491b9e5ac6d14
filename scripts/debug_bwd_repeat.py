"""Runs ONE forward, then the decoder bucket of the backward repeatedly on the same workspace and
reports which step outputs change between repeats (debug aid)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
import unet_amd
from test_unet_gpu import make_pair, disks

dev = torch.device("cuda", 0)
dims, channels, sp, nb = 2, (16, 32, 64, 128, 256), (64, 64), 2
o, m = make_pair(dims, channels, 2, 0.0, torch.float32, dev)
m.train()
g = torch.Generator().manual_seed(1234)
x = torch.rand(nb, 1, *sp, generator=g).to(dev)
dl = (torch.randn(nb, 1, *sp, generator=g) * 1e-4).to(dev)
flat = m._ensure_flat(dev)
plan = m._plan_for(x, True)
ws = plan.take_workspace(dev)
logits = torch.empty_like(x)
plan.forward(flat, x, logits, ws, True, 1)
grads = torch.zeros_like(flat)
rows = []
lib = plan.lib
for i, s in enumerate(plan.steps(True)):
    rows.append((i, s["layer"], s["kernel"], lib.mi_unet_step_output_offset(plan.h, 1, i)))
offs = sorted(set(r[3] for r in rows if r[3] >= 0)) + [plan.workspace_bytes]
ref = None
for r in range(30):
    plan.backward(flat, x, dl, grads, ws, True, 1, 0)
    torch.cuda.synchronize()
    snap = ws.clone().view(torch.float32)
    if ref is None:
        ref = snap; continue
    out = []
    for i, layer, kern, off in rows:
        if off < 0: continue
        end = offs[offs.index(off) + 1]
        a, b = ref[off // 4:end // 4], snap[off // 4:end // 4]
        fin = torch.isfinite(a) & torch.isfinite(b)
        d = (a[fin] - b[fin]).abs()
        sc = a[fin].abs().max().item() if fin.any() else 0.0
        if d.numel() and d.max().item() > 1e-4 * sc:
            out.append((i, layer, d.max().item() / max(sc, 1e-30), int((d > 1e-4 * sc).sum())))
    if out:
        print("repeat", r, "differs:")
        for e in out[:6]: print("    step %d %s rel %.3e n %d" % e)
print("done")
