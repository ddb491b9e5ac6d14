"""Repeats forward+backward of one configuration and reports, per parameter, the worst gradient error
over the repeats (debug aid for run-to-run variation)."""
import sys, os, math
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
import oracle, unet_amd
from test_unet_gpu import make_pair, disks

def main():
    dims = int(sys.argv[1]); channels = tuple(int(v) for v in sys.argv[2].split(","))
    sp = tuple(int(v) for v in sys.argv[3].split(",")); nb = int(sys.argv[4]); reps = int(sys.argv[5])
    dev = torch.device("cuda", 0)
    o, m = make_pair(dims, channels, 2, 0.0, torch.float32, dev)
    o.train(); m.train()
    g = torch.Generator().manual_seed(1234)
    x = torch.rand(nb, 1, *sp, generator=g); t = disks(nb, sp, 77)
    lo = oracle.DiceCELoss(sigmoid=True); lm = unet_amd.DiceCELoss(sigmoid=True)
    o.zero_grad(); lo(o(x), t).backward()
    ref = [p.grad.clone() for p in o.parameters()]
    names = [n for n, _ in o.named_parameters()]
    worst = [0.0] * len(ref); first = None
    xg, tg = x.to(dev), t.to(dev)
    for r in range(reps):
        m.zero_grad()
        lm(m(xg), tg).backward()
        torch.cuda.synchronize()
        errs = [float((p.grad.cpu() - gr).abs().max()) / max(float(gr.abs().max()), 1e-30) for p, gr in zip(m.parameters(), ref)]
        big = [i for i, e in enumerate(errs) if e > 1e-4 and not names[i].endswith("bias")]
        if big and first is None:
            first = r
            print("first bad repeat", r)
            for i, e in enumerate(errs):
                if not names[i].endswith("conv.bias"):
                    print(f"   {names[i]:75s} {e:.3e}")
        worst = [max(a, b) for a, b in zip(worst, errs)]
        if big:
            print("repeat", r, "bad params:", len(big))
    print("worst non-bias err", max(e for n, e in zip(names, worst) if not n.endswith("bias")))
main()
