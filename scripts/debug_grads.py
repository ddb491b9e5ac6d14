"""Prints per-parameter gradient errors of the HIP engine against the CPU oracle (debug aid)."""
import sys, os, math
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from test_unet_gpu import make_pair, disks, run_both

def main():
    dims = int(sys.argv[1]); channels = tuple(int(v) for v in sys.argv[2].split(","))
    sp = tuple(int(v) for v in sys.argv[3].split(",")); nb = int(sys.argv[4]); dt = sys.argv[5] if len(sys.argv) > 5 else "fp32"
    dev = torch.device("cuda", 0)
    if os.environ.get("POISON"):
        # fill freed allocator blocks with NaN so that any read of unwritten workspace shows up
        junk = [torch.full((1 << 28,), float("nan"), device=dev) for _ in range(6)]
        torch.cuda.synchronize(); del junk
    o, m = make_pair(dims, channels, 2, 0.0, torch.float32 if dt == "fp32" else torch.bfloat16, dev)
    o.train(); m.train()
    g = torch.Generator().manual_seed(1234)
    x = torch.rand(nb, 1, *sp, generator=g)
    t = disks(nb, sp, 77)
    yo, lo, ym, lm = run_both(o, m, x, t, dev)
    print("logits err", float((ym - yo).abs().max() / yo.abs().max()), "loss", float(lo), float(lm))
    gnorm = math.sqrt(sum(float(p.grad.double().pow(2).sum()) for p in o.parameters()))
    for (n, po), pm in zip(o.named_parameters(), m.parameters()):
        go, gm = po.grad, pm.grad.cpu()
        print(f"{n:75s} max|g| {float(go.abs().max()):.3e} err {float((gm-go).abs().max())/max(float(go.abs().max()),1e-30):.3e}")
main()
