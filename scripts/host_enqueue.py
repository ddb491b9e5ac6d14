"""Host time to enqueue one training step (no synchronisation inside the timed loop) against the GPU time of a step:
python scripts/host_enqueue.py [--dims 3 --size 96 --batch 4]"""
import argparse, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import unet_amd

ap = argparse.ArgumentParser()
ap.add_argument("--dims", type=int, default=2)
ap.add_argument("--size", type=int, default=512)
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--steps", type=int, default=20)
a = ap.parse_args()
dev = torch.device("cuda", 0)
m = unet_amd.UNet(a.dims, 1, 1, (16, 32, 64, 128, 256), (2, 2, 2, 2), num_res_units=2, dropout=0.1, compute_dtype=torch.bfloat16).to(dev).train()
opt = unet_amd.AdamW(m.parameters(), lr=1e-3, weight_decay=1e-5, model=m)
tr = unet_amd.Trainer(m, unet_amd.DiceCELoss(sigmoid=True), opt)
shape = (a.batch, 1) + (a.size,) * a.dims
x = torch.rand(shape, device=dev)
t = (torch.rand(shape, device=dev) > 0.6).float()
for _ in range(5):
    tr.step(x, t)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.steps):
    tr.step(x, t)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host enqueue {1e3 * (t1 - t0) / a.steps:.3f} ms/step, with the GPU {1e3 * (t2 - t0) / a.steps:.3f} ms/step")
