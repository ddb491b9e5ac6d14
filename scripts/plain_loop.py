"""The reference's own loop (reference/train.py:111-119: zero_grad, model(x), loss, backward, step) on this package's
modules, against the fused Trainer.step on the same workload.  usage: python scripts/plain_loop.py [batch] [size]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import unet_amd  # noqa: E402


def main():
    batch = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    size = int(sys.argv[2]) if len(sys.argv) > 2 else 512
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    model = unet_amd.UNet(2, 1, 1, (16, 32, 64, 128, 256), (2, 2, 2, 2), num_res_units=2, dropout=0.1,
                          compute_dtype=torch.bfloat16).to(dev).train()
    loss_fn = unet_amd.DiceCELoss(to_onehot_y=False, sigmoid=True)
    opt = unet_amd.AdamW(model.parameters(), lr=5e-4, model=model)
    x = torch.randn(batch, 1, size, size, device=dev)
    t = (torch.rand_like(x) > 0.5).float()

    def step():
        opt.zero_grad()
        out = model(x)
        loss = loss_fn(out, t)
        loss.backward()
        opt.step()

    for _ in range(10):
        step()
    torch.cuda.synchronize()
    n = 40
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"plain loop, batch {batch} x {size}^2: host {1e3 * (t1 - t0) / n:.3f} ms/step, total {1e3 * (t2 - t0) / n:.3f} ms/step "
          f"({batch * n / (t2 - t0):.0f} slices/s)")


if __name__ == "__main__":
    main()
