"""Host time to enqueue one training step against the GPU time of the step (no synchronisation inside the loop: when
the first figure is the larger one the host is what bounds the step).
usage: python scripts/host_enqueue.py [batch] [size] [dims] [channels, e.g. 32,64,128,256,512]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import unet_amd  # noqa: E402


def main():
    batch = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    size = int(sys.argv[2]) if len(sys.argv) > 2 else 512
    dims = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    channels = tuple(int(c) for c in sys.argv[4].split(",")) if len(sys.argv) > 4 else (16, 32, 64, 128, 256)
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    model = unet_amd.UNet(dims, 1, 1, channels, (2,) * (len(channels) - 1), num_res_units=2, dropout=0.1,
                          compute_dtype=torch.bfloat16).to(dev).train()
    loss = unet_amd.DiceLoss(sigmoid=True, fold_label_two=True)
    opt = unet_amd.AdamW(model.parameters(), lr=5e-4, weight_decay=1e-5, model=model)
    tr = unet_amd.Trainer(model, loss, opt)
    x = torch.randn(batch, 1, *([size] * dims), device=dev)
    t = (torch.rand_like(x) > 0.5).float()
    for _ in range(10):
        tr.step(x, t)
    torch.cuda.synchronize()
    n = 50
    t0 = time.perf_counter()
    for _ in range(n):
        tr.step(x, t)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"batch {batch} size {size} dims {dims} channels {channels[0]}-{channels[-1]}: host enqueue {1e3 * (t1 - t0) / n:.3f} ms/step, "
          f"enqueue + drain {1e3 * (t2 - t0) / n:.3f} ms/step")


if __name__ == "__main__":
    main()
