"""Does the chip run two half-batch chains side by side faster than one full-batch chain?  Two independent trainers of
half the batch, each on a stream of its own, stepped alternately without synchronisation, against one trainer on the
whole batch.  usage: python scripts/split_probe.py batch size dims [parts]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import unet_amd  # noqa: E402


def make(batch, size, dims, dev):
    torch.manual_seed(0)
    ch = (16, 32, 64, 128, 256)
    model = unet_amd.UNet(dims, 1, 1, ch, (2, 2, 2, 2), num_res_units=2, dropout=0.1, compute_dtype=torch.bfloat16).to(dev).train()
    tr = unet_amd.Trainer(model, unet_amd.DiceLoss(sigmoid=True, fold_label_two=True),
                          unet_amd.AdamW(model.parameters(), lr=5e-4, weight_decay=1e-5, model=model))
    x = torch.randn(batch, 1, *([size] * dims), device=dev)
    t = (torch.rand_like(x) > 0.5).float()
    return tr, x, t


def main():
    batch, size, dims = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    parts = int(sys.argv[4]) if len(sys.argv) > 4 else 2
    dev = torch.device("cuda", 0)
    n = 60
    tr, x, t = make(batch, size, dims, dev)
    for _ in range(10):
        tr.step(x, t)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        tr.step(x, t)
    th = time.perf_counter() - t0
    torch.cuda.synchronize()
    one = (time.perf_counter() - t0) / n
    print(f"one chain, batch {batch}: {1e3 * one:.3f} ms/step (host {1e3 * th / n:.3f})")
    del tr
    trs = [make(batch // parts, size, dims, dev) for _ in range(parts)]
    streams = [torch.cuda.Stream(dev) for _ in range(parts)]
    for _ in range(10):
        for (tr, x, t), s in zip(trs, streams):
            with torch.cuda.stream(s):
                tr.step(x, t)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        for (tr, x, t), s in zip(trs, streams):
            with torch.cuda.stream(s):
                tr.step(x, t)
    th = time.perf_counter() - t0
    torch.cuda.synchronize()
    two = (time.perf_counter() - t0) / n
    print(f"{parts} chains of batch {batch // parts} on {parts} streams: {1e3 * two:.3f} ms per step of the whole batch (host {1e3 * th / n:.3f}): x{one / two:.2f}")


if __name__ == "__main__":
    main()
