"""Where the host time of one Trainer.step goes: cProfile over 200 steps of batch 8 x 256^2 (no synchronisation inside).
usage: python scripts/host_profile.py [batch] [size]"""
import cProfile
import os
import pstats
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import unet_amd  # noqa: E402


def main():
    batch = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    size = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    model = unet_amd.UNet(2, 1, 1, (16, 32, 64, 128, 256), (2, 2, 2, 2), num_res_units=2, dropout=0.1,
                          compute_dtype=torch.bfloat16).to(dev).train()
    tr = unet_amd.Trainer(model, unet_amd.DiceLoss(sigmoid=True, fold_label_two=True),
                          unet_amd.AdamW(model.parameters(), lr=5e-4, weight_decay=1e-5, model=model))
    x = torch.randn(batch, 1, size, size, device=dev)
    t = (torch.rand_like(x) > 0.5).float()
    for _ in range(20):
        tr.step(x, t)
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(200):
        tr.step(x, t)
    pr.disable()
    torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(18)


if __name__ == "__main__":
    main()
