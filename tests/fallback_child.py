"""Child process of tests/test_fallbacks_gpu.py: one bf16 forward + loss + backward of a small U-Net (2-D, 3-D with CHILD_3D=1, a larger 2-D batch with CHILD_BIG=1) under whatever
kernel-family switches the environment holds (they are read once per process), results saved with torch.save."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import unet_amd  # noqa: E402


def main(out_path: str) -> None:
    dev = torch.device("cuda", 0)
    torch.manual_seed(5)
    channels = (16, 32, 64, 128)
    dt = torch.float32 if os.environ.get("CHILD_FP32") else torch.bfloat16
    m = unet_amd.UNet(2, 1, 1, channels, (2, 2, 2), num_res_units=2, dropout=0.1, compute_dtype=dt)
    m.to(dev).train()
    m.seed_dropout(7)
    g = torch.Generator().manual_seed(1)
    shape = (2, 1, 96, 160)
    if os.environ.get("CHILD_3D"):       # the 3-D kernel families: a 16-128 network on two 32 x 48 x 64 volumes
        m = unet_amd.UNet(3, 1, 1, channels, (2, 2, 2), num_res_units=2, dropout=0.1, compute_dtype=dt).to(dev).train()
        m.seed_dropout(7)
        shape = (2, 1, 32, 48, 64)
    if os.environ.get("CHILD_BIG"):      # a batch at which the large-workload schedule is chosen (16 slices of 256 x 256)
        shape = (16, 1, 256, 256)
    x = torch.rand(*shape, generator=g).to(dev)
    t = (torch.rand(*shape, generator=g) > 0.7).float().to(dev)
    y = m(x)
    loss = unet_amd.DiceCELoss(sigmoid=True)(y, t)
    loss.backward()
    torch.cuda.synchronize()
    grads = {k: p.grad.detach().float().cpu() for k, p in m.named_parameters()}
    torch.save({"logits": y.detach().float().cpu(), "loss": float(loss), "grads": grads}, out_path)


if __name__ == "__main__":
    main(sys.argv[1])
