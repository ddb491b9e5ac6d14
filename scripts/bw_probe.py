"""Measures what plain streaming kernels reach on this GPU (torch copy / add / fill / sum) for calibration."""
import torch
dev = torch.device("cuda", 0)
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e-3
for mb in (16, 64, 256, 1024):
    n = mb * 1024 * 1024 // 2
    x = torch.randn(n, device=dev, dtype=torch.bfloat16); y = torch.empty_like(x); z = torch.randn_like(x)
    tc = t(lambda: y.copy_(x)); ta = t(lambda: torch.add(x, z, out=y)); tf = t(lambda: y.fill_(1.0)); ts = t(lambda: x.float().sum() if False else torch.sum(x, dtype=torch.float32))
    print(f"{mb:5d} MB: copy {2*mb/1024/tc/1e3*1.0737:6.2f} TB/s  add(2r1w) {3*mb/1024/ta/1e3*1.0737:6.2f} TB/s  fill {mb/1024/tf/1e3*1.0737:6.2f} TB/s  sum {mb/1024/ts/1e3*1.0737:6.2f} TB/s", flush=True)
