"""Times single conv launches (forward, data gradient, weight gradient) at the layer shapes of the benchmark
network, through the operator-level C ABI.  usage: python scripts/bench_ops.py [filter-substring]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unet_amd import _lib, ops  # noqa: E402

LAYERS = [  # name, cin, cout, H_in, stride, transposed, ksize
    ("c16 s1 256", 16, 16, 256, 1, False, 3),
    ("c32 s1 128", 32, 32, 128, 1, False, 3),
    ("c64 s1 64", 64, 64, 64, 1, False, 3),
    ("c128 s1 32", 128, 128, 32, 1, False, 3),
    ("c128-256 s1 32", 128, 256, 32, 1, False, 3),
    ("c256 s1 32", 256, 256, 32, 1, False, 3),
    ("c16-64 s2 256", 16, 64, 256, 2, False, 3),
    ("c32-128 s2 128", 32, 128, 128, 2, False, 3),
    ("c64-256 s2 64", 64, 256, 64, 2, False, 3),
    ("c384-64 T 32", 384, 64, 32, 2, True, 3),
    ("c128-32 T 64", 128, 32, 64, 2, True, 3),
    ("c64-16 T 128", 64, 16, 128, 2, True, 3),
]


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3  # us


def main():
    flt = sys.argv[1] if len(sys.argv) > 1 else ""
    nb = int(os.environ.get("NB", "32"))
    dt = torch.bfloat16
    dev = torch.device("cuda", 0)
    lib = _lib.load()
    st = _lib.current_stream(dev)
    print(f"{'layer':18s} {'fwd us':>9s} {'TF/s':>7s} {'GB/s':>7s} | {'dgrad us':>9s} {'TF/s':>7s} {'GB/s':>7s} | {'wgrad us':>9s} {'TF/s':>7s} {'GB/s':>7s}")
    for name, cin, cout, h, stride, tr, k in LAYERS:
        if flt not in name:
            continue
        ho = 2 * h if tr else h // stride
        x = torch.randn(nb, h, h, cin, device=dev).to(dt)
        y = torch.empty(nb, ho, ho, cout, device=dev, dtype=dt)
        dy = torch.randn(nb, ho, ho, cout, device=dev).to(dt)
        dx = torch.empty_like(x)
        w = torch.randn((cin, cout, k, k) if tr else (cout, cin, k, k), device=dev) * 0.05
        d = ops._desc(x.shape, cin, cout, k, stride, tr, dt)
        pf = ops.pack_weights(w, x.shape, k, stride, tr, dt, False)
        pd = ops.pack_weights(w, x.shape, k, stride, tr, dt, True)
        stats = torch.zeros(nb, 8, cout, 2, dtype=torch.float64, device=dev)
        ws = torch.empty(lib.mi_conv_wgrad_workspace_bytes(C.byref(d)), dtype=torch.uint8, device=dev)
        dw = torch.empty_like(w)
        tx, ty, tdy, tdx = ops._ten(x), ops._ten(y), ops._ten(dy), ops._ten(dx)

        def f_fwd():
            _lib.check(lib.mi_conv_forward(C.byref(d), C.byref(tx), pf.data_ptr(), None, C.byref(ty), None, stats.data_ptr(), st), "fwd")

        def f_dg():
            _lib.check(lib.mi_conv_dgrad(C.byref(d), C.byref(tdy), pd.data_ptr(), C.byref(tdx), None, st), "dgrad")

        def f_wg():
            _lib.check(lib.mi_conv_wgrad(C.byref(d), C.byref(tx), C.byref(tdy), dw.data_ptr(), None, ws.data_ptr(), st), "wgrad")

        flops = 2.0 * nb * (h * h if tr else ho * ho) * k * k * cin * cout
        byts = 2.0 * (x.numel() + y.numel())
        out = f"{name:18s}"
        for fn in (f_fwd, f_dg, f_wg):
            us = timeit(fn)
            out += f" {us:9.1f} {flops / us / 1e6:7.1f} {byts / us / 1e3:7.1f} |"
        print(out, flush=True)


if __name__ == "__main__":
    main()
