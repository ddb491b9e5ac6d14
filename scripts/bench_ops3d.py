"""Times single 3-D conv launches (forward, data gradient) at the layer shapes of the 96^3 benchmark network
(batch 4), through the operator-level C ABI (no contraction split).  usage: python scripts/bench_ops3d.py [filter]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unet_amd import _lib, ops  # noqa: E402
from bench_ops import timeit  # noqa: E402

LAYERS = [  # name, cin, cout, D_in, stride, transposed, ksize
    ("c16 s1 48", 16, 16, 48, 1, False, 3),
    ("c32 s1 24", 32, 32, 24, 1, False, 3),
    ("c64 s1 12", 64, 64, 12, 1, False, 3),
    ("c128 s1 6", 128, 128, 6, 1, False, 3),
    ("c128-256 s1 6", 128, 256, 6, 1, False, 3),
    ("c256 s1 6", 256, 256, 6, 1, False, 3),
    ("c128-256 k1 6", 128, 256, 6, 1, False, 1),
    ("c16-64 s2 48", 16, 64, 48, 2, False, 3),
    ("c32-128 s2 24", 32, 128, 24, 2, False, 3),
    ("c64-256 s2 12", 64, 256, 12, 2, False, 3),
    ("c384-64 T 6", 384, 64, 6, 2, True, 3),
    ("c128-32 T 12", 128, 32, 12, 2, True, 3),
    ("c64-16 T 24", 64, 16, 24, 2, True, 3),
]


def main():
    flt = sys.argv[1] if len(sys.argv) > 1 else ""
    nb = int(os.environ.get("NB", "4"))
    dt = torch.bfloat16
    dev = torch.device("cuda", 0)
    lib = _lib.load()
    st = _lib.current_stream(dev)
    print(f"{'layer':18s} {'fwd us':>9s} {'TF/s':>7s} | {'dgrad us':>9s} {'TF/s':>7s}")
    for name, cin, cout, h, stride, tr, k in LAYERS:
        if flt not in name:
            continue
        ho = 2 * h if tr else h // stride
        x = torch.randn(nb, h, h, h, cin, device=dev).to(dt)
        y = torch.empty(nb, ho, ho, ho, cout, device=dev, dtype=dt)
        dy = torch.randn(nb, ho, ho, ho, cout, device=dev).to(dt)
        dx = torch.empty_like(x)
        w = torch.randn((cin, cout, k, k, k) if tr else (cout, cin, k, k, k), device=dev) * 0.05
        d = ops._desc(x.shape, cin, cout, k, stride, tr, dt)
        pf = ops.pack_weights(w, x.shape, k, stride, tr, dt, False)
        pd = ops.pack_weights(w, x.shape, k, stride, tr, dt, True)
        stats = torch.zeros(nb, 8, cout, 2, dtype=torch.float64, device=dev)
        tx, ty, tdy, tdx = ops._ten(x), ops._ten(y), ops._ten(dy), ops._ten(dx)

        def f_fwd():
            _lib.check(lib.mi_conv_forward(C.byref(d), C.byref(tx), pf.data_ptr(), None, C.byref(ty), None, stats.data_ptr(), st), "fwd")

        def f_dg():
            _lib.check(lib.mi_conv_dgrad(C.byref(d), C.byref(tdy), pd.data_ptr(), C.byref(tdx), None, st), "dgrad")

        flops = 2.0 * nb * (h ** 3 if tr else ho ** 3) * k ** 3 * cin * cout
        out = f"{name:18s}"
        for fn in (f_fwd, f_dg):
            us = timeit(fn)
            out += f" {us:9.1f} {flops / us / 1e6:7.1f} |"
        print(out, flush=True)


if __name__ == "__main__":
    main()
