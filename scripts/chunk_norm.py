"""Times the norm backward pair in image chunks: does the second pass find the first pass's data in a cache?"""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.getcwd())
from unet_amd import _lib
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
dev = torch.device("cuda", 0)
lib = _lib.load(); st = _lib.current_stream(dev)
nb = 32
for c, h in ((16, 256), (32, 128), (64, 64)):
    raw = torch.randn(nb, h, h, c, device=dev).to(torch.bfloat16)
    out = torch.empty_like(raw)
    dy = torch.randn(nb, h, h, c, device=dev).to(torch.bfloat16)
    stats = torch.zeros(nb, 8, c, 2, dtype=torch.float64, device=dev)
    x32 = raw.float()
    stats[:, 0, :, 0] = x32.sum(dim=(1, 2)).double(); stats[:, 0, :, 1] = (x32 * x32).sum(dim=(1, 2)).double()
    alpha = torch.full((1,), 0.25, device=dev)
    bst = torch.zeros(nb, 8 * c * 2, dtype=torch.float64, device=dev)
    da = torch.zeros(1, device=dev)
    npix = h * h
    esz = 2
    def ten(t, i0):
        m = _lib.MiTensor(); m.ptr, m.pitch, m.coff = t.data_ptr() + i0 * npix * c * esz, c, 0; return m
    for ch in (32, 16, 8, 4, 2):
        args = []
        for i0 in range(0, nb, ch):
            args.append((ten(dy, i0), ten(raw, i0), ten(out, i0), stats[i0:].data_ptr(), bst[i0:].data_ptr()))
        def bwd():
            for tdy, traw, tout, sp, bp in args:
                _lib.check(lib.mi_instnorm_prelu_bwd(1, ch, npix, c, C.byref(tdy), C.byref(traw), sp, alpha.data_ptr(), bp, da.data_ptr(), C.byref(tout), 0.1, C.c_uint64(0), 0, st), "b")
        tb = timeit(bwd)
        n = nb * npix * c * 2
        print(f"c{c:3d} {h:3d}^2 chunk {ch:2d}: bwd {tb:6.1f} us  {3*n/tb/1e6:5.2f} TB/s algorithmic", flush=True)
