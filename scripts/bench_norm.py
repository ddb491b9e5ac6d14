"""Times the instance-norm + PReLU kernels on contiguous and on half-pitch (channel-slice) tensors."""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unet_amd import _lib, ops

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
for c, h in ((16, 256), (32, 128), (64, 64), (128, 32), (256, 32)):
    for mode in ("contig", "half"):
        pitch = c if mode == "contig" else 2 * c
        raw = torch.randn(nb, h, h, pitch, device=dev).to(torch.bfloat16)
        out = torch.empty(nb, h, h, pitch, device=dev, dtype=torch.bfloat16)
        dy = torch.randn(nb, h, h, pitch, device=dev).to(torch.bfloat16)
        stats = torch.zeros(nb, 8, c, 2, dtype=torch.float64, device=dev)
        x32 = raw[..., :c].float()
        stats[:, 0, :, 0] = x32.sum(dim=(1, 2)).double(); stats[:, 0, :, 1] = (x32 * x32).sum(dim=(1, 2)).double()
        alpha = torch.full((1,), 0.25, device=dev)
        bst = torch.zeros(nb * 8 * c * 2, dtype=torch.float64, device=dev)
        da = torch.zeros(1, device=dev)
        def ten(t):
            m = _lib.MiTensor(); m.ptr, m.pitch, m.coff = t.data_ptr(), pitch, 0; return m
        traw, tout, tdy = ten(raw), ten(out), ten(dy)
        npix = h * h
        def fwd():
            _lib.check(lib.mi_instnorm_prelu_fwd(1, nb, npix, c, C.byref(traw), stats.data_ptr(), alpha.data_ptr(), None, C.byref(tout), 0.0, C.c_uint64(0), 0, st), "f")
        def bwd():
            bst.zero_()
            _lib.check(lib.mi_instnorm_prelu_bwd(1, nb, npix, c, C.byref(tdy), C.byref(traw), stats.data_ptr(), alpha.data_ptr(), bst.data_ptr(), da.data_ptr(), C.byref(tout), 0.0, C.c_uint64(0), 0, st), "b")
        n = nb * npix * c * 2
        tf, tb = timeit(fwd), timeit(bwd)
        # torch elementwise on the same bytes for calibration
        a = torch.randn(n // 2, device=dev, dtype=torch.bfloat16); b = torch.empty_like(a)
        tt = timeit(lambda: torch.mul(a, 1.5, out=b))
        print(f"c{c:3d} {h:3d}^2 {mode:6s}: fwd {tf:6.1f} us {2*n/tf/1e6:5.2f} TB/s | bwd {tb:6.1f} us {5*n/tb/1e6:5.2f} TB/s | torch mul {tt:6.1f} us {2*n/tt/1e6:5.2f} TB/s", flush=True)
