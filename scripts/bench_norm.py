"""Times the instance-norm + PReLU kernels layer by layer (shapes of the 512x512 batch-32 benchmark);
run under `rocprofv3 --kernel-trace --stats` to see the reduce and apply passes separately."""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
drop = float(os.environ.get("DROP", "0.1"))
shapes = ((16, 256), (32, 128), (64, 64), (128, 32), (256, 32))
if len(sys.argv) > 1:
    shapes = tuple(s for s in shapes if str(s[0]) in sys.argv[1:])
for c, h in shapes:
    raw = torch.randn(nb, h, h, c, device=dev).to(torch.bfloat16)
    out = torch.empty_like(raw)
    dy = torch.randn(nb, h, h, c, device=dev).to(torch.bfloat16)
    stats = torch.zeros(nb, 8, c, 2, dtype=torch.float64, device=dev)
    x32 = raw.float()
    stats[:, 0, :, 0] = x32.sum(dim=(1, 2)).double(); stats[:, 0, :, 1] = (x32 * x32).sum(dim=(1, 2)).double()
    alpha = torch.full((1,), 0.25, device=dev)
    bst = torch.zeros(nb * 8 * c * 2, dtype=torch.float64, device=dev)
    da = torch.zeros(1, device=dev)
    def ten(t):
        m = _lib.MiTensor(); m.ptr, m.pitch, m.coff = t.data_ptr(), c, 0; return m
    traw, tout, tdy = ten(raw), ten(out), ten(dy)
    npix = h * h
    def fwd():
        _lib.check(lib.mi_instnorm_prelu_fwd(1, nb, npix, c, C.byref(traw), stats.data_ptr(), alpha.data_ptr(), None, C.byref(tout), drop, C.c_uint64(0), 0, st), "f")
    def bwd():
        _lib.check(lib.mi_instnorm_prelu_bwd(1, nb, npix, c, C.byref(tdy), C.byref(traw), stats.data_ptr(), alpha.data_ptr(), bst.data_ptr(), da.data_ptr(), C.byref(tout), drop, C.c_uint64(0), 0, st), "b")
    n = nb * npix * c * 2
    tf, tb = timeit(fwd), timeit(bwd)
    print(f"c{c:3d} {h:3d}^2: fwd {tf:6.1f} us {2*n/tf/1e6:5.2f} TB/s | bwd {tb:6.1f} us {5*n/tb/1e6:5.2f} TB/s", flush=True)
