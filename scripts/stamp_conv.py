import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unet_amd import _lib, ops
dev = torch.device("cuda", 0); lib = _lib.load(); st = _lib.current_stream(dev)
cin, cout, h, nb = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), 32
dt = torch.bfloat16
x = torch.randn(nb, h, h, cin, device=dev).to(dt); y = torch.empty(nb, h, h, cout, device=dev, dtype=dt)
w = torch.randn(cout, cin, 3, 3, device=dev) * 0.05
d = ops._desc(x.shape, cin, cout, 3, 1, False, dt); pf = ops.pack_weights(w, x.shape, 3, 1, False, dt, False)
tx, ty = ops._ten(x), ops._ten(y)
dbg = torch.zeros(1 << 20, dtype=torch.int64, device=dev)
os.environ["MI_DBG_PTR"] = str(dbg.data_ptr())
for _ in range(3):
    _lib.check(lib.mi_conv_forward(C.byref(d), C.byref(tx), pf.data_ptr(), None, C.byref(ty), None, None, st), "fwd")
torch.cuda.synchronize()
v = dbg.view(-1, 8).cpu()
v = v[v[:, 4] > 0].double()
print("waves", v.shape[0], "chunks/wave", v[0, 5].item())
tot = v[:, 4]
for i, n in enumerate(["vmcnt wait", "barrier", "issue dma", "reads+mfma"]):
    print(f"{n:12s} mean {v[:, i].mean():9.0f} cyc  = {100 * (v[:, i] / tot).mean():5.1f} % of the wave's time")
print(f"total        mean {tot.mean():9.0f} cyc (median {tot.median():.0f}, max {tot.max():.0f})  other {100 * (1 - (v[:, :4].sum(1) / tot).mean()):.1f} %")
