"""Operator-level Python bindings (channels-last tensors) over the C ABI: convolution forward / data
gradient / weight gradient and instance-norm + PReLU.  Used by the operator parity tests and by
callers that want to assemble other networks from the same kernels.

Tensors are ``[N, (D,) H, W, C]`` contiguous, dtype ``torch.bfloat16`` or ``torch.float32``.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _lib

_DT = {torch.float32: 0, torch.bfloat16: 1}


def _desc(x_shape, cin, cout, ksize, stride, transposed, dtype):
    d = _lib.MiConvDesc()
    nd = len(x_shape) - 2
    d.ndim, d.transposed, d.ksize, d.stride = nd, 1 if transposed else 0, ksize, stride
    d.cin, d.cout, d.nb = cin, cout, x_shape[0]
    sp = [1] * (3 - nd) + [int(v) for v in x_shape[1:-1]]
    for i in range(3):
        d.in_dims[i] = sp[i]
    d.dtype = _DT[dtype]
    return d


def _ten(t: Optional[torch.Tensor], coff: int = 0):
    m = _lib.MiTensor()
    if t is not None:
        m.ptr, m.pitch, m.coff = t.data_ptr(), t.shape[-1], coff
    return m


def _out_spatial(sp, ksize, stride, transposed):
    if transposed:
        return [2 * s for s in sp]
    pad = (ksize - 1) // 2
    return [(s + 2 * pad - ksize) // stride + 1 for s in sp]


def pack_weights(w: torch.Tensor, x_shape, ksize, stride, transposed, dtype, dgrad=False) -> torch.Tensor:
    lib = _lib.load()
    cin, cout = (w.shape[0], w.shape[1]) if transposed else (w.shape[1], w.shape[0])
    d = _desc(x_shape, cin, cout, ksize, stride, transposed, dtype)
    nbytes = lib.mi_conv_packed_bytes(C.byref(d))
    out = torch.empty(nbytes, dtype=torch.uint8, device=w.device)
    wf = w.contiguous().float()
    _lib.check(lib.mi_conv_pack(C.byref(d), 1 if dgrad else 0, wf.data_ptr(), out.data_ptr(),
                                _lib.current_stream(w.device)), "mi_conv_pack")
    return out


def conv_forward(x, w, bias=None, ksize=3, stride=1, transposed=False, addend=None, want_stats=False):
    """x: channels-last; w: torch layout fp32.  Returns y (channels-last) [, stats [N,8,Cout,2] fp64]."""
    lib = _lib.load()
    cin, cout = (w.shape[0], w.shape[1]) if transposed else (w.shape[1], w.shape[0])
    d = _desc(x.shape, cin, cout, ksize, stride, transposed, x.dtype)
    packed = pack_weights(w, x.shape, ksize, stride, transposed, x.dtype, False)
    osp = _out_spatial(list(x.shape[1:-1]), ksize, stride, transposed)
    y = torch.empty([x.shape[0]] + osp + [cout], dtype=x.dtype, device=x.device)
    stats = torch.zeros(x.shape[0], 8, cout, 2, dtype=torch.float64, device=x.device) if want_stats else None
    b = bias.contiguous().float() if bias is not None else None
    tx, ty, ta = _ten(x), _ten(y), _ten(addend)
    _lib.check(lib.mi_conv_forward(C.byref(d), C.byref(tx), packed.data_ptr(), _lib.ptr(b), C.byref(ty),
                                   C.byref(ta) if addend is not None else None, _lib.ptr(stats),
                                   _lib.current_stream(x.device)), "mi_conv_forward")
    return (y, stats) if want_stats else y


def conv_dgrad(dy, w, x_shape, ksize=3, stride=1, transposed=False, addend=None):
    lib = _lib.load()
    cin, cout = (w.shape[0], w.shape[1]) if transposed else (w.shape[1], w.shape[0])
    d = _desc(x_shape, cin, cout, ksize, stride, transposed, dy.dtype)
    packed = pack_weights(w, x_shape, ksize, stride, transposed, dy.dtype, True)
    dx = torch.empty(list(x_shape[:-1]) + [cin], dtype=dy.dtype, device=dy.device)
    tdy, tdx, ta = _ten(dy), _ten(dx), _ten(addend)
    _lib.check(lib.mi_conv_dgrad(C.byref(d), C.byref(tdy), packed.data_ptr(), C.byref(tdx),
                                 C.byref(ta) if addend is not None else None, _lib.current_stream(dy.device)),
               "mi_conv_dgrad")
    return dx


def conv_wgrad(x, dy, w_shape, ksize=3, stride=1, transposed=False):
    """Returns (dw in the torch layout, db [cout] or None for transposed convolutions)."""
    lib = _lib.load()
    cin, cout = (w_shape[0], w_shape[1]) if transposed else (w_shape[1], w_shape[0])
    d = _desc(x.shape, cin, cout, ksize, stride, transposed, x.dtype)
    ws = torch.empty(lib.mi_conv_wgrad_workspace_bytes(C.byref(d)), dtype=torch.uint8, device=x.device)
    dw = torch.empty(w_shape, dtype=torch.float32, device=x.device)
    db = None if transposed else torch.empty(cout, dtype=torch.float32, device=x.device)
    tx, tdy = _ten(x), _ten(dy)
    _lib.check(lib.mi_conv_wgrad(C.byref(d), C.byref(tx), C.byref(tdy), dw.data_ptr(), _lib.ptr(db), ws.data_ptr(),
                                 _lib.current_stream(x.device)), "mi_conv_wgrad")
    return dw, db


def _dims3(sp):
    return (C.c_int32 * 3)(*([1] * (3 - len(sp)) + [int(v) for v in sp]))


def conv1_forward(x, w, bias=None, dtype=torch.float32, split=None, want_stats=False):
    """F.conv (k 3, stride 2, padding 1) of a one-channel fp32 tensor ``x`` [N, (D,) H, W] with ``w`` [C, 1, 3, ..]:
    channels-last result of ``dtype``; ``split``: write the channels from ``split`` on to a second tensor (returned as
    a pair).  With ``want_stats`` also the [N, 8, C, 2] fp64 sums of the bias-free result."""
    lib = _lib.load()
    nd, n = x.dim() - 1, w.shape[0]
    osp = [v // 2 for v in x.shape[1:]]
    c1 = n if split is None else split
    y = torch.empty([x.shape[0]] + osp + [c1], dtype=dtype, device=x.device)
    y2 = torch.empty([x.shape[0]] + osp + [n - c1], dtype=dtype, device=x.device) if split is not None else None
    ty = _ten(y)
    if y2 is not None:
        ty.ptr2, ty.pitch2, ty.coff2, ty.c1 = y2.data_ptr(), y2.shape[-1], 0, c1
    stats = torch.zeros(x.shape[0], 8, n, 2, dtype=torch.float64, device=x.device) if want_stats else None
    wf = w.contiguous().float()
    b = bias.contiguous().float() if bias is not None else None
    _lib.check(lib.mi_conv1_forward(_DT[dtype], nd, x.shape[0], _dims3(x.shape[1:]), n, x.data_ptr(), wf.data_ptr(), _lib.ptr(b),
                                    C.byref(ty), _lib.ptr(stats), _lib.current_stream(x.device)), "mi_conv1_forward")
    out = y if y2 is None else (y, y2)
    return (out, stats) if want_stats else out


def conv1_transposed(y, w, bias=None, y2=None, want_stats=False):
    """F.conv_transpose (k 3, stride 2, padding 1, output_padding 1) of channels-last ``y`` (+ ``y2``) with ``w``
    [C, 1, 3, ..] to ONE output channel: fp32 [N, (D,) H, W] on the fine grid."""
    lib = _lib.load()
    nd = y.dim() - 2
    n = y.shape[-1] + (y2.shape[-1] if y2 is not None else 0)
    fsp = [2 * v for v in y.shape[1:-1]]
    out = torch.empty([y.shape[0]] + fsp, dtype=torch.float32, device=y.device)
    ty = _ten(y)
    if y2 is not None:
        ty.ptr2, ty.pitch2, ty.coff2, ty.c1 = y2.data_ptr(), y2.shape[-1], 0, y.shape[-1]
    stats = torch.zeros(y.shape[0], 8, 1, 2, dtype=torch.float64, device=y.device) if want_stats else None
    wf = w.contiguous().float()
    b = bias.contiguous().float() if bias is not None else None
    _lib.check(lib.mi_conv1_transposed(_DT[y.dtype], nd, y.shape[0], _dims3(fsp), n, C.byref(ty), wf.data_ptr(), _lib.ptr(b),
                                       out.data_ptr(), _lib.ptr(stats), _lib.current_stream(y.device)), "mi_conv1_transposed")
    return (out, stats) if want_stats else out


def conv1_wgrad(x, y, y2=None):
    """Weight and bias gradient of the 3^d stride-2 convolution between a one-channel fp32 tensor ``x``
    ([N, (D,) H, W], fine grid) and channels-last ``y`` (coarse grid; ``y2``: a second source whose channels follow
    those of ``y``).  Returns (dw [C, 1, 3, ..] fp32, db [C] fp32)."""
    lib = _lib.load()
    nd = x.dim() - 1
    n = y.shape[-1] + (y2.shape[-1] if y2 is not None else 0)
    dims = _dims3(x.shape[1:])
    ws = torch.empty(lib.mi_conv1_wgrad_workspace_bytes(nd, n, dims), dtype=torch.uint8, device=x.device)
    dw = torch.empty([n, 1] + [3] * nd, dtype=torch.float32, device=x.device)
    db = torch.empty(n, dtype=torch.float32, device=x.device)
    ty = _ten(y)
    if y2 is not None:
        ty.ptr2, ty.pitch2, ty.coff2, ty.c1 = y2.data_ptr(), y2.shape[-1], 0, y.shape[-1]
    _lib.check(lib.mi_conv1_wgrad(_DT[y.dtype], nd, x.shape[0], dims, n, x.data_ptr(), C.byref(ty), dw.data_ptr(),
                                  db.data_ptr(), ws.data_ptr(), _lib.current_stream(x.device)), "mi_conv1_wgrad")
    return dw, db


def instnorm_prelu_fwd(raw, stats, alpha, res=None, dropout=0.0, seed=0, layer=0):
    lib = _lib.load()
    nb, c = raw.shape[0], raw.shape[-1]
    npix = raw[0].numel() // c
    out = torch.empty_like(raw)
    traw, tres, tout = _ten(raw), _ten(res), _ten(out)
    _lib.check(lib.mi_instnorm_prelu_fwd(_DT[raw.dtype], nb, npix, c, C.byref(traw), stats.data_ptr(), alpha.data_ptr(),
                                         C.byref(tres) if res is not None else None, C.byref(tout), float(dropout),
                                         C.c_uint64(seed), layer, _lib.current_stream(raw.device)),
               "mi_instnorm_prelu_fwd")
    return out


def instnorm_prelu_bwd(dy, raw, stats, alpha, dropout=0.0, seed=0, layer=0):
    """Returns (dx, dalpha [1])."""
    lib = _lib.load()
    nb, c = raw.shape[0], raw.shape[-1]
    npix = raw[0].numel() // c
    dx = torch.empty_like(raw)
    bstats = torch.empty(nb * 8 * c * 2, dtype=torch.float64, device=raw.device)
    dalpha = torch.zeros(1, dtype=torch.float32, device=raw.device)
    tdy, traw, tdx = _ten(dy), _ten(raw), _ten(dx)
    _lib.check(lib.mi_instnorm_prelu_bwd(_DT[raw.dtype], nb, npix, c, C.byref(tdy), C.byref(traw), stats.data_ptr(),
                                         alpha.data_ptr(), bstats.data_ptr(), dalpha.data_ptr(), C.byref(tdx),
                                         float(dropout), C.c_uint64(seed), layer, _lib.current_stream(raw.device)),
               "mi_instnorm_prelu_bwd")
    return dx, dalpha
