"""Oracle (test infrastructure): numpy restatement of the stateless dropout stream of the HIP path.

``nn.Dropout`` draws from torch's Philox stream (reference/train.py:85 passes ``dropout=0.1`` to MONAI, which
builds ``nn.Dropout``); a hand-written kernel cannot reproduce that stream, so the HIP path defines its own
counter-based one (``unet_amd/csrc/common.h``, "counter-based dropout RNG": ``mix32``, ``dropout_keys``,
``dropout_keep4``).  This file restates that definition with 32-bit integer arithmetic in numpy, so that the
oracle can be run with exactly the mask the HIP path used and train-mode logits / gradients can be compared
element for element.  If this restatement and the kernels disagreed on a single bit, a tenth of the masks'
entries would differ and the comparison would fail by orders of magnitude.

Definition: the elements of one image are numbered channels-last, ``idx = pixel * C + channel``; groups of four
consecutive indices share one 64-bit draw (two 32-bit hashes of the group number under keys derived from
(seed, layer, image)), 16 bits per element; an element is dropped when its 16 bits are below
``thresh16 = round(p * 65536)`` (clamped to [1, 65535]).  Kept elements are scaled by ``1 / (1 - p)``.
"""
from __future__ import annotations

import numpy as np

_M32 = np.uint64(0xFFFFFFFF)


def _u32(x):
    return np.asarray(x, dtype=np.uint64) & _M32


def mix32(x):
    x = _u32(x)
    x ^= x >> np.uint64(16); x = _u32(x * np.uint64(0x7FEB352D))
    x ^= x >> np.uint64(15); x = _u32(x * np.uint64(0x846CA68B))
    x ^= x >> np.uint64(16)
    return x


def dropout_thresh16(p: float) -> int:
    t = int(float(p) * 65536.0 + 0.5)
    return 1 if t < 1 else 65535 if t > 65535 else t


def dropout_keys(seed: int, layer: int, img: int):
    seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    a = mix32((seed + 0x9E3779B9) & 0xFFFFFFFF)
    a = mix32(a ^ np.uint64(seed >> 32))
    a = mix32(a ^ np.uint64(layer))
    a = mix32(a ^ np.uint64(img))
    return a, mix32(a ^ np.uint64(0x5BD1E995)), mix32(a ^ np.uint64(0xC2B2AE35))


def keep_mask_image(seed: int, layer: int, img: int, n_elems: int, p: float) -> np.ndarray:
    """bool[n_elems]: which elements (channels-last numbering) of image ``img`` survive."""
    k0, k1, k2 = dropout_keys(seed, layer, img)
    cut = np.uint64(dropout_thresh16(p) << 16)
    ng = (n_elems + 3) // 4
    g = np.arange(ng, dtype=np.uint64)
    y = _u32(g ^ k0)
    y ^= y >> np.uint64(16); y = _u32(y * np.uint64(0x7FEB352D))
    a, b = _u32(y + k1), _u32(y + k2)
    a ^= a >> np.uint64(15); a = _u32(a * np.uint64(0x846CA68B)); a ^= a >> np.uint64(16)
    b ^= b >> np.uint64(15); b = _u32(b * np.uint64(0x735A2D97)); b ^= b >> np.uint64(15)
    keep = np.empty((ng, 4), dtype=bool)
    keep[:, 0] = _u32(a << np.uint64(16)) >= cut
    keep[:, 1] = a >= cut
    keep[:, 2] = _u32(b << np.uint64(16)) >= cut
    keep[:, 3] = b >= cut
    return keep.reshape(-1)[:n_elems]


def keep_mask(seed: int, layer: int, shape, p: float) -> np.ndarray:
    """bool mask of a channels-first tensor ``[N, C, *spatial]`` (the layout the oracle computes in)."""
    n, c = int(shape[0]), int(shape[1])
    sp = tuple(int(s) for s in shape[2:])
    npix = int(np.prod(sp)) if sp else 1
    out = np.empty((n, c) + sp, dtype=bool)
    for i in range(n):
        m = keep_mask_image(seed, layer, i, npix * c, p).reshape(npix, c)      # [pixel][channel]
        out[i] = m.T.reshape((c,) + sp)
    return out
