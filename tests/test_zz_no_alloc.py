"""The loss forward that allocates nothing (mi_dice_ce_loss_fwd_noalloc: its per-sample sums live in a device
array inside the library) against mi_dice_ce_loss_fwd with a caller buffer, through the C ABI: two calls in a
row, the second on a smaller input, so that what the first call left in the shared array must not leak."""
import ctypes as C
import gc

import pytest
import torch

from unet_amd import _lib

pytestmark = pytest.mark.gpu


def _bind(lib):
    return lib.mi_dice_ce_loss_fwd_noalloc


def _both(lib, fn, z, t, tcode, fold2, lambda_ce):
    dev = z.device
    nb, npix = z.shape[0], z[0].numel()
    st = _lib.current_stream(dev)
    sums = torch.empty(nb * 4, dtype=torch.float64, device=dev)
    ref, got = torch.zeros(1, device=dev), torch.full((1,), -1.0, device=dev)
    acc = torch.full((1,), 10.0, device=dev)
    _lib.check(lib.mi_dice_ce_loss_fwd(z.data_ptr(), t.data_ptr(), tcode, fold2, nb, npix, 1e-5, 1e-5, 1.0, lambda_ce,
                                       sums.data_ptr(), ref.data_ptr(), 0, st), "mi_dice_ce_loss_fwd")
    # the allocator's count must not move across the call; the cycle collector is held off meanwhile (tensors of
    # earlier tests that it frees at a moment of its own choosing would move the count without any allocation)
    gc.collect()
    gc.disable()
    try:
        before = torch.cuda.memory_allocated(dev)
        _lib.check(fn(z.data_ptr(), t.data_ptr(), tcode, fold2, nb, npix, 1e-5, 1e-5, 1.0, lambda_ce, got.data_ptr(),
                      acc.data_ptr(), st), "mi_dice_ce_loss_fwd_noalloc")
        assert torch.cuda.memory_allocated(dev) == before
    finally:
        gc.enable()
    torch.cuda.synchronize()
    return float(ref), float(got), float(acc)


def test_noalloc_loss_matches_the_workspace_variant_on_two_calls_in_a_row(dev):
    lib = _lib.load()
    fn = _bind(lib)
    g = torch.Generator().manual_seed(12)
    # first call: 24 samples of 192 x 160, float labels with the value 2 present (folded)
    z1 = (torch.randn(24, 1, 192, 160, generator=g) * 3).to(dev)
    t1 = torch.randint(0, 3, (24, 1, 192, 160), generator=g).float().to(dev)
    r1, g1, a1 = _both(lib, fn, z1, t1, 0, 1, 1.0)
    assert abs(g1 - r1) <= 1e-6 * abs(r1) and r1 > 0
    assert abs(a1 - (10.0 + g1)) < 1e-5
    # second call right after it: fewer and smaller samples, uint8 labels, odd pixel count (scalar path), Dice only
    z2 = (torch.randn(5, 1, 37, 23, generator=g) * 2).to(dev)
    t2 = (torch.rand(5, 1, 37, 23, generator=g) > 0.6).to(torch.uint8).to(dev)
    r2, g2, _ = _both(lib, fn, z2, t2, 1, 0, 0.0)
    assert abs(g2 - r2) <= 1e-6 * abs(r2) and r2 > 0
    # and the first input again: the array is cleared per call, so the first value comes back
    r3, g3, _ = _both(lib, fn, z1, t1, 0, 1, 1.0)
    assert abs(g3 - r1) <= 1e-6 * abs(r1)
    # sizes beyond the built-in array are refused with a message, nothing is launched
    rc = fn(z1.data_ptr(), t1.data_ptr(), 0, 1, 4097, 16, 1e-5, 1e-5, 1.0, 1.0, z1.data_ptr(), 0, _lib.current_stream(dev))
    assert rc != 0 and b"4096" in lib.mi_last_error()
