"""The C-level gradient exchange (include/unet_mi355.h, mi_comm_*): what a host without torch.distributed calls between
mi_unet_backward_bucket and AdamW.  One GPU is visible, and RCCL refuses two ranks on one device, so the collective runs
in a communicator of one rank: creation from a unique id, an in-place all-reduce of a bucket of a flat buffer on a side
stream (a sum over one rank leaves the bucket as it is and must not touch its neighbours), an empty bucket, teardown."""
import ctypes as C

import pytest
import torch

from unet_amd import _lib

pytestmark = pytest.mark.gpu


def test_one_rank_communicator_reduces_a_bucket_in_place(dev):
    lib = _lib.load()
    uid = (C.c_char * 128)()
    rc = lib.mi_comm_unique_id(C.cast(uid, C.c_void_p))
    if rc == -3:
        pytest.skip("RCCL is not available in this process: " + lib.mi_last_error().decode())
    _lib.check(rc, "mi_comm_unique_id")
    comm = C.c_void_p()
    torch.cuda.set_device(dev)
    _lib.check(lib.mi_comm_init(1, 0, C.cast(uid, C.c_void_p), C.byref(comm)), "mi_comm_init")
    try:
        g = torch.Generator().manual_seed(3)
        flat = torch.randn(5000, generator=g).to(dev)
        want = flat.clone()
        side = torch.cuda.Stream(dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        _lib.check(lib.mi_comm_allreduce_bucket(comm, flat.data_ptr(), 1000, 3000, side.cuda_stream), "mi_comm_allreduce_bucket")
        _lib.check(lib.mi_comm_allreduce_bucket(comm, flat.data_ptr(), 4000, 0, side.cuda_stream), "mi_comm_allreduce_bucket (empty)")
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize()
        assert torch.equal(flat, want)
        # argument checks fail loudly, not silently
        assert lib.mi_comm_allreduce_bucket(comm, flat.data_ptr(), -1, 10, side.cuda_stream) != 0
        assert lib.mi_comm_init(2, 2, C.cast(uid, C.c_void_p), C.byref(C.c_void_p())) != 0
    finally:
        _lib.check(lib.mi_comm_destroy(comm), "mi_comm_destroy")
