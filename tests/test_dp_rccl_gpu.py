"""The data-parallel training step on the GPU over RCCL (torch.distributed backend "nccl"), in a process group
of one rank: the bucket-by-bucket backward, the all-reduce of every bucket on the communication stream and the
event ordering between the engine's streams, the communication stream and AdamW all run on hardware; a
one-rank sum leaves the numbers unchanged, so the result must equal the plain single-process step bit for bit."""
import os
import socket

import pytest
import torch
import torch.distributed as dist

import unet_amd
from test_unet_gpu import disks

pytestmark = pytest.mark.gpu


@pytest.fixture()
def rccl_group(dev):
    if dist.is_initialized():
        pytest.skip("a process group already exists in this process")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)
    try:
        yield
    finally:
        torch.cuda.synchronize()
        dist.destroy_process_group()


def _run(dev, force, steps=3, overlap=True):
    torch.manual_seed(9)
    m = unet_amd.UNet(2, 1, 1, (16, 32, 64, 128), (2, 2, 2), num_res_units=2, dropout=0.1, compute_dtype=torch.bfloat16).to(dev).train()
    m.seed_dropout(77)
    opt = unet_amd.AdamW(m.parameters(), lr=1e-3, weight_decay=1e-5, model=m)
    tr = unet_amd.Trainer(m, unet_amd.DiceLoss(sigmoid=True, fold_label_two=True), opt, force_reducer=force, overlap=overlap)
    g = torch.Generator().manual_seed(5)
    x = torch.rand(4, 1, 64, 96, generator=g).to(dev)
    t = disks(4, (64, 96), 6, with_two=True).to(dev)
    grads = []
    for _ in range(steps):
        tr.step(x, t)
        torch.cuda.synchronize()
        grads.append(tr._grad.clone())
    assert (tr._reducer is not None) == force
    return m.flat_parameters().detach().clone(), grads, tr.loss_sum()


def test_bucketed_rccl_step_equals_the_plain_step_bit_for_bit(dev, rccl_group):
    p0, g0, l0 = _run(dev, False)
    p1, g1, l1 = _run(dev, True, overlap=True)
    p2, g2, l2 = _run(dev, True, overlap=False)
    for a, b, c in zip(g0, g1, g2):
        assert torch.equal(a, b) and torch.equal(a, c)
    assert torch.equal(p0, p1) and torch.equal(p0, p2)
    assert l0 == l1 == l2 and l0 > 0
    assert bool(torch.isfinite(p0).all())


def test_rccl_all_reduce_of_a_bucket_view_sums_in_place(dev, rccl_group):
    flat = torch.arange(1000, dtype=torch.float32, device=dev)
    ar = unet_amd.parallel.BucketedAllReduce(flat, [(0, 300), (300, 0), (300, 700)], force=True)
    for i in range(3):
        ar.launch(i)
    ar.wait()
    torch.cuda.synchronize()
    assert torch.equal(flat.cpu(), torch.arange(1000, dtype=torch.float32))
