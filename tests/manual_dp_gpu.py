"""Data-parallel training step on the GPU: two ranks (gloo all-reduce on device tensors, both ranks on the
one visible GPU) train on the halves of a batch with the bucketed, overlapped gradient exchange; their
parameters must follow a single-process run on the whole batch.

Not collected by pytest (the file name does not start with test_): run it by hand with
``python -m pytest tests/manual_dp_gpu.py -s``.  Two processes time-share the one GPU and gloo completes its
collectives on the host, so this is a correctness rehearsal, not a model of the RCCL run.  An earlier version
issued one blocking all-reduce per bucket between the bucket launches of the backward pass and armed a 30 s
watchdog from process start; about one run in six ended in that watchdog with both ranks inside an
all-reduce.  The collectives are now asynchronous until ``BucketedAllReduce.wait`` (the host no longer waits
for the peer in the middle of the enqueue) and the watchdog allows 150 s with time stamps printed per step.
The same logic runs on every round in tests/test_parallel.py with CPU tensors, and the bucket / stream / event
path runs on the GPU over RCCL in tests/test_dp_rccl_gpu.py."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
CHANNELS = (16, 32, 64)


def _data():
    g = torch.Generator().manual_seed(21)
    x = torch.rand(4, 1, 32, 32, generator=g)
    t = (torch.rand(4, 1, 32, 32, generator=g) > 0.6).float()
    return x, t


def _train(x, t, steps, group_ready):
    import unet_amd
    dev = torch.device("cuda", 0)
    torch.manual_seed(5)
    m = unet_amd.UNet(2, 1, 1, CHANNELS, (2, 2), num_res_units=2, compute_dtype=torch.float32).to(dev).train()
    opt = unet_amd.AdamW(m.parameters(), lr=1e-3, weight_decay=1e-5, model=m)
    tr = unet_amd.Trainer(m, unet_amd.DiceCELoss(sigmoid=True), opt)
    grads = None
    import time
    t0 = time.time()
    for i in range(steps):
        tr.step(x.to(dev), t.to(dev))
        torch.cuda.synchronize()
        print(f"[rank {tr.rank}/{tr.world}] step {i} done at +{time.time() - t0:.2f} s", flush=True)
        if i == 0:
            torch.cuda.synchronize()
            grads = (tr._grad / tr.world).cpu().clone()       # the gradient the optimizer consumed in step 1
    torch.cuda.synchronize()
    return m.flat_parameters().detach().cpu().clone(), grads


def _worker(rank, world, port, out):
    import torch.distributed as dist
    from unet_amd.parallel import init_from_env, shard_batch
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import faulthandler
    faulthandler.dump_traceback_later(150, exit=True)      # a stuck collective must not outlive the test
    try:
        init_from_env("gloo")
        torch.cuda.set_device(0)
        x, t = _data()
        lo, hi = shard_batch(x.shape[0], rank, world)
        out[rank] = _train(x[lo:hi], t[lo:hi], 3, True)
    except Exception as e:  # reported to the parent; the peer would wait for ever in its collective, so leave at once
        import traceback
        out[rank] = repr(e)
        traceback.print_exc()
        os._exit(3)
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def test_two_ranks_follow_the_full_batch_run(dev):
    x, t = _data()
    ref_params, ref_grads = _train(x, t, 3, False)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    assert len(out) == 2
    for r in range(2):
        if isinstance(out[r], str):
            if "gloo" in out[r].lower() or "not supported" in out[r].lower() or "backend" in out[r].lower():
                pytest.skip(f"device tensors not supported by this gloo build: {out[r]}")
            raise AssertionError(out[r])
    assert torch.equal(out[0][0], out[1][0])                 # replicas stay identical, bit for bit
    assert torch.equal(out[0][1], out[1][1])
    # averaged shard gradients = full-batch gradient (Dice and BCE are means over samples / pixels)
    err = float((out[0][1] - ref_grads).abs().max() / ref_grads.abs().max())
    assert err < 1e-4, err
    # Adam turns rounding-level gradients (biases in front of an instance norm) into +-lr steps, so the
    # parameters are only compared loosely: three steps of lr = 1e-3
    assert float((out[0][0] - ref_params).abs().max()) < 3.5e-3
