"""One rank of the two-rank data-parallel rehearsal (started by tests/test_zy_dp_two_rank_gpu.py, never collected itself):

    python dp_child.py RANK WORLD PORT OUT.pt

Trains three steps on this rank's shard of a fixed batch with the bucketed, overlapped gradient exchange of
``unet_amd.Trainer`` (gloo all-reduce, staged through pinned host memory by the reducer: all ranks share the one visible
GPU, where RCCL refuses two ranks) and saves its parameters and the averaged gradient of the first step.  WORLD = 1 is the full-batch run the
ranks are compared with.  A watchdog ends the process after 100 s: a stuck collective must not outlive the test."""
import faulthandler
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

CHANNELS = (16, 32, 64)


def data():
    g = torch.Generator().manual_seed(21)
    x = torch.rand(4, 1, 32, 32, generator=g)
    t = (torch.rand(4, 1, 32, 32, generator=g) > 0.6).float()
    return x, t


def train(x, t, steps):
    import unet_amd

    dev = torch.device("cuda", 0)
    torch.manual_seed(5)
    m = unet_amd.UNet(2, 1, 1, CHANNELS, (2, 2), num_res_units=2, compute_dtype=torch.float32).to(dev).train()
    opt = unet_amd.AdamW(m.parameters(), lr=1e-3, weight_decay=1e-5, model=m)
    tr = unet_amd.Trainer(m, unet_amd.DiceCELoss(sigmoid=True), opt)
    grads = None
    t0 = time.time()
    for i in range(steps):
        tr.step(x.to(dev), t.to(dev))
        torch.cuda.synchronize()
        print(f"[rank {tr.rank}/{tr.world}] step {i} done at +{time.time() - t0:.2f} s", flush=True)
        if i == 0:
            grads = (tr._grad / tr.world).cpu().clone()       # the gradient the optimizer consumed in step 1
    nb = len(tr._reducer.buckets) if tr._reducer is not None else 0
    return {"params": m.flat_parameters().detach().cpu().clone(), "grads": grads, "world": tr.world, "buckets": nb}


def main():
    rank, world, port, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    faulthandler.dump_traceback_later(100, exit=True)
    import torch.distributed as dist
    from unet_amd.parallel import init_from_env, shard_batch

    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
    init_from_env("gloo")
    torch.cuda.set_device(0)
    x, t = data()
    lo, hi = shard_batch(x.shape[0], rank, world)
    res = train(x[lo:hi], t[lo:hi], 3)
    torch.save(res, out)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    try:
        main()
    except BaseException:   # the peer would wait for ever in its collective: report and leave at once
        import traceback

        traceback.print_exc()
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(3)
