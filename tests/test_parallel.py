"""world_size-2 tests of the data-parallel logic on CPU (gloo): bucketed gradient all-reduce, the
equality 'average of shard gradients = full-batch gradient' the training step relies on, sharding
helpers and the Dice-partial reduction of sharded evaluation."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle
from unet_amd.inferers import shard_range
from unet_amd.parallel import (BucketedAllReduce, all_reduce_dice_partials, broadcast_parameters, init_from_env,
                               shard_batch)
from unet_amd.utils import per_volume_dice, volume_id


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _flat_grads(model):
    return torch.cat([p.grad.flatten() for p in model.parameters()])


def _worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    torch.set_num_threads(2)
    r, w, _ = init_from_env("gloo")
    assert (r, w) == (rank, world)
    try:
        # -- replicas start equal -------------------------------------------------------------------
        torch.manual_seed(100 + rank)                      # different seeds on purpose
        model = oracle.UNet(2, 1, 1, (16, 32), (2,), num_res_units=2)
        flat = torch.cat([p.detach().flatten() for p in model.parameters()])
        broadcast_parameters(flat, src=0)
        o = 0
        with torch.no_grad():
            for p in model.parameters():
                p.copy_(flat[o:o + p.numel()].view_as(p)); o += p.numel()
        # -- shard the global batch -------------------------------------------------------------------
        g = torch.Generator().manual_seed(7)
        x = torch.rand(4, 1, 16, 16, generator=g)
        t = (torch.rand(4, 1, 16, 16, generator=g) > 0.6).float()
        lo, hi = shard_batch(4, rank, world)
        loss = oracle.DiceCELoss(sigmoid=True)
        model.eval()
        loss(model(x[lo:hi]), t[lo:hi]).backward()
        grads = _flat_grads(model).clone()
        n = grads.numel()
        cuts = [0, n // 3, n // 3, n]                      # three buckets, the middle one empty
        ar = BucketedAllReduce(grads, [(cuts[i], cuts[i + 1] - cuts[i]) for i in range(3)])
        for i in range(3):
            ar.launch(i)
        ar.wait()
        grads /= world
        # -- full-batch gradient on every rank for comparison --------------------------------------------
        model.zero_grad()
        loss(model(x), t).backward()
        full = _flat_grads(model)
        err = float((grads - full).abs().max() / full.abs().max())
        # -- evaluation: slices sharded, only the Dice partials travel -----------------------------------
        names = [f"liver_{c}_{s}" for c in (3, 11) for s in range(5)]
        dices = [0.1 * (i % 7) for i in range(len(names))]
        vols = sorted({volume_id(nm) for nm in names})
        part = torch.zeros(len(vols), 2, dtype=torch.float64)
        a, b = shard_range(len(names), rank, world)
        for nm, d in list(zip(names, dices))[a:b]:
            part[vols.index(volume_id(nm)), 0] += d
            part[vols.index(volume_id(nm)), 1] += 1
        all_reduce_dice_partials(part)
        table = {v: float(part[i, 0] / part[i, 1]) for i, v in enumerate(vols)}
        expect = per_volume_dice(names, dices)
        ok_table = all(abs(table[v] - expect[v]) < 1e-12 for v in vols)
        out[rank] = (err, ok_table, float(flat.sum()))
    finally:
        dist.destroy_process_group()


def test_two_rank_gradient_average_equals_full_batch_gradient():
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    assert len(out) == world
    assert out[0][2] == out[1][2]                       # broadcast made the replicas identical
    for r in range(world):
        err, ok_table, _ = out[r]
        assert err < 1e-5, err
        assert ok_table


def test_shard_helpers_cover_everything_once():
    for fn in (shard_batch, shard_range):
        for n in (0, 1, 7, 32, 257):
            for world in (1, 2, 3, 8):
                seen = []
                for r in range(world):
                    lo, hi = fn(n, r, world)
                    assert 0 <= lo <= hi <= n
                    seen += list(range(lo, hi))
                assert seen == list(range(n))
    for n in (7, 32, 257):
        sizes = [shard_batch(n, r, 8)[1] - shard_batch(n, r, 8)[0] for r in range(8)]
        assert max(sizes) - min(sizes) <= 1


def test_single_process_is_a_no_op():
    g = torch.arange(10, dtype=torch.float32)
    ar = BucketedAllReduce(g, [(0, 5), (5, 5)])
    ar.launch(0); ar.launch(1); ar.wait()
    assert torch.equal(g, torch.arange(10, dtype=torch.float32))
    assert init_from_env() == (0, 1, 0) or dist.is_initialized()


def _run_bench(args, env_extra=None):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(root, "bench.py")] + args, capture_output=True, text=True, env=env,
                          timeout=240)


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` without torchrun's environment starts two ranks itself; rank 0's line reports
    the size of the process group."""
    import json
    r = _run_bench(["--gpus", "2", "--dry-run"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    assert lines[0]["n_gpus"] == 2 and lines[0]["config"]["parallelism"] == "dp2"


def test_bench_refuses_a_world_size_that_differs_from_gpus():
    r = _run_bench(["--gpus", "2", "--dry-run"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0
    assert "--gpus 2" in (r.stderr + r.stdout)
