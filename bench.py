#!/usr/bin/env python3
"""Benchmark of the U-Net training hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

One "step" = one pass of the reference's inner training loop body (reference/train.py:111-119) over one
synthetic batch: forward, Dice loss, backward, (gradient all-reduce when N > 1) and AdamW, as HIP
kernels on one stream.  Workload at N = 1: BASELINE.json configs[1] - 2-D residual U-Net
(channels 16-256, 2 res units), bf16 activations, 32 slices of 512x512 per GPU.  For N > 1 the script is
launched by torch.distributed.run (one rank per GPU, RCCL), every rank keeps 32 slices (weak scaling).

Rank 0 prints ONE JSON line on stdout; per-kernel tables go to stderr.
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

MFMA_BF16_PEAK = 2.5e15   # dense bf16 MFMA, FLOP/s (MI355X)
HBM_PEAK = 8.0e12         # HBM3E, B/s
RIDGE = MFMA_BF16_PEAK / HBM_PEAK


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def synthetic_batch(nb, sp, seed, device):
    """Images ~ U[0,1) (the reference scales CT intensities to [0,1]); labels = one disk per slice with
    5-25 % foreground and an inner disk of value 2 (exercises the label fold of train.py:112-113)."""
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(nb, 1, *sp, generator=g)
    grids = torch.meshgrid(*[torch.arange(s, dtype=torch.float32) for s in sp], indexing="ij")
    t = torch.zeros(nb, 1, *sp)
    for b in range(nb):
        c = [(0.3 + 0.4 * torch.rand(1, generator=g).item()) * s for s in sp]
        r = (0.1 + 0.2 * torch.rand(1, generator=g).item()) * min(sp)
        d2 = sum((gr - ci) ** 2 for gr, ci in zip(grids, c))
        t[b, 0] = (d2 < r * r).float()
        t[b, 0][d2 < (0.4 * r) ** 2] = 2.0
    return x.to(device), t.to(device)


def host_threads():
    """Threads the CPU baseline may use: the process's CPU share.  On a shared host os.cpu_count() reports the
    whole machine, so the count is capped by the scheduler affinity and by OMP_NUM_THREADS when that is set."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    env = os.environ.get("OMP_NUM_THREADS", "")
    if env.isdigit() and int(env) > 0:
        n = min(n, int(env))
    return max(1, n)


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(budget_s=12.0, dropout=0.0):
    """The CPU oracle (plain torch restatement of the reference path) on BASELINE.json configs[0]:
    2-D 16-256 U-Net, fp32, batch 8 of 256x256, zero_grad + forward + Dice loss + backward + AdamW,
    dropout 0.0; 5 warm-up steps, then up to 20 timed steps with every host thread of this process's
    CPU share, then a single-thread run (fewer steps: both legs are bounded by `budget_s`)."""
    import oracle

    threads = host_threads()
    x, t = synthetic_batch(8, (256, 256), 1234, "cpu")
    t[t == 2] = 1

    def run(nthreads, warm, steps, budget):
        torch.set_num_threads(nthreads)
        torch.manual_seed(0)
        model = oracle.UNet(2, 1, 1, (16, 32, 64, 128, 256), (2, 2, 2, 2), num_res_units=2, dropout=dropout)
        model.train()
        loss = oracle.DiceLoss(sigmoid=True)
        opt = torch.optim.AdamW(model.parameters(), lr=5e-4, weight_decay=1e-5)

        def step():
            opt.zero_grad()
            loss(model(x), t).backward()
            opt.step()

        for _ in range(warm):
            step()
        n, t0 = 0, time.perf_counter()
        while n < steps:
            step()
            n += 1
            if time.perf_counter() - t0 > budget:
                break
        return n, time.perf_counter() - t0

    before = torch.get_num_threads()
    n, el = run(threads, 5, 20, budget_s)
    n1, el1 = run(1, 1, 5, budget_s)
    torch.set_num_threads(before)
    gf_step = 36.2
    return {"value": round(8 * n / el, 2), "unit": "slices/s", "cores": threads, "kind": "port",
            "value_1thread": round(8 * n1 / el1, 2), "gflops": round(gf_step * n / el, 1),
            "gflops_1thread": round(gf_step * n1 / el1, 1), "cpu_model": cpu_model(), "host_cpu_count": os.cpu_count(),
            "dropout": dropout,
            "sample": f"{n} steps ({threads} threads, {el:.1f} s) and {n1} steps (1 thread, {el1:.1f} s) of batch 8 x 256x256 "
                      f"fp32, zero_grad+fwd+Dice loss+bwd+AdamW, dropout {dropout}; configs[0]; oracle = torch-CPU "
                      f"restatement of the reference path"}


def kernel_table(plan, nsteps):
    rows = {}
    for backward in (False, True):
        for s in plan.steps(backward):
            if s["launches"] == 0:
                continue
            r = rows.setdefault(s["kernel"], dict(ms=0.0, launches=0, flops=0.0, bytes=0.0))
            r["ms"] += s["total_ms"]
            r["launches"] += s["launches"]
            r["bytes"] += s["bytes"] * s["launches"]
            if s["work_is_flops"]:
                r["flops"] += s["work"] * s["launches"]
    return rows


CONV_KERNEL_TAGS = ("igemm", "conv2d_", "conv3d_", "wgrad_kernel", "wgrad_thick", "wgrad_thin", "wgrad_dma")


def plan_roofline(plan, nprof, dtype="bf16", backward=(False, True)):
    """Roofline entry of the dominant kernel (largest share of the instrumented kernel time) of `plan`, and the MFMA
    utilisation of its convolution kernels, from the per-launch HIP-event times mi_unet_prof_collect accumulated over
    `nprof` passes.  achieved = sum of the algorithmic work of that kernel's launches / sum of their durations."""
    rows = {}
    for bw in backward:
        for s in plan.steps(bw):
            if s["launches"] == 0:
                continue
            r = rows.setdefault(s["kernel"], dict(ms=0.0, launches=0, flops=0.0, bytes=0.0))
            r["ms"] += s["total_ms"]
            r["launches"] += s["launches"]
            r["bytes"] += s["bytes"] * s["launches"]
            if s["work_is_flops"]:
                r["flops"] += s["work"] * s["launches"]
    if not rows:
        return None, None, rows
    tot_ms = sum(r["ms"] for r in rows.values())
    conv = {k: r for k, r in rows.items() if r["flops"] > 0 and any(t in k for t in CONV_KERNEL_TAGS)}
    conv_s = sum(r["ms"] for r in conv.values()) / 1e3
    mfma_util = round(sum(r["flops"] for r in conv.values()) / conv_s / MFMA_BF16_PEAK, 4) if conv_s else None
    dom_name, dom = max(rows.items(), key=lambda kv: kv[1]["ms"])
    sec = dom["ms"] / 1e3
    ai = dom["flops"] / max(dom["bytes"], 1.0)
    if dom["flops"] > 0 and ai > RIDGE and dtype == "bf16":
        roof = {"bound": "mfma", "achieved": round(dom["flops"] / sec / 1e12, 2), "peak": MFMA_BF16_PEAK / 1e12, "unit": "TFLOP/s"}
    else:
        roof = {"bound": "hbm", "achieved": round(dom["bytes"] / sec / 1e9, 1), "peak": HBM_PEAK / 1e9, "unit": "GB/s"}
    roof["frac"] = round(roof["achieved"] / roof["peak"], 4)
    roof["traffic"] = None
    roof["algorithmic_bytes_per_launch"] = round(dom["bytes"] / max(1, dom["launches"]))
    roof["kernel"] = dom_name
    roof["avg_launch_us"] = round(1e3 * dom["ms"] / dom["launches"], 2)
    roof["launches_per_step"] = round(dom["launches"] / nprof, 1)
    roof["share_of_kernel_time"] = round(dom["ms"] / tot_ms, 3)
    return roof, mfma_util, rows


def profiled(plan, nprof, fn):
    """Runs fn() nprof times with every launch of `plan` bracketed by HIP events on its stream."""
    lib = plan.lib
    lib.mi_unet_prof_reset(plan.h)
    lib.mi_unet_prof_enable(plan.h, 1)
    for _ in range(nprof):
        fn()
    torch.cuda.synchronize()
    lib.mi_unet_prof_collect(plan.h)
    lib.mi_unet_prof_enable(plan.h, 0)


def secondary_train(dims, size, batch, dev, dropout, steps=10, warmup=3, config=""):
    """A side measurement of the training step on another workload of BASELINE.json (never `value`)."""
    import unet_amd

    ch = (16, 32, 64, 128, 256)
    torch.manual_seed(0)
    m = unet_amd.UNet(dims, 1, 1, ch, (2, 2, 2, 2), num_res_units=2, dropout=dropout, compute_dtype=torch.bfloat16).to(dev).train()
    tr = unet_amd.Trainer(m, unet_amd.DiceLoss(sigmoid=True, fold_label_two=True),
                          unet_amd.AdamW(m.parameters(), lr=5e-4, weight_decay=1e-5, model=m))
    x, t = synthetic_batch(batch, (size,) * dims, 4321, dev)
    for _ in range(warmup):
        tr.step(x, t)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        tr.step(x, t)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    plan = m._plan_for(x, True)
    flops = plan.flops(False) + plan.flops(True)
    nprof = min(steps, 5)
    profiled(plan, nprof, lambda: tr.step(x, t))
    roof, util, _ = plan_roofline(plan, nprof)
    unit = "slices/s" if dims == 2 else "patches/s"
    return {"workload": f"{dims}D U-Net ch 16-256 training, batch {batch} of {'x'.join([str(size)] * dims)}, bf16, dropout {dropout} ({config})",
            "value": round(batch * steps / el, 2), "unit": unit, "steps": steps, "ms_per_step": round(1e3 * el / steps, 4),
            "conv_gflop_per_step": round(flops / 1e9, 1), "conv_tflops_whole_step": round(flops / (el / steps) / 1e12, 1),
            "mfma_util": util, "roofline": roof}


def secondary_infer(dev, batch=32, size=512, roi=256, steps=10, warmup=3):
    """Sliding-window evaluation (reference/train.py:128-153) of `batch` slices per step: side measurement with the
    roofline entry of the dominant kernel of the forward pass."""
    import unet_amd
    from unet_amd.evaluator import VolumeEvaluator

    ch = (16, 32, 64, 128, 256)
    torch.manual_seed(0)
    m = unet_amd.UNet(2, 1, 1, ch, (2, 2, 2, 2), num_res_units=2, dropout=0.1, compute_dtype=torch.bfloat16).to(dev).eval()
    x, t = synthetic_batch(batch, (size, size), 4321, dev)
    ev = VolumeEvaluator(m, (roi, roi), sw_batch_size=288, slices_per_call=32)
    for _ in range(warmup):
        ev.slice_dice(x, t)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        ev.slice_dice(x, t)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    nwin = len(unet_amd.window_starts([size, size], [roi, roi], 0.25))
    plans = [p for p in m._plans.values()]
    flops = sum(p.flops(False) for p in plans if p.batch == batch * nwin) or batch * nwin * 1.509e9
    plan = max(plans, key=lambda p: p.batch)
    nprof = min(steps, 3)
    profiled(plan, nprof, lambda: ev.slice_dice(x, t))
    roof, util, _ = plan_roofline(plan, nprof, backward=(False,))
    return {"workload": f"sliding-window inference + hard Dice, {batch} slices of {size}x{size} per step, roi {roi}, overlap 0.25 ({nwin} windows per "
                        f"slice, 288 windows per model call), bf16 (BASELINE.json configs[4] shape, one GPU)",
            "value": round(batch * steps / el, 2), "unit": "slices/s", "steps": steps, "ms_per_step": round(1e3 * el / steps, 4),
            "conv_gflop_per_step": round(flops / 1e9, 1), "conv_tflops_whole_step": round(flops / (el / steps) / 1e12, 1),
            "mfma_util": util, "roofline": roof}


def launch_ranks(n, argv):
    """Parent of a multi-GPU run started without torchrun: starts `n` ranks of this script under
    torch.distributed.run (one per GPU), passes rank 0's JSON line through and fails when any rank fails or the
    line does not report `n` ranks.  The parent never touches the GPU."""
    import socket
    import subprocess

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    log(f"[bench] launching {n} ranks: {' '.join(cmd[1:])}")
    proc = subprocess.run(cmd, stdout=subprocess.PIPE, text=True, env=env)
    line = None
    for ln in proc.stdout.splitlines():
        try:
            d = json.loads(ln)
        except ValueError:
            continue
        if isinstance(d, dict) and "metric" in d:
            line = (ln, d)
    if proc.returncode != 0:
        raise SystemExit(f"bench.py: the {n}-rank run failed (exit code {proc.returncode})")
    if line is None:
        raise SystemExit("bench.py: rank 0 printed no result line")
    if line[1].get("n_gpus") != n:
        raise SystemExit(f"bench.py: asked for {n} ranks, the run reports {line[1].get('n_gpus')}")
    print(line[0], flush=True)


def dry_run(args):
    """Rendezvous check without a GPU: the ranks meet over gloo, sum their ranks and rank 0 prints a line of
    the usual shape.  Used by the CPU tests of the launcher."""
    import torch.distributed as dist
    from unet_amd.parallel import init_from_env

    rank, world, _ = init_from_env("gloo")
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the process group has {world} rank(s)")
    v = torch.tensor([float(rank)], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(v)
        dist.barrier()
    assert float(v.item()) == world * (world - 1) / 2
    if rank == 0:
        print(json.dumps({"metric": "dry run (rendezvous only)", "value": 0.0, "unit": "none", "n_gpus": world, "steps": 0,
                          "warmup": 0, "ms_per_step": 0.0, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                          "dtype": "none", "data": "none", "config": {"workload": "none", "parallelism": f"dp{world}"}}),
              flush=True)
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dry-run", action="store_true", help="rendezvous of the ranks over gloo only (no GPU work): launcher test")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=32, help="slices per GPU")
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--dims", type=int, default=2, choices=(2, 3))
    ap.add_argument("--dtype", default="bf16", choices=("bf16", "fp32"))
    ap.add_argument("--dropout", type=float, default=0.1, help="reference/train.py:85 trains with 0.1")
    ap.add_argument("--channels", default="16,32,64,128,256")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--no-reference-network", action="store_true",
                    help="skip the side measurement on the channel widths of reference/train.py:82 (32-512)")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the side measurements of the other BASELINE.json workloads (3-D patches, inference, batch 8 x 256^2)")
    ap.add_argument("--no-overlap", action="store_true", help="all-reduce after the whole backward")
    ap.add_argument("--force-reducer", action="store_true",
                    help="run the bucketed all-reduce path in a process group of one rank (what a rank of a data-parallel job "
                         "pays for the bucket-by-bucket backward and the communication stream, without a peer); needs torchrun")
    ap.add_argument("--host-input", action="store_true",
                    help="train mode: every step's batch starts in pinned host memory (fp32 image and label) and is copied "
                         "over PCIe on a copy stream, double-buffered; reports the PCIe-inclusive rate, never the headline value")
    ap.add_argument("--mode", default="train", choices=("train", "infer", "pipeline"),
                    help="infer: sliding-window evaluation of --batch slices per GPU and step (roi 256, 9 windows per slice)")
    ap.add_argument("--roi", type=int, default=256)
    ap.add_argument("--sw-batch", type=int, default=288,
                    help="windows per model call in --mode infer (reference/train.py:140 passes 4; windows are independent samples, so the count changes throughput only)")
    args = ap.parse_args()

    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be at least 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        launch_ranks(args.gpus, sys.argv[1:])      # this process becomes the parent of the ranks
        return
    if args.dry_run:
        dry_run(args)
        return

    import unet_amd
    from unet_amd.parallel import init_from_env, broadcast_parameters
    import torch.distributed as dist

    rank, world, local = init_from_env()
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE is {world}: start it as `python bench.py --gpus N` "
                         f"or under torch.distributed.run with --nproc-per-node N")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an AMD GPU: the HIP path has no CPU fallback")
    if local >= torch.cuda.device_count():
        raise SystemExit(f"bench.py: rank {rank} wants GPU {local} but only {torch.cuda.device_count()} are visible")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    channels = tuple(int(c) for c in args.channels.split(","))
    sp = (args.size,) * args.dims
    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    torch.manual_seed(0)
    model = unet_amd.UNet(args.dims, 1, 1, channels, (2,) * (len(channels) - 1), num_res_units=2, dropout=args.dropout,
                          compute_dtype=dtype).to(dev).train()
    broadcast_parameters(model.flat_parameters())
    loss = unet_amd.DiceLoss(sigmoid=True, fold_label_two=True)
    opt = unet_amd.AdamW(model.parameters(), lr=5e-4, weight_decay=1e-5, model=model)
    trainer = unet_amd.Trainer(model, loss, opt, overlap=not args.no_overlap, force_reducer=args.force_reducer)
    x, t = synthetic_batch(args.batch, sp, 1234 + rank, dev)

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    if args.mode == "pipeline":
        # reference/unetDataset.py:68-93 on synthetic int16 CT-like slices resident in HBM: 4 crops of 256^2 per slice
        pipe = unet_amd.SliceCropPipeline()
        g = torch.Generator().manual_seed(7 + rank)
        raw = (torch.rand(args.batch, args.size, args.size, generator=g) * 900 - 400).to(torch.int16).to(dev)
        lab = (t[:, 0] > 0).to(torch.uint8).contiguous()
        for _ in range(args.warmup):
            pipe(raw, lab)
        sync_all()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            ci, cl = pipe(raw, lab)
        sync_all()
        elapsed = time.perf_counter() - t0
        if rank == 0:
            ncrop = world * args.batch * pipe.num_samples * args.steps
            print(json.dumps({
                "metric": "training crops/sec (whole node), GPU input pipeline", "value": round(ncrop / elapsed, 1), "unit": "crops/s",
                "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 4),
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int16->f32", "data": "synthetic",
                "config": {"workload": f"{args.batch} int16 slices of {args.size}x{args.size} per GPU and step -> {pipe.num_samples} crops of "
                                       f"{pipe.roi}x{pipe.roi} each: intensity window, foreground box, pad, pos/neg crop, rot90"}}), flush=True)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    if args.mode == "infer":
        # reference/train.py:128-153 on synthetic slices: sliding-window forward, sigmoid + threshold, Dice per slice
        from unet_amd.evaluator import VolumeEvaluator
        model.eval()
        ev = VolumeEvaluator(model, (args.roi,) * args.dims, sw_batch_size=args.sw_batch, slices_per_call=max(1, args.sw_batch // 9))
        for _ in range(args.warmup):
            ev.slice_dice(x, t)
        sync_all()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            d = ev.slice_dice(x, t)
        sync_all()
        elapsed = time.perf_counter() - t0
        if world > 1:
            tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            elapsed = float(tt.item())
        nwin = len(unet_amd.window_starts([max(args.size, args.roi)] * args.dims, [args.roi] * args.dims, 0.25))
        if rank == 0 and world == 1 and os.environ.get("BENCH_LAYERS"):
            # per-launch table of the forward pass of the largest window batch (every launch bracketed by HIP events)
            plan = max(model._plans.values(), key=lambda p: p.batch)
            profiled(plan, 3, lambda: ev.slice_dice(x, t))
            log("[bench] per-launch (layer) averages:")
            for st in plan.steps(False):
                if st["launches"]:
                    us = 1e3 * st["total_ms"] / st["launches"]
                    tf = st["work"] / (us * 1e-6) / 1e12 if st["work_is_flops"] else 0.0
                    log(f"  fwd {st['layer']:48s} {st['kernel']:34s} {us:9.2f} us {tf:8.1f} TF/s {st['bytes'] / (us * 1e-6) / 1e9:9.1f} GB/s")
        if rank == 0:
            print(json.dumps({
                "metric": "inference slices/sec (whole node), sliding window", "value": round(world * args.batch * args.steps / elapsed, 2),
                "unit": "slices/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": round(1e3 * elapsed / args.steps, 4), "higher_is_better": True, "scaling": "weak",
                "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
                "config": {"workload": f"{args.dims}D U-Net ch {'-'.join(map(str, channels))}, {args.batch} slices of "
                                       f"{'x'.join(map(str, sp))} per GPU and step, roi {args.roi}, overlap 0.25 ({nwin} windows per "
                                       f"slice), {args.sw_batch} windows per model call, hard Dice per slice (BASELINE.json configs[4] shape)",
                           "parallelism": f"slices sharded over {world} GPU(s)"},
                "mean_dice": round(float(torch.nan_to_num(d).mean()), 5)}), flush=True)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    if args.host_input:
        hx, ht = x.cpu().pin_memory(), t.cpu().pin_memory()
        bufs = [(torch.empty_like(x), torch.empty_like(t)) for _ in range(2)]
        copy_stream = torch.cuda.Stream(device=dev)
        ready = [torch.cuda.Event() for _ in range(2)]     # copy i has landed
        free = [torch.cuda.Event() for _ in range(2)]      # the step that read buffer i has been enqueued and finished
        main = torch.cuda.current_stream(dev)

        def upload(i):
            with torch.cuda.stream(copy_stream):
                copy_stream.wait_event(free[i])
                bufs[i][0].copy_(hx, non_blocking=True)
                bufs[i][1].copy_(ht, non_blocking=True)
                ready[i].record(copy_stream)

        def host_steps(n):
            for e in free:
                e.record(main)
            upload(0)
            for k in range(n):
                i = k & 1
                if k + 1 < n:
                    upload(i ^ 1)
                main.wait_event(ready[i])
                trainer.step(*bufs[i])
                free[i].record(main)

        host_steps(args.warmup)
        sync_all()
        t0 = time.perf_counter()
        host_steps(args.steps)
        sync_all()
        elapsed = time.perf_counter() - t0
        if rank == 0:
            print(json.dumps({
                "metric": "train slices/sec (whole node), batch copied from pinned host memory every step", "unit": "slices/s",
                "value": round(world * args.batch * args.steps / elapsed, 2), "n_gpus": world, "steps": args.steps,
                "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 4), "higher_is_better": True,
                "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
                "config": {"workload": f"as the default run, plus {(hx.numel() * 4 + ht.numel() * 4) / 1e6:.0f} MB per step over PCIe "
                                       "(fp32 image and label), double-buffered on a copy stream"}}), flush=True)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    for _ in range(args.warmup):
        trainer.step(x, t)
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        trainer.step(x, t)
    sync_all()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    last_loss = trainer.loss_sum() / max(1, args.steps + args.warmup)
    ms_per_step = 1e3 * elapsed / args.steps
    value = world * args.batch * args.steps / elapsed

    plan = model._plan_for(x, True)
    flops_step = plan.flops(False) + plan.flops(True)
    if args.dims == 3:
        which_config = "BASELINE.json configs[3]" if (args.size, args.channels) == (96, "16,32,64,128,256") else "3-D variant of BASELINE.json configs[3]"
    elif (args.size, args.batch, args.channels, args.dtype) == (512, 32, "16,32,64,128,256", "bf16"):
        which_config = "BASELINE.json configs[1]" if world == 1 else "BASELINE.json configs[2]"
    elif (args.size, args.batch, args.channels) == (256, 8, "16,32,64,128,256"):
        which_config = "shape of BASELINE.json configs[0] on the GPU"
    else:
        which_config = "variant of BASELINE.json configs[1]"
    out = {
        "metric": f"train slices/sec (whole node) 2D U-Net {args.size}x{args.size} {args.dtype}" if args.dims == 2 else "train patches/sec (whole node) 3D U-Net",
        "value": round(value, 2), "unit": "slices/s" if args.dims == 2 else "patches/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": f"{args.dims}D residual U-Net ch {'-'.join(map(str, channels))}, 2 res units, "
                               f"{'x'.join(map(str, sp))} 1-ch {'slices' if args.dims == 2 else 'patches'}, "
                               f"batch {args.batch}/GPU, Dice loss, AdamW ({which_config})",
                   "global_batch": world * args.batch, "parallelism": f"dp{world}", "dropout": args.dropout,
                   "conv_gflop_per_step_per_gpu": round(flops_step / 1e9, 1)},
        "mean_loss": round(last_loss, 5),
    }

    if rank == 0 and world == 1 and not args.no_kernel_timing:
        # per-kernel durations: the same step repeated with every launch bracketed by HIP events on its stream
        lib = plan.lib
        lib.mi_unet_prof_reset(plan.h)
        lib.mi_unet_prof_enable(plan.h, 1)
        nprof = min(args.steps, 10)
        for _ in range(nprof):
            trainer.step(x, t)
        torch.cuda.synchronize()
        lib.mi_unet_prof_collect(plan.h)
        lib.mi_unet_prof_enable(plan.h, 0)
        rows = kernel_table(plan, nprof)
        tot_ms = sum(r["ms"] for r in rows.values())
        log(f"[bench] per-kernel time over {nprof} instrumented steps ({tot_ms / nprof:.3f} ms/step in kernels):")
        log(f"  {'kernel':44s} {'ms/step':>9s} {'launch/step':>11s} {'avg us':>9s} {'TFLOP/s':>9s} {'GB/s':>9s}")
        for k, r in sorted(rows.items(), key=lambda kv: -kv[1]["ms"]):
            sec = r["ms"] / 1e3
            log(f"  {k:44s} {r['ms'] / nprof:9.4f} {r['launches'] / nprof:11.1f} {1e3 * r['ms'] / max(1, r['launches']):9.2f} "
                f"{(r['flops'] / sec / 1e12 if sec else 0):9.1f} {(r['bytes'] / sec / 1e9 if sec else 0):9.1f}")
        if os.environ.get("BENCH_LAYERS"):
            log("[bench] per-launch (layer) averages:")
            for backward in (False, True):
                for s in plan.steps(backward):
                    if s["launches"]:
                        us = 1e3 * s["total_ms"] / s["launches"]
                        tf = s["work"] / (us * 1e-6) / 1e12 if s["work_is_flops"] else 0.0
                        log(f"  {'bwd' if backward else 'fwd'} {s['layer']:48s} {s['kernel']:34s} {us:9.2f} us {tf:8.1f} TF/s {s['bytes'] / (us * 1e-6) / 1e9:9.1f} GB/s")
        conv = {k: r for k, r in rows.items() if r["flops"] > 0 and ("igemm" in k or "conv2d_" in k or "conv3d_" in k or "wgrad_kernel" in k or "wgrad_thick" in k or "wgrad_thin" in k)}
        conv_s = sum(r["ms"] for r in conv.values()) / 1e3
        conv_f = sum(r["flops"] for r in conv.values())
        out["mfma_util"] = round(conv_f / conv_s / MFMA_BF16_PEAK, 4) if conv_s else None
        # the same over the launches whose arithmetic intensity lies above the ridge of the roofline (FLOP per
        # algorithmic byte > MFMA peak / HBM peak): the layers the matrix cores, not the memory, can bound
        above = [st for bw in (False, True) for st in plan.steps(bw)
                 if st["work_is_flops"] and st["launches"] and st["bytes"] > 0 and st["work"] / st["bytes"] > RIDGE]
        if above and args.dtype == "bf16":
            a_ms = sum(st["total_ms"] for st in above)
            a_fl = sum(st["work"] * st["launches"] for st in above)
            out["mfma_bound_layers"] = {"launches_per_step": round(sum(st["launches"] for st in above) / nprof, 1),
                                        "tflops": round(a_fl / (a_ms / 1e3) / 1e12, 1),
                                        "frac_of_peak": round(a_fl / (a_ms / 1e3) / MFMA_BF16_PEAK, 4),
                                        "share_of_kernel_time": round(a_ms / tot_ms, 3),
                                        "ridge_flop_per_byte": round(RIDGE, 1)}
        dom_name, dom = max(rows.items(), key=lambda kv: kv[1]["ms"])
        sec = dom["ms"] / 1e3
        ai = dom["flops"] / max(dom["bytes"], 1.0)
        if dom["flops"] > 0 and ai > RIDGE and args.dtype == "bf16":
            roof = {"bound": "mfma", "achieved": round(dom["flops"] / sec / 1e12, 2), "peak": MFMA_BF16_PEAK / 1e12,
                    "unit": "TFLOP/s"}
        else:
            roof = {"bound": "hbm", "achieved": round(dom["bytes"] / sec / 1e9, 1), "peak": HBM_PEAK / 1e9, "unit": "GB/s"}
        roof["frac"] = round(roof["achieved"] / roof["peak"], 4)
        # HBM bytes per launch of this kernel from the committed counter passes (FETCH_SIZE x2 + WRITE_SIZE), if any
        roof["traffic"] = None
        try:
            import glob
            files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_hbm_traffic.json")))
            if files and args.batch == 32 and args.size == 512 and args.dims == 2 and args.dtype == "bf16":
                tk = json.load(open(files[-1]))["kernels"].get(dom_name)
                if tk:
                    roof["traffic"] = tk["bytes_per_launch"]
                    roof["traffic_source"] = os.path.basename(files[-1])
        except (OSError, ValueError, KeyError):
            pass
        roof["algorithmic_bytes_per_launch"] = round(dom["bytes"] / max(1, dom["launches"]))
        roof["kernel"] = dom_name
        roof["avg_launch_us"] = round(1e3 * dom["ms"] / dom["launches"], 2)
        roof["launches_per_step"] = round(dom["launches"] / nprof, 1)
        roof["share_of_kernel_time"] = round(dom["ms"] / tot_ms, 3)
        out["roofline"] = roof

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline()

    if (rank == 0 and world == 1 and not args.no_reference_network and args.dims == 2 and args.dtype == "bf16"
            and args.channels == "16,32,64,128,256"):
        # the same step on the channel widths reference/train.py:82 itself builds (four times the convolution work);
        # a side figure, never `value`
        ref_ch = (32, 64, 128, 256, 512)
        torch.manual_seed(0)
        m2 = unet_amd.UNet(2, 1, 1, ref_ch, (2, 2, 2, 2), num_res_units=2, dropout=args.dropout, compute_dtype=dtype).to(dev).train()
        o2 = unet_amd.AdamW(m2.parameters(), lr=5e-4, weight_decay=1e-5, model=m2)
        t2 = unet_amd.Trainer(m2, loss, o2)
        n2 = max(3, min(args.steps, 10))
        for _ in range(3):
            t2.step(x, t)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n2):
            t2.step(x, t)
        torch.cuda.synchronize()
        e2 = time.perf_counter() - t0
        p2 = m2._plan_for(x, True)
        f2 = p2.flops(False) + p2.flops(True)
        out["reference_network"] = {"channels": "-".join(map(str, ref_ch)), "value": round(args.batch * n2 / e2, 2), "unit": "slices/s",
                                    "steps": n2, "ms_per_step": round(1e3 * e2 / n2, 4),
                                    "conv_gflop_per_step_per_gpu": round(f2 / 1e9, 1),
                                    "conv_tflops_whole_step": round(f2 / (e2 / n2) / 1e12, 1)}

    if (rank == 0 and world == 1 and not args.no_secondary and args.dims == 2 and args.dtype == "bf16"
            and (args.size, args.batch, args.channels) == (512, 32, "16,32,64,128,256")):
        # the other single-GPU workloads BASELINE.json names, measured after the timed region (side figures, never `value`)
        del trainer, opt, model, plan
        torch.cuda.empty_cache()
        out["secondary"] = {
            "train_3d_96": secondary_train(3, 96, 4, dev, args.dropout, config="BASELINE.json configs[3]"),
            "infer_sliding_window": secondary_infer(dev),
            "train_2d_256_b8": secondary_train(2, 256, 8, dev, args.dropout, config="shape of BASELINE.json configs[0]: the reference's own batch, reference/config.py:69"),
        }

    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
