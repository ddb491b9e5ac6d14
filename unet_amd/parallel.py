"""Data-parallel pieces (new capability: the reference is single-device, reference/train.py:75).

One process per GPU, ``torch.distributed`` (backend "nccl" = RCCL over xGMI on ROCm; "gloo" on CPU for
tests).  Training: the batch is sharded, every rank holds a replica, gradients live in ONE flat
fp32 buffer that the engine fills in four buckets in backward-completion order (decoder, bottom,
encoder levels 1 and deeper, encoder level 0); each bucket is all-reduced on a side stream as soon as its part of the backward has been
enqueued, overlapping the rest of the backward.  Instance norm needs no cross-rank statistics.
Inference: independent slices are sharded; the only exchange is the final sum of Dice partials.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def init_from_env(backend: Optional[str] = None) -> Tuple[int, int, int]:
    """Initialises the default process group from RANK / WORLD_SIZE / MASTER_* (torchrun's environment).
    Returns (rank, world_size, local_rank); a plain single-process run gives (0, 1, 0)."""
    import os

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group(backend, rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world, local


class BucketedAllReduce:
    """All-reduce (sum) of contiguous buckets of one flat gradient buffer, asynchronously.

    ``launch(i)`` must be called after the work that produces bucket ``i`` has been enqueued on the
    current stream; ``wait()`` makes the current stream (GPU) or the host (CPU tensors) wait for all
    launched buckets.  Averaging is left to the consumer (the fused AdamW takes ``grad_scale``).

    Every collective is issued with ``async_op=True`` and only ``wait()`` waits: ``launch`` never blocks the
    host, whatever the backend, so the kernels of the following buckets are enqueued without waiting for a
    peer (with a backend whose collectives complete on the host, such as gloo on device tensors, a blocking
    call per bucket would stall the enqueue of the rest of the backward pass behind the slowest rank).
    ``force=True`` runs the collectives even in a group of one rank (tests of the stream / event ordering).
    """

    def __init__(self, flat: torch.Tensor, buckets: Sequence[Tuple[int, int]], group=None, force: bool = False):
        self.flat = flat
        self.buckets = [(int(o), int(c)) for o, c in buckets]
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.force = bool(force) and dist.is_initialized()
        self._works: List = []
        self._comm_stream = torch.cuda.Stream(flat.device) if flat.is_cuda else None
        # A backend that reduces on the host (gloo) gets host tensors: each bucket is copied to pinned memory on the
        # communication stream as soon as it is complete, reduced there by wait(), and copied back.  torch's own gloo
        # path for device tensors does the same staging from worker threads of its own; two ranks sharing one GPU
        # hung in it now and then (both hosts in wait(), no kernel of ours pending).
        self._staged = bool(flat.is_cuda and dist.is_initialized() and dist.get_backend(group) == "gloo")
        self._pinned: List = []
        self._copied: List = []

    @property
    def comm_stream(self):
        """The stream the collectives are issued under (``None`` on CPU tensors)."""
        return self._comm_stream

    def launch(self, i: int) -> None:
        if self.world == 1 and not self.force:
            return
        o, c = self.buckets[i]
        if c == 0:
            return
        view = self.flat[o:o + c]
        if self._staged:
            while len(self._pinned) <= i:
                self._pinned.append(None)
            if self._pinned[i] is None or self._pinned[i].numel() != c:
                self._pinned[i] = torch.empty(c, dtype=self.flat.dtype, pin_memory=True)
            ready = torch.cuda.Event()
            ready.record(torch.cuda.current_stream(self.flat.device))
            with torch.cuda.stream(self._comm_stream):
                self._comm_stream.wait_event(ready)
                self._pinned[i].copy_(view, non_blocking=True)
                copied = torch.cuda.Event()
                copied.record(self._comm_stream)
            self._copied.append((i, copied))
        elif self._comm_stream is not None:
            ready = torch.cuda.Event()
            ready.record(torch.cuda.current_stream(self.flat.device))
            with torch.cuda.stream(self._comm_stream):
                self._comm_stream.wait_event(ready)
                self._works.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        else:
            self._works.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def wait(self) -> None:
        if not self._works and self._comm_stream is None:
            return
        if self._staged:
            # in launch order: the later buckets are still being computed while the first ones are reduced
            for i, copied in self._copied:
                copied.synchronize()
                dist.all_reduce(self._pinned[i], op=dist.ReduceOp.SUM, group=self.group)
                o, c = self.buckets[i]
                with torch.cuda.stream(self._comm_stream):
                    self.flat[o:o + c].copy_(self._pinned[i], non_blocking=True)
            self._copied = []
            done = torch.cuda.Event()
            done.record(self._comm_stream)
            torch.cuda.current_stream(self.flat.device).wait_event(done)
            return
        if self._comm_stream is not None:       # also without collectives: the stream may carry waits the caller handed it
            with torch.cuda.stream(self._comm_stream):
                for w in self._works:
                    w.wait()              # RCCL: the communication stream waits; host-side backends: the host does
                done = torch.cuda.Event()
                done.record(self._comm_stream)
            torch.cuda.current_stream(self.flat.device).wait_event(done)
        else:
            for w in self._works:
                w.wait()
        self._works = []


def shard_batch(n: int, rank: int, world: int) -> Tuple[int, int]:
    """[lo, hi) of a global batch / slice list owned by ``rank`` (contiguous, sizes differ by at most 1)."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def all_reduce_dice_partials(partials: torch.Tensor, group=None) -> torch.Tensor:
    """Sums per-volume Dice partials ([..., k] tensors of slice-dice sums and counts, or tp/fp/fn counts)
    over the ranks: the one collective of sharded inference."""
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(partials, op=dist.ReduceOp.SUM, group=group)
    return partials


def broadcast_parameters(flat: torch.Tensor, src: int = 0, group=None) -> None:
    """Makes every replica start from rank ``src``'s parameters."""
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.broadcast(flat, src=src, group=group)
