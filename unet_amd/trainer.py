"""Fused training step: the body of the reference's inner loop (reference/train.py:111-119) as one
sequence of HIP launches on one stream with no host synchronisation and no Python per layer:

    labels==2 -> 1 (folded into the loss read) -> forward -> Dice(+CE) loss -> backward
    -> bucketed gradient all-reduce (overlapped, multi-GPU only) -> AdamW

``loss.item()`` per step (train.py:119) is replaced by a device-side running sum read once per epoch.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _lib
from .losses import DiceCELoss
from .nets import UNet
from .optim import AdamW
from .parallel import BucketedAllReduce


class Trainer:
    def __init__(self, model: UNet, loss: DiceCELoss, optimizer: AdamW, *, group=None, overlap: bool = True,
                 force_reducer: bool = False, split_calls: bool = False):
        """``force_reducer``: run the bucketed all-reduce path even in a process group of one rank (it then
        exercises the bucket-by-bucket backward, the communication stream and its events; the sums are unchanged).
        ``split_calls``: on a single device, enqueue the step as separate library calls (forward, loss, backward,
        AdamW) instead of the one-call ``mi_unet_train_step``; the results are the same bit for bit."""
        self.split_calls = split_calls
        self.model, self.loss, self.opt = model, loss, optimizer
        optimizer.bind(model)
        self.group = group
        self.overlap = overlap
        self.force_reducer = force_reducer
        self._bufs = {}
        self._reducer = None
        self._grad = None
        import torch.distributed as dist

        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0

    def _buffers(self, x):
        key = (tuple(x.shape), x.device)
        b = self._bufs.get(key)
        if b is None:
            dev = x.device
            plan = self.model._plan_for(x, True)
            b = dict(plan=plan, logits=torch.empty_like(x), dlogits=torch.empty_like(x),
                     sums=torch.empty(x.shape[0] * 4, dtype=torch.float64, device=dev),
                     loss=torch.zeros(1, dtype=torch.float32, device=dev),
                     ws=torch.empty(plan.workspace_bytes, dtype=torch.uint8, device=dev))
            self._bufs[key] = b
        return b

    def loss_sum(self) -> float:
        """Sum of the step losses since the last call (synchronises once)."""
        if getattr(self, "_loss_accum", None) is None:
            return 0.0
        v = float(self._loss_accum.item())
        self._loss_accum.zero_()
        return v

    def last_logits(self, x) -> torch.Tensor:
        return self._buffers(x)["logits"]

    @torch.no_grad()
    def step(self, x: torch.Tensor, labels: torch.Tensor) -> None:
        model = self.model
        model._check_input(x)
        if not model.training:
            raise RuntimeError("Trainer.step needs model.train()")
        lib = _lib.load()
        x = x.contiguous()
        if x.dtype != torch.float32:
            x = x.float()
        labels = labels.contiguous()
        tcode = _lib.label_dtype_code(labels)
        if tcode < 0:
            labels = labels.float()
            tcode = 0
        if labels.shape != x.shape:
            raise AssertionError(f"ground truth has different shape ({labels.shape}) from input ({x.shape})")
        dev = x.device
        flat = model._ensure_flat(dev, self.opt._params_of(model))
        b = self._buffers(x)
        plan = b["plan"]
        if self._grad is None or self._grad.device != dev:
            self._grad = torch.zeros_like(flat)
            self._loss_accum = torch.zeros(1, dtype=torch.float32, device=dev)
            self._reducer = (BucketedAllReduce(self._grad, model.gradient_buckets(), self.group, force=self.force_reducer)
                             if (self.world > 1 or self.force_reducer) else None)
        st = _lib.current_stream(dev)
        # every rank draws its own dropout masks
        seed = (model._next_seed() + 0x9E3779B97F4A7C15 * self.rank) & 0xFFFFFFFFFFFFFFFF
        L = self.loss
        if self._reducer is None and not self.split_calls:
            # single device: the whole step is one call into the library (mi_unet_train_step)
            opt = self.opt
            opt._ensure_state(flat)
            opt._step += 1
            grp = opt.param_groups[0]
            a = b.get("args")
            if a is None:
                a = b["args"] = _lib.MiTrainStepArgs()
                a.logits, a.dlogits, a.loss_sums = b["logits"].data_ptr(), b["dlogits"].data_ptr(), b["sums"].data_ptr()
                a.loss, a.workspace, a.do_adamw, a.grad_scale = b["loss"].data_ptr(), b["ws"].data_ptr(), 1, 1.0
            a.params, a.grads, a.loss_accum = flat.data_ptr(), self._grad.data_ptr(), self._loss_accum.data_ptr()
            a.exp_avg, a.exp_avg_sq = opt._m.data_ptr(), opt._v.data_ptr()
            a.x, a.target, a.target_dtype, a.fold2 = x.data_ptr(), labels.data_ptr(), tcode, 1 if L.fold_label_two else 0
            a.smooth_nr, a.smooth_dr, a.lambda_dice, a.lambda_ce = L.smooth_nr, L.smooth_dr, L.lambda_dice, L.lambda_ce
            a.lr, (a.beta1, a.beta2), a.eps, a.weight_decay = grp["lr"], grp["betas"], grp["eps"], grp["weight_decay"]
            a.step, a.seed = opt._step, seed
            _lib.check(lib.mi_unet_train_step(plan.h, C.byref(a), st), "mi_unet_train_step")
            return
        nb, npix = x.shape[0], x[0].numel()
        plan.forward(flat, x, b["logits"], b["ws"], True, seed)
        _lib.check(lib.mi_dice_ce_loss_fwd(b["logits"].data_ptr(), labels.data_ptr(), tcode, 1 if L.fold_label_two else 0,
                                           nb, npix, L.smooth_nr, L.smooth_dr, L.lambda_dice, L.lambda_ce,
                                           b["sums"].data_ptr(), b["loss"].data_ptr(), self._loss_accum.data_ptr(), st),
                   "mi_dice_ce_loss_fwd")
        _lib.check(lib.mi_dice_ce_loss_bwd(b["logits"].data_ptr(), labels.data_ptr(), tcode, 1 if L.fold_label_two else 0,
                                           nb, npix, L.smooth_nr, L.smooth_dr, L.lambda_dice, L.lambda_ce,
                                           b["sums"].data_ptr(), 0, 1.0, b["dlogits"].data_ptr(), st),
                   "mi_dice_ce_loss_bwd")
        if self._reducer is not None and self.overlap:
            # the communication stream, not the compute stream, waits for a bucket's weight gradients and reduction:
            # the compute stream goes on with the next bucket's chain; reducer.wait() joins everything before AdamW
            comm = self._reducer.comm_stream
            for bucket in range(len(self._reducer.buckets)):
                plan.backward(flat, x, b["dlogits"], self._grad, b["ws"], True, seed, bucket, ready_stream=comm)
                self._reducer.launch(bucket)
            self._reducer.wait()
        else:
            plan.backward(flat, x, b["dlogits"], self._grad, b["ws"], True, seed, -1)
            if self._reducer is not None:
                for bucket in range(len(self._reducer.buckets)):
                    self._reducer.launch(bucket)
                self._reducer.wait()
        self.opt.step(flat_grad=self._grad, grad_scale=1.0 / self.world, flat=flat)
