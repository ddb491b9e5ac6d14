"""Fused AdamW over the UNet's flat parameter buffer (reference/train.py:90,114,118 use
``torch.optim.AdamW``): one HIP kernel reads param / grad / exp_avg / exp_avg_sq once and writes three.

Subclasses ``torch.optim.Optimizer`` so that ``StepLR`` (train.py:91,125) and ``zero_grad`` work unchanged.
"""
from __future__ import annotations

import torch

from . import _lib


class AdamW(torch.optim.Optimizer):
    """``torch.optim.AdamW`` semantics (decoupled weight decay, bias correction, no amsgrad).

    ``params`` must be ``model.parameters()`` of one ``unet_amd.UNet`` (pass ``model=`` as well, or the
    UNet is found through the parameters' flat buffer on the first step).
    """

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, amsgrad=False, *, model=None,
                 maximize=False, foreach=None, capturable=False, differentiable=False, fused=None):
        if amsgrad or maximize or differentiable:
            raise NotImplementedError("unet_amd.AdamW: amsgrad / maximize / differentiable are not implemented")
        if lr < 0.0 or eps < 0.0 or weight_decay < 0.0 or not (0 <= betas[0] < 1) or not (0 <= betas[1] < 1):
            raise ValueError("invalid AdamW hyper-parameter")
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        super().__init__(params, defaults)
        if len(self.param_groups) != 1:
            raise NotImplementedError("unet_amd.AdamW supports a single parameter group")
        self._model = model
        self._step = 0
        self._m = None
        self._v = None
        self._gflat = None

    def bind(self, model):
        self._model = model
        self._owned = None
        return self

    def _params_of(self, model):
        """This optimizer's parameter list, checked once to be exactly ``model.parameters()`` (same objects, same
        order).  Like any ``torch.optim`` optimizer it keeps updating these objects, whatever is assigned to the
        module's attributes later."""
        params = self.param_groups[0]["params"]
        if getattr(self, "_owned", None) is not model:
            mp = list(model.parameters())
            if len(params) != len(mp) or any(a is not b for a, b in zip(params, mp)):
                raise RuntimeError("unet_amd.AdamW must own exactly the parameters of its UNet")
            self._owned = model
        return params

    def _flat_grad(self, model, flat):
        """Returns a flat gradient laid out like ``flat``: the buffer the last backward produced when every
        ``p.grad`` still is a view of it, else a gathered copy."""
        g = model._last_grad_flat
        table = model._ptable_ordered
        params = self.param_groups[0]["params"]
        if g is not None and g.device == flat.device:
            base = g.data_ptr()
            if all(p.grad is not None and p.grad.data_ptr() == base + 4 * o for p, (_, _, _, o, _) in zip(params, table)):
                return g
        if self._gflat is None or self._gflat.device != flat.device:
            self._gflat = torch.zeros_like(flat)
        for p, (_, s, n, o, _) in zip(params, table):
            if p.grad is None:
                self._gflat[o:o + n].zero_()
            else:
                self._gflat[o:o + n].view(s).copy_(p.grad)
        return self._gflat

    # ---- checkpointing: torch.optim.AdamW's own format ---------------------------------------------------------
    def state_dict(self):
        """``{"state": {i: {"step", "exp_avg", "exp_avg_sq"}}, "param_groups": [...]}`` as ``torch.optim.AdamW``
        writes it (one entry per parameter, in ``model.parameters()`` order), so the file loads into either."""
        base = super().state_dict()
        state = {}
        if self._m is not None and self._model is not None:
            table = self._model._ptable_ordered
            for i, (_, shp, n, o, _) in enumerate(table):
                state[i] = {"step": torch.tensor(float(self._step)),
                            "exp_avg": self._m[o:o + n].view(shp).clone(),
                            "exp_avg_sq": self._v[o:o + n].view(shp).clone()}
        base["state"] = state
        return base

    def load_state_dict(self, state_dict):
        model = self._model
        if model is None:
            raise RuntimeError("unet_amd.AdamW: call bind(model) or pass model= at construction before load_state_dict")
        state = state_dict.get("state", {})
        super().load_state_dict({"state": {}, "param_groups": state_dict["param_groups"]})
        if not state:
            self._step, self._m, self._v = 0, None, None
            return
        params = self.param_groups[0]["params"]
        flat = model._ensure_flat(params[0].device)
        table = model._ptable_ordered
        if len(state) != len(table):
            raise ValueError(f"optimizer state holds {len(state)} parameters, the UNet has {len(table)}")
        m, v = torch.zeros_like(flat), torch.zeros_like(flat)
        steps = set()
        for i, (name, shp, n, o, _) in enumerate(table):
            st = state[i] if i in state else state[str(i)]
            if tuple(st["exp_avg"].shape) != tuple(shp):
                raise ValueError(f"optimizer state of {name}: shape {tuple(st['exp_avg'].shape)} != {tuple(shp)}")
            m[o:o + n].view(shp).copy_(st["exp_avg"])
            v[o:o + n].view(shp).copy_(st["exp_avg_sq"])
            steps.add(int(float(st["step"])))
        if len(steps) != 1:
            raise ValueError("unet_amd.AdamW keeps one step counter: the loaded parameters disagree on it")
        self._m, self._v, self._step = m, v, steps.pop()

    def _ensure_state(self, flat):
        """Moment buffers laid out like ``flat`` on its device: created at zero on first use; moments loaded (or
        accumulated) while the model lived elsewhere follow it and are never reset."""
        if self._m is None:
            self._m = torch.zeros_like(flat)
            self._v = torch.zeros_like(flat)
        elif self._m.device != flat.device:
            self._m = self._m.to(flat.device)
            self._v = self._v.to(flat.device)

    @torch.no_grad()
    def step(self, closure=None, *, flat_grad=None, grad_scale: float = 1.0, flat=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        model = self._model
        if model is None:
            raise RuntimeError("unet_amd.AdamW: call bind(model) or pass model= at construction")
        if flat is None:                # ``flat``: the caller has just validated the flat parameter buffer itself
            params = self._params_of(model)
            flat = model._ensure_flat(params[0].device, params)
        if not flat.is_cuda:
            raise RuntimeError("unet_amd.AdamW runs on an AMD GPU only (no CPU or eager fallback)")
        g = flat_grad if flat_grad is not None else self._flat_grad(model, flat)
        self._ensure_state(flat)
        grp = self.param_groups[0]
        self._step += 1
        lib = _lib.load()
        _lib.check(lib.mi_adamw_step(flat.data_ptr(), g.data_ptr(), self._m.data_ptr(), self._v.data_ptr(),
                                     flat.numel(), float(grp["lr"]), float(grp["betas"][0]), float(grp["betas"][1]),
                                     float(grp["eps"]), float(grp["weight_decay"]), self._step, float(grad_scale),
                                     _lib.current_stream(flat.device)), "mi_adamw_step")
        return loss
