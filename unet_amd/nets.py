"""``UNet`` with the constructor, module tree and ``state_dict`` layout of ``monai.networks.nets.UNet``
(the class reference/train.py:78-86 instantiates), whose ``forward`` and backward run as hand-written
gfx950 kernels behind the C ABI of ``libunet_mi355.so``.

The ``torch.nn`` sub-modules created here only *hold* parameters (so that ``.to()``,
``.parameters()``, ``.state_dict()``, ``load_state_dict`` and checkpoint files behave as in the
reference, train.py:86,90,157); none of them is ever executed.  There is no eager fallback:
calling the model on a CPU tensor, or without the built library, raises.
"""
from __future__ import annotations

import ctypes as C
import os
import threading
import weakref
from collections import OrderedDict
from typing import Sequence

import torch
import torch.nn as nn

from . import _lib

_CONV = {2: nn.Conv2d, 3: nn.Conv3d}
_CONVT = {2: nn.ConvTranspose2d, 3: nn.ConvTranspose3d}
_INORM = {2: nn.InstanceNorm2d, 3: nn.InstanceNorm3d}


class _Holder(nn.Module):
    """A container of parameters; its arithmetic lives in the HIP engine."""

    def forward(self, *a, **k):  # pragma: no cover - guard
        raise RuntimeError("unet_amd sub-modules only hold parameters; call the UNet itself")


class _Seq(_Holder):
    """Sequential-like container (numbered children), never executed."""

    def __init__(self, *mods):
        super().__init__()
        for i, m in enumerate(mods):
            self.add_module(str(i), m)

    def __getitem__(self, i):
        return self._modules[str(i)]

    def __len__(self):
        return len(self._modules)


class ADN(_Holder):
    def __init__(self, dims, channels, dropout):
        super().__init__()
        self.add_module("N", _INORM[dims](channels))
        self.add_module("D", nn.Dropout(float(dropout)))
        self.add_module("A", nn.PReLU())


class Convolution(_Holder):
    def __init__(self, dims, cin, cout, strides=1, kernel_size=3, dropout=0.0, conv_only=False, is_transposed=False):
        super().__init__()
        pad = (kernel_size - 1) // 2
        if is_transposed:
            conv = _CONVT[dims](cin, cout, kernel_size=kernel_size, stride=strides, padding=pad,
                                output_padding=strides - 1, bias=True)
        else:
            conv = _CONV[dims](cin, cout, kernel_size=kernel_size, stride=strides, padding=pad, bias=True)
        self.add_module("conv", conv)
        if not conv_only:
            self.add_module("adn", ADN(dims, cout, dropout))


class ResidualUnit(_Holder):
    def __init__(self, dims, cin, cout, strides=1, subunits=2, dropout=0.0, last_conv_only=False):
        super().__init__()
        self.conv = _Holder()
        self.residual = nn.Identity()
        sc, ss = cin, strides
        subunits = max(1, subunits)
        for su in range(subunits):
            conv_only = last_conv_only and su == subunits - 1
            self.conv.add_module(f"unit{su:d}", Convolution(dims, sc, cout, strides=ss, dropout=dropout, conv_only=conv_only))
            sc, ss = cout, 1
        if strides != 1 or cin != cout:
            rk, rp = (3, 1) if strides != 1 else (1, 0)
            self.residual = _CONV[dims](cin, cout, rk, strides, rp, bias=True)


class SkipConnection(_Holder):
    def __init__(self, submodule):
        super().__init__()
        self.submodule = submodule


_DT = {torch.float32: 0, torch.bfloat16: 1}


class _Plan:
    """One engine instance: fixed batch, spatial size, dtype and direction set."""

    def __init__(self, cfg_kwargs, batch, dims, dtype_code, with_backward):
        lib = _lib.load()
        cfg = _lib.MiUnetCfg()
        cfg.spatial_dims = cfg_kwargs["spatial_dims"]
        cfg.in_channels = cfg_kwargs["in_channels"]
        cfg.out_channels = cfg_kwargs["out_channels"]
        ch = cfg_kwargs["channels"]
        cfg.n_levels = len(ch)
        for i, v in enumerate(ch):
            cfg.channels[i] = int(v)
        for i, v in enumerate(cfg_kwargs["strides"][: len(ch) - 1]):
            cfg.strides[i] = int(v)
        cfg.num_res_units = cfg_kwargs["num_res_units"]
        cfg.dropout = float(cfg_kwargs["dropout"])
        cfg.dtype = dtype_code
        cfg.batch = batch
        d3 = (1,) * (3 - len(dims)) + tuple(int(v) for v in dims)
        for i in range(3):
            cfg.dims[i] = d3[i]
        cfg.with_backward = 1 if with_backward else 0
        h = C.c_void_p()
        _lib.check(lib.mi_unet_create(C.byref(cfg), C.byref(h)), "mi_unet_create")
        self.lib, self.h = lib, h
        self.batch, self.dims, self.with_backward = batch, tuple(dims), with_backward
        self.workspace_bytes = lib.mi_unet_workspace_bytes(h)
        self.nfloats = lib.mi_unet_param_floats(h)
        self._pool = []  # free workspaces
        self._lock = threading.Lock()

    def __del__(self):
        try:
            if self.h:
                self.lib.mi_unet_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def param_table(self):
        n = self.lib.mi_unet_num_params(self.h)
        out = []
        for i in range(n):
            info = _lib.MiParamInfo()
            _lib.check(self.lib.mi_unet_param_info(self.h, i, C.byref(info)))
            out.append((info.name.decode(), tuple(info.shape[: info.ndim]), int(info.numel), int(info.offset), int(info.bucket)))
        return out

    def bucket_ranges(self):
        out = []
        for b in range(self.lib.mi_unet_num_buckets(self.h)):
            o, c = C.c_int64(), C.c_int64()
            _lib.check(self.lib.mi_unet_bucket_range(self.h, b, C.byref(o), C.byref(c)))
            out.append((o.value, c.value))
        return out

    def flops(self, backward=False):
        return self.lib.mi_unet_flops(self.h, 1 if backward else 0)

    def steps(self, backward=False):
        n = self.lib.mi_unet_num_steps(self.h, 1 if backward else 0)
        out = []
        for i in range(n):
            info = _lib.MiStepInfo()
            _lib.check(self.lib.mi_unet_step_info(self.h, 1 if backward else 0, i, C.byref(info)))
            out.append(dict(layer=info.layer.decode(), kernel=info.kernel.decode(), work=info.work,
                            work_is_flops=bool(info.work_is_flops), bytes=info.bytes, total_ms=info.total_ms,
                            launches=info.launches))
        return out

    def take_workspace(self, device):
        with self._lock:
            if self._pool:
                return self._pool.pop()
        return torch.empty(self.workspace_bytes, dtype=torch.uint8, device=device)

    def give_workspace(self, ws):
        """Keeps at most one idle workspace per plan; the rest goes back to torch's allocator."""
        with self._lock:
            if not self._pool:
                self._pool.append(ws)

    def drop_workspaces(self):
        with self._lock:
            self._pool.clear()

    def forward(self, flat, x, logits, ws, training, seed):
        _lib.check(self.lib.mi_unet_forward(self.h, flat.data_ptr(), x.data_ptr(), logits.data_ptr(), ws.data_ptr(),
                                            1 if training else 0, C.c_uint64(seed), _lib.current_stream(x.device)),
                   "mi_unet_forward")

    def backward(self, flat, x, dlogits, flat_grad, ws, training, seed, bucket=-1, ready_stream=None):
        """``ready_stream`` (a ``torch.cuda.Stream``, with ``bucket >= 0``): that stream, not the current one, waits for
        the bucket's weight gradients and gradient reduction (``mi_unet_backward_bucket``)."""
        if ready_stream is not None and bucket >= 0:
            _lib.check(self.lib.mi_unet_backward_bucket(self.h, flat.data_ptr(), x.data_ptr(), dlogits.data_ptr(), flat_grad.data_ptr(),
                                                        ws.data_ptr(), 1 if training else 0, C.c_uint64(seed), bucket,
                                                        _lib.current_stream(dlogits.device), C.c_void_p(ready_stream.cuda_stream)),
                       "mi_unet_backward_bucket")
            return
        _lib.check(self.lib.mi_unet_backward(self.h, flat.data_ptr(), x.data_ptr(), dlogits.data_ptr(), flat_grad.data_ptr(),
                                             ws.data_ptr(), 1 if training else 0, C.c_uint64(seed), bucket,
                                             _lib.current_stream(dlogits.device)),
                   "mi_unet_backward")


class _WorkspaceLease:
    """Returns a workspace to its plan when the autograd node that used it is released."""

    def __init__(self, plan, ws):
        self.plan, self.ws = plan, ws

    def __del__(self):
        try:
            self.plan.give_workspace(self.ws)
        except Exception:
            pass


class _UNetFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, need_bwd, x, *params):
        plan = model._plan_for(x, need_bwd)
        flat = model._flat
        logits = torch.empty_like(x)
        seed = model._next_seed()
        ws = plan.take_workspace(x.device)
        plan.forward(flat, x, logits, ws, model.training, seed)
        if need_bwd:
            ctx.plan, ctx.seed, ctx.model = plan, seed, weakref.ref(model)
            ctx.lease = _WorkspaceLease(plan, ws)
            ctx.training = model.training
            ctx.x = x
        else:
            plan.give_workspace(ws)
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        model = ctx.model()
        plan = ctx.plan
        if model is None:
            raise RuntimeError("the UNet that produced this output no longer exists")
        dlogits = dlogits.contiguous()
        if dlogits.dtype != torch.float32:
            dlogits = dlogits.float()
        # zero-filled: parameter offsets are padded to 16 bytes and the engine never writes the gaps
        g = torch.zeros_like(model._flat)
        plan.backward(model._flat, ctx.x, dlogits, g, ctx.lease.ws, ctx.training, ctx.seed)
        model._last_grad_flat = g
        grads = tuple(g[o:o + n].view(s) for (_, s, n, o, _) in model._ptable_ordered)
        ctx.lease = None
        ctx.x = None
        return (None, None, None) + grads


class UNet(nn.Module):
    """Residual U-Net, MONAI constructor signature (reference/train.py:78-86).

    Extra keyword ``compute_dtype`` (``torch.bfloat16`` default, or ``torch.float32``) selects the
    storage / MFMA input type of the hidden activations; parameters, logits and gradients are fp32
    either way.  Restrictions of the HIP path: ``in_channels == out_channels == 1``, ``kernel_size ==
    up_kernel_size == 3``, all strides 2, ``num_res_units >= 1``, PReLU + instance norm in "NDA" order,
    channels multiples of 16, spatial sizes divisible by ``2**(len(channels)-1)``.
    """

    def __init__(self, spatial_dims: int, in_channels: int, out_channels: int, channels: Sequence[int],
                 strides: Sequence[int], kernel_size=3, up_kernel_size=3, num_res_units: int = 0, act="PRELU",
                 norm="INSTANCE", dropout: float = 0.0, bias: bool = True, adn_ordering: str = "NDA",
                 compute_dtype=None):
        super().__init__()
        if len(channels) < 2:
            raise ValueError("the length of `channels` should be no less than 2.")
        delta = len(strides) - (len(channels) - 1)
        if delta < 0:
            raise ValueError("the length of `strides` should equal to `len(channels) - 1`.")
        act_name = act.upper() if isinstance(act, str) else act
        norm_name = norm.upper() if isinstance(norm, str) else norm
        unsupported = []
        if spatial_dims not in (2, 3): unsupported.append("spatial_dims must be 2 or 3")
        if kernel_size != 3 or up_kernel_size != 3: unsupported.append("kernel sizes must be 3")
        if act_name != "PRELU": unsupported.append("act must be PRELU")
        if norm_name != "INSTANCE": unsupported.append("norm must be INSTANCE")
        if adn_ordering.upper() != "NDA": unsupported.append("adn_ordering must be NDA")
        if not bias: unsupported.append("bias must be True")
        if num_res_units < 1: unsupported.append("num_res_units must be >= 1")
        if in_channels != 1 or out_channels != 1: unsupported.append("in_channels and out_channels must be 1")
        if any(int(s) != 2 for s in strides[: len(channels) - 1]): unsupported.append("strides must all be 2")
        if any(int(c) % 16 for c in channels): unsupported.append("channels must be multiples of 16")
        if unsupported:
            raise NotImplementedError("unet_amd.UNet (HIP path): " + "; ".join(unsupported))
        self.dimensions = spatial_dims
        self.in_channels, self.out_channels = in_channels, out_channels
        self.channels, self.strides = tuple(int(c) for c in channels), tuple(int(s) for s in strides)
        self.kernel_size, self.up_kernel_size = kernel_size, up_kernel_size
        self.num_res_units, self.act, self.norm = num_res_units, act, norm
        self.dropout, self.bias, self.adn_ordering = float(dropout or 0.0), bias, adn_ordering
        if compute_dtype is None:
            compute_dtype = {"fp32": torch.float32, "float32": torch.float32}.get(
                os.environ.get("UNET_AMD_DTYPE", "bf16").lower(), torch.bfloat16)
        if compute_dtype not in _DT:
            raise ValueError("compute_dtype must be torch.bfloat16 or torch.float32")
        self.compute_dtype = compute_dtype

        dims = spatial_dims

        def down_layer(cin, cout, s):
            return ResidualUnit(dims, cin, cout, strides=s, subunits=num_res_units, dropout=self.dropout)

        def up_layer(cin, cout, s, is_top):
            conv = Convolution(dims, cin, cout, strides=s, dropout=self.dropout, is_transposed=True)
            ru = ResidualUnit(dims, cout, cout, strides=1, subunits=1, dropout=self.dropout, last_conv_only=is_top)
            return _Seq(conv, ru)

        def create_block(inc, outc, chs, strs, is_top):  # same creation order as MONAI: sub-block, down, up
            c, s = chs[0], strs[0]
            if len(chs) > 2:
                sub = create_block(c, c, chs[1:], strs[1:], False)
                upc = c * 2
            else:
                sub = down_layer(c, chs[1], 1)
                upc = c + chs[1]
            down = down_layer(inc, c, s)
            up = up_layer(upc, outc, s, is_top)
            return _Seq(down, SkipConnection(sub), up)

        self.model = create_block(in_channels, out_channels, self.channels, self.strides, True)

        self._cfg_kwargs = dict(spatial_dims=spatial_dims, in_channels=in_channels, out_channels=out_channels,
                                channels=self.channels, strides=self.strides, num_res_units=num_res_units,
                                dropout=self.dropout)
        self._plans = OrderedDict()     # least recently used first; bounded by MAX_PLANS
        self._flat = None
        self._ptable = None           # state_dict order: (name, shape, numel, offset, bucket)
        self._ptable_ordered = None   # same entries in self.parameters() order
        self._last_grad_flat = None
        self._seed_base = int(torch.initial_seed()) & 0xFFFFFFFFFFFF
        self._seed_ctr = 0

    # ---- engine plumbing ------------------------------------------------------------------------
    def _next_seed(self) -> int:
        self._seed_ctr += 1
        return ((self._seed_base * 0x9E3779B97F4A7C15) + self._seed_ctr) & 0xFFFFFFFFFFFFFFFF

    def seed_dropout(self, seed: int) -> None:
        """Re-seeds the stateless dropout stream (``set_determinism`` calls this through the global seed)."""
        self._seed_base, self._seed_ctr = int(seed) & 0xFFFFFFFFFFFF, 0

    def _layout(self):
        if self._ptable is None:
            div = 2 ** (len(self.channels) - 1)
            plan = _Plan(self._cfg_kwargs, 1, (div,) * self.dimensions, _DT[self.compute_dtype], False)
            self._ptable = plan.param_table()
            self._nfloats = plan.nfloats
            self._buckets = plan.bucket_ranges()
            by_name = {e[0]: e for e in self._ptable}
            names = [n for n, _ in self.named_parameters()]
            missing = [n for n in names if n not in by_name]
            if missing or len(names) != len(by_name):
                raise RuntimeError(f"parameter layout mismatch between the module tree and the engine: {missing[:3]}")
            for n, p in self.named_parameters():
                if tuple(p.shape) != by_name[n][1]:
                    raise RuntimeError(f"shape mismatch for {n}: {tuple(p.shape)} vs {by_name[n][1]}")
            self._ptable_ordered = [by_name[n] for n in names]
            # where every parameter lives, in parameters() order: (owning module, attribute name).  The module tree is
            # fixed by the constructor (the engine's layout follows it); reading the slots returns the Parameter
            # objects now assigned, without walking the tree (130 us of host time per walk).
            owners = dict(self.named_modules())
            self._param_slots = [(owners[n.rpartition(".")[0]], n.rpartition(".")[2]) for n in names]
        return self._ptable

    def _params_now(self):
        """The model's Parameter objects in ``parameters()`` order."""
        self._layout()
        return [m._parameters[k] for m, k in self._param_slots]

    def _ensure_flat(self, device, params=None):
        """Keeps every parameter a view into one flat fp32 buffer (what the engine and AdamW read).  ``params``: the
        model's Parameter objects in ``parameters()`` order when the caller already holds them (an optimizer does, as
        ``torch.optim`` optimizers do: walking the module tree costs more host time than enqueueing ten kernels)."""
        self._layout()
        if params is None:
            params = list(self.parameters())
        flat = self._flat
        ok = flat is not None and flat.device == device
        if ok:
            base = flat.data_ptr()
            for p, (_, _, n, o, _) in zip(params, self._ptable_ordered):
                if p.data_ptr() != base + 4 * o or p.dtype != torch.float32:
                    ok = False
                    break
        if not ok:
            new = torch.zeros(self._nfloats, dtype=torch.float32, device=device)
            with torch.no_grad():
                for p, (_, s, n, o, _) in zip(params, self._ptable_ordered):
                    view = new[o:o + n].view(s)
                    view.copy_(p.data.to(device=device, dtype=torch.float32))
                    p.data = view
            self._flat = new
        return self._flat

    def flat_parameters(self) -> torch.Tensor:
        """The flat fp32 parameter buffer (parameters are views into it); needs a GPU-resident model."""
        dev = next(self.parameters()).device
        return self._ensure_flat(dev)

    def gradient_buckets(self):
        """[(offset, count)] of the gradient buckets in backward-completion order (decoder, bottom, encoder levels
        1 and deeper, encoder level 0); a bucket may be empty."""
        self._layout()
        return list(self._buckets)

    MAX_PLANS = 12    # engine plans (one per batch size / spatial size / direction set) kept per model: sliding-window
                      # inference alone asks for the batch sizes 1, 2, 4 ... 256 and 288 (ten plans)

    def _plan_for(self, x, with_backward):
        key = (x.shape[0], tuple(x.shape[2:]), _DT[self.compute_dtype], bool(with_backward))
        plans = self._plans
        plan = plans.get(key)
        if plan is None and not with_backward:
            key_bwd = key[:3] + (True,)
            if key_bwd in plans:
                key, plan = key_bwd, plans[key_bwd]
        if plan is None:
            plan = _Plan(self._cfg_kwargs, key[0], key[1], key[2], with_backward)
            plans[key] = plan
            limit = max(1, int(os.environ.get("UNET_AMD_MAX_PLANS", self.MAX_PLANS)))
            while len(plans) > limit:
                _, old = plans.popitem(last=False)      # its engine and idle workspace go when the last user lets go
                old.drop_workspaces()
        else:
            plans.move_to_end(key)
        return plan

    def _check_input(self, x):
        if not isinstance(x, torch.Tensor):
            raise TypeError("UNet expects a torch.Tensor")
        if x.dim() != self.dimensions + 2 or x.shape[1] != self.in_channels:
            raise ValueError(f"expected input of shape [B,{self.in_channels}," + ",".join(["*"] * self.dimensions) + f"], got {tuple(x.shape)}")
        if not x.is_cuda:
            raise RuntimeError("unet_amd.UNet runs on an AMD GPU only (no CPU or eager fallback): move the model "
                               "and its input to 'cuda'")
        div = 2 ** (len(self.channels) - 1)
        if any(int(s) % div for s in x.shape[2:]):
            raise ValueError(f"every spatial size must be divisible by {div}, got {tuple(x.shape[2:])}")

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        self._check_input(x)
        _lib.load()
        xin = x.as_subclass(torch.Tensor) if type(x) is not torch.Tensor else x
        in_dtype = xin.dtype
        xin = xin.contiguous()
        if xin.dtype != torch.float32:
            xin = xin.float()
        params = self._params_now()
        self._ensure_flat(xin.device, params)
        need_bwd = torch.is_grad_enabled() and any(p.requires_grad for p in params)
        # without a backward pass the parameters need not travel through the autograd function's argument list
        out = _UNetFunction.apply(self, need_bwd, xin, *params) if need_bwd else _UNetFunction.apply(self, False, xin)
        return out if in_dtype == torch.float32 else out.to(in_dtype)
