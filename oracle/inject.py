"""Oracle (test infrastructure): run the restated U-Net with *given* dropout masks and *given* PReLU branch
decisions.

Two implementations of the same network cannot be compared element for element through a PReLU kink or a
random dropout mask unless both take the same branch / the same mask.  ``adn_blocks`` lists the
norm-dropout-activation blocks of an oracle ``UNet`` in execution order (the order in which MONAI's ADN
modules run in reference/train.py:115); ``record_preactivations`` returns the input of every PReLU;
``inject`` replaces, for one forward / backward, ``nn.Dropout`` by a fixed mask multiply and ``nn.PReLU`` by
``where(decision, d, alpha * d)`` with the decisions taken from the implementation under test.  The caller
checks separately that those decisions differ from the oracle's own only where ``|d|`` is at rounding level.
"""
from __future__ import annotations

import contextlib
from typing import List, Optional, Sequence

import torch
import torch.nn as nn


def adn_blocks(model: nn.Module, x: torch.Tensor) -> List[nn.Module]:
    """ADN modules (children N, D, A) in the order one forward pass runs them."""
    order = []
    hooks = []
    for mod in model.modules():
        kids = dict(mod.named_children())
        if "A" in kids and "N" in kids and isinstance(kids["A"], nn.PReLU):
            hooks.append(mod.register_forward_pre_hook(lambda m, inp: order.append(m)))
    was = model.training
    model.eval()
    with torch.no_grad():
        model(x)
    model.train(was)
    for h in hooks:
        h.remove()
    return order


def record_preactivations(model: nn.Module, x: torch.Tensor) -> List[torch.Tensor]:
    """Input of every PReLU (after norm and dropout), in execution order, for the model's current mode."""
    blocks = adn_blocks(model, x)
    got = {}
    hooks = [b.A.register_forward_pre_hook(lambda m, inp, i=i: got.__setitem__(i, inp[0].detach().clone()))
             for i, b in enumerate(blocks)]
    with torch.no_grad():
        model(x)
    for h in hooks:
        h.remove()
    return [got[i] for i in range(len(blocks))]


@contextlib.contextmanager
def inject(model: nn.Module, x: torch.Tensor, decisions: Optional[Sequence[torch.Tensor]] = None,
           keep: Optional[Sequence[torch.Tensor]] = None, p: float = 0.0, slope_mass: Optional[dict] = None):
    """Inside the context the k-th block multiplies by ``keep[k] / (1 - p)`` instead of drawing a dropout mask
    (when ``keep`` is given) and takes the PReLU branch ``decisions[k]`` (True: identity side) instead of
    testing the sign (when ``decisions`` is given).  ``slope_mass`` (a dict) receives, per PReLU module, the sum
    of the absolute values of the terms whose signed sum is the gradient of that module's slope: the scale
    against which the rounding error of that one-scalar reduction has to be read when its terms cancel."""
    blocks = adn_blocks(model, x)
    saved = []
    for k, b in enumerate(blocks):
        if keep is not None and hasattr(b, "D"):
            mask = keep[k].contiguous().to(x.dtype) / (1.0 - p)
            saved.append((b.D, b.D.forward))
            b.D.forward = (lambda t, mask=mask: t * mask)
        if decisions is not None:
            dec = decisions[k].contiguous()     # dense channels-first: torch's CPU conv3d backward mishandles channels-last-3d strided inputs
            saved.append((b.A, b.A.forward))

            def fwd(t, dec=dec, act=b.A):
                out = torch.where(dec, t, act.weight * t)
                if slope_mass is not None and out.requires_grad:
                    neg = torch.where(dec, torch.zeros_like(t), t).detach()
                    out.register_hook(lambda g, neg=neg, act=act: slope_mass.__setitem__(act, float((g * neg).abs().sum())))
                return out

            b.A.forward = fwd
    try:
        yield blocks
    finally:
        for mod, fwd in saved:
            del mod.forward          # the instance attribute shadows the class method; removing it restores the original
            assert mod.forward.__func__ is type(mod).forward
