"""Oracle (test infrastructure): plain-torch restatement of ``monai.networks.nets.UNet``.

Follows the call site reference/train.py:78-86 (``UNet(spatial_dims=2, in_channels=1,
out_channels=1, channels=(32,64,128,256,512), strides=(2,2,2,2), num_res_units=2,
dropout=0.1)``) and MONAI 1.3's ``monai/networks/nets/unet.py``,
``monai/networks/blocks/convolutions.py`` (``Convolution``, ``ResidualUnit``),
``monai/networks/blocks/acti_norm.py`` (``ADN``) and
``monai/networks/layers/simplelayers.py`` (``SkipConnection``).

Module names, creation order (hence RNG draw order of ``reset_parameters``) and
``state_dict`` keys are the ones MONAI produces.  Parity unpinned (see package docstring).
"""
from __future__ import annotations

from typing import Sequence

import numpy as np
import torch
import torch.nn as nn

_CONV = {1: nn.Conv1d, 2: nn.Conv2d, 3: nn.Conv3d}
_CONVT = {1: nn.ConvTranspose1d, 2: nn.ConvTranspose2d, 3: nn.ConvTranspose3d}
_INORM = {1: nn.InstanceNorm1d, 2: nn.InstanceNorm2d, 3: nn.InstanceNorm3d}
_BNORM = {1: nn.BatchNorm1d, 2: nn.BatchNorm2d, 3: nn.BatchNorm3d}


def same_padding(kernel_size: int, dilation: int = 1) -> int:
    # monai.networks.layers.convutils.same_padding
    return (kernel_size - 1) // 2 * dilation


def _norm_layer(norm, dims: int, channels: int):
    name = norm.upper() if isinstance(norm, str) else norm
    if name == "INSTANCE":
        return _INORM[dims](channels)  # torch defaults: eps 1e-5, affine False, no running stats
    if name == "BATCH":
        return _BNORM[dims](channels)
    raise NotImplementedError(f"oracle restates INSTANCE / BATCH norms only, got {norm}")


def _act_layer(act):
    name = act.upper() if isinstance(act, str) else act
    if name == "PRELU":
        return nn.PReLU()  # one shared slope, init 0.25
    if name == "RELU":
        return nn.ReLU()
    raise NotImplementedError(f"oracle restates PRELU / RELU only, got {act}")


class ADN(nn.Sequential):
    """Activation-Dropout-Norm in the order given by ``ordering`` (MONAI ``ADN``)."""

    def __init__(self, ordering="NDA", in_channels=None, act="PRELU", norm="INSTANCE", norm_dim=2, dropout=None):
        super().__init__()
        ops = {"A": None, "D": None, "N": None}
        if norm is not None:
            ops["N"] = _norm_layer(norm, norm_dim, in_channels)
        if act is not None:
            ops["A"] = _act_layer(act)
        if dropout is not None:
            ops["D"] = nn.Dropout(float(dropout))  # dropout_dim=1 -> element-wise nn.Dropout
        for item in ordering.upper():
            if ops[item] is not None:
                self.add_module(item, ops[item])


class Convolution(nn.Sequential):
    """conv (or transposed conv) followed by ADN unless ``conv_only`` (MONAI ``Convolution``)."""

    def __init__(self, spatial_dims, in_channels, out_channels, strides=1, kernel_size=3, adn_ordering="NDA",
                 act="PRELU", norm="INSTANCE", dropout=None, bias=True, conv_only=False, is_transposed=False,
                 padding=None, output_padding=None):
        super().__init__()
        if padding is None:
            padding = same_padding(kernel_size)
        if is_transposed:
            if output_padding is None:
                output_padding = strides - 1  # stride_minus_kernel_padding(1, strides)
            conv = _CONVT[spatial_dims](in_channels, out_channels, kernel_size=kernel_size, stride=strides,
                                        padding=padding, output_padding=output_padding, bias=bias)
        else:
            conv = _CONV[spatial_dims](in_channels, out_channels, kernel_size=kernel_size, stride=strides,
                                       padding=padding, bias=bias)
        self.add_module("conv", conv)
        if conv_only:
            return
        if act is None and norm is None and dropout is None:
            return
        self.add_module("adn", ADN(adn_ordering, out_channels, act, norm, spatial_dims, dropout))


class ResidualUnit(nn.Module):
    """MONAI ``ResidualUnit``: ``conv(x) + residual(x)``."""

    def __init__(self, spatial_dims, in_channels, out_channels, strides=1, kernel_size=3, subunits=2,
                 adn_ordering="NDA", act="PRELU", norm="INSTANCE", dropout=None, bias=True, last_conv_only=False):
        super().__init__()
        self.conv = nn.Sequential()
        self.residual = nn.Identity()
        padding = same_padding(kernel_size)
        schannels, sstrides = in_channels, strides
        subunits = max(1, subunits)
        for su in range(subunits):
            conv_only = last_conv_only and su == (subunits - 1)
            unit = Convolution(spatial_dims, schannels, out_channels, strides=sstrides, kernel_size=kernel_size,
                               adn_ordering=adn_ordering, act=act, norm=norm, dropout=dropout, bias=bias,
                               conv_only=conv_only, padding=padding)
            self.conv.add_module(f"unit{su:d}", unit)
            schannels, sstrides = out_channels, 1
        if np.prod(strides) != 1 or in_channels != out_channels:
            rkernel_size, rpadding = kernel_size, padding
            if np.prod(strides) == 1:  # only adapting channels: 1x1 kernel, no padding
                rkernel_size, rpadding = 1, 0
            self.residual = _CONV[spatial_dims](in_channels, out_channels, rkernel_size, strides, rpadding, bias=bias)

    def forward(self, x):
        res = self.residual(x)
        cx = self.conv(x)
        return cx + res


class SkipConnection(nn.Module):
    """``cat([x, submodule(x)], dim=1)`` (MONAI ``SkipConnection`` mode "cat")."""

    def __init__(self, submodule):
        super().__init__()
        self.submodule = submodule

    def forward(self, x):
        return torch.cat([x, self.submodule(x)], dim=1)


class UNet(nn.Module):
    """Restatement of ``monai.networks.nets.UNet`` (constructor signature kept)."""

    def __init__(self, spatial_dims: int, in_channels: int, out_channels: int, channels: Sequence[int],
                 strides: Sequence[int], kernel_size=3, up_kernel_size=3, num_res_units: int = 0, act="PRELU",
                 norm="INSTANCE", dropout: float = 0.0, bias: bool = True, adn_ordering: str = "NDA"):
        super().__init__()
        if len(channels) < 2:
            raise ValueError("the length of `channels` should be no less than 2.")
        delta = len(strides) - (len(channels) - 1)
        if delta < 0:
            raise ValueError("the length of `strides` should equal to `len(channels) - 1`.")
        self.dimensions = spatial_dims
        self.in_channels, self.out_channels = in_channels, out_channels
        self.channels, self.strides = tuple(channels), tuple(strides)
        self.kernel_size, self.up_kernel_size = kernel_size, up_kernel_size
        self.num_res_units, self.act, self.norm = num_res_units, act, norm
        self.dropout, self.bias, self.adn_ordering = dropout, bias, adn_ordering

        def _create_block(inc, outc, channels, strides, is_top):
            c, s = channels[0], strides[0]
            if len(channels) > 2:
                subblock = _create_block(c, c, channels[1:], strides[1:], False)
                upc = c * 2
            else:
                subblock = self._get_down_layer(c, channels[1], 1, False)  # bottom layer
                upc = c + channels[1]
            down = self._get_down_layer(inc, c, s, is_top)
            up = self._get_up_layer(upc, outc, s, is_top)
            return nn.Sequential(down, SkipConnection(subblock), up)

        self.model = _create_block(in_channels, out_channels, self.channels, self.strides, True)

    def _get_down_layer(self, in_channels, out_channels, strides, is_top):
        if self.num_res_units > 0:
            return ResidualUnit(self.dimensions, in_channels, out_channels, strides=strides,
                                kernel_size=self.kernel_size, subunits=self.num_res_units, act=self.act,
                                norm=self.norm, dropout=self.dropout, bias=self.bias,
                                adn_ordering=self.adn_ordering)
        return Convolution(self.dimensions, in_channels, out_channels, strides=strides, kernel_size=self.kernel_size,
                           act=self.act, norm=self.norm, dropout=self.dropout, bias=self.bias,
                           adn_ordering=self.adn_ordering)

    def _get_up_layer(self, in_channels, out_channels, strides, is_top):
        conv = Convolution(self.dimensions, in_channels, out_channels, strides=strides,
                           kernel_size=self.up_kernel_size, act=self.act, norm=self.norm, dropout=self.dropout,
                           bias=self.bias, conv_only=is_top and self.num_res_units == 0, is_transposed=True,
                           adn_ordering=self.adn_ordering)
        if self.num_res_units > 0:
            ru = ResidualUnit(self.dimensions, out_channels, out_channels, strides=1, kernel_size=self.kernel_size,
                              subunits=1, act=self.act, norm=self.norm, dropout=self.dropout, bias=self.bias,
                              last_conv_only=is_top, adn_ordering=self.adn_ordering)
            conv = nn.Sequential(conv, ru)
        return conv

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self.model(x)
