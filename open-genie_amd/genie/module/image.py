"""2-D image building blocks of the GAN critic on the HIP kernels (drop-in for reference genie/module/image.py).

An image batch (N, C, H, W) is run as a one-frame video (N, C, 1, H, W) in the CL layout, so the 2-D convolutions are the
kt = 1 case of the gather-GEMM (`genie_conv_igemm` / `genie_conv_wgrad`), GroupNorm + LeakyReLU is one fused streaming pass
(`genie_groupnorm_fwd`, act = 2) and the blur-pool is the kt = 1 case of the channel-sum + stencil kernels.  Parameters keep
``nn.Conv2d``'s shapes and names -- ``weight`` (Cout, Cin, kh, kw), ``bias`` -- so reference checkpoints load.
Every module accepts 4-D (N, C, H, W) or 5-D one-frame input and returns the same rank.
"""
from __future__ import annotations

import math
from typing import Optional, Tuple

import torch
import torch.nn as nn
from torch import Tensor

from .. import functional as GF
from ..cl import to_cl
from ..conv import ConvSpec
from ..utils import default, exists


def _pair(v) -> Tuple[int, int]:
    return (v, v) if isinstance(v, int) else tuple(v)


def as_frames(x: Tensor) -> Tuple[Tensor, bool]:
    """(N, C, H, W) -> CL (N, C, 1, H, W); the flag says whether the caller handed in 4-D."""
    if x.dim() == 4:
        return to_cl(x.unsqueeze(2)), True
    if x.dim() == 5 and x.shape[2] == 1:
        return to_cl(x), False
    raise ValueError(f'expected an image batch (N, C, H, W) or one-frame video (N, C, 1, H, W), got {tuple(x.shape)}')


def like_input(y: Tensor, was_4d: bool) -> Tensor:
    return y.squeeze(2) if was_4d else y


def get_blur_kernel(kernel_size, device=None, dtype=None, norm: bool = True) -> Tensor:
    """Pascal blur taps, reference image.py:16-40 (the w taps take their LENGTH from kernel_size[0], as there)."""
    k0, k1 = _pair(kernel_size)
    a = torch.tensor([math.comb(k0 - 1, i) for i in range(k0)], device=device, dtype=dtype).unsqueeze(-1)
    b = torch.tensor([math.comb(k1 - 1, i) for i in range(k0)], device=device, dtype=dtype).unsqueeze(0)
    k = a @ b
    return k / k.sum() if norm else k


class Conv2d(nn.Module):
    """``nn.Conv2d`` (zero padding, no groups / dilation) as the kt = 1 case of the gather-GEMM."""

    def __init__(self, in_channels: int, out_channels: int, kernel_size, stride=1, padding=0, bias: bool = True) -> None:
        super().__init__()
        kh, kw = _pair(kernel_size)
        sh, sw = _pair(stride)
        ph, pw = _pair(padding)
        w = torch.empty(out_channels, in_channels, kh, kw)
        nn.init.kaiming_uniform_(w, a=math.sqrt(5))                 # nn.Conv2d.reset_parameters
        self.weight = nn.Parameter(w)
        if bias:
            bound = 1 / math.sqrt(in_channels * kh * kw)
            self.bias = nn.Parameter(torch.empty(out_channels).uniform_(-bound, bound))
        else:
            self.register_parameter('bias', None)
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride, self.padding = (kh, kw), (sh, sw), (ph, pw)
        self.spec = ConvSpec(in_channels, out_channels, (1, kh, kw), (1, sh, sw), (1, 1, 1), (0, ph, pw), (0, ph, pw), None)
        self.op = GF.ConvOp(self.spec)

    def forward(self, inp: Tensor, resid: Optional[Tensor] = None) -> Tensor:
        x, was_4d = as_frames(inp)
        r = None if resid is None else as_frames(resid)[0]
        return like_input(GF.conv3d(x, self.weight.unsqueeze(2), self.bias, self.op, r), was_4d)


class LeakyReLU(nn.LeakyReLU):
    def forward(self, inp: Tensor) -> Tensor:
        if inp.dim() == 4:
            x, was_4d = as_frames(inp)
            return like_input(GF.leaky_relu(x, self.negative_slope), was_4d)
        return GF.leaky_relu(inp, self.negative_slope)


class BlurPooling2d(nn.Module):
    """reference image.py:44-84: ``conv2d`` with every (out, in / groups) tap equal to the blur kernel, i.e. every output channel
    is the strided blur of the SUM of its group's input channels: the channel-sum + stencil kernels, once per group."""

    def __init__(self, kernel_size, stride=2, num_groups: int = 1, **kwargs) -> None:
        super().__init__()
        if kwargs:
            raise NotImplementedError(f'BlurPooling2d: extra conv2d arguments {sorted(kwargs)} are not implemented on the HIP path')
        self.register_buffer('blur', get_blur_kernel(kernel_size))
        self.stride, self.kwargs, self.num_groups = stride, kwargs, num_groups
        sh, sw = _pair(stride)
        kh, kw = _pair(kernel_size)
        self.padding = ((kh - 1) // sh, (kw - 1) // sw)

    def forward(self, inp: Tensor) -> Tensor:
        x, was_4d = as_frames(inp)
        sh, sw = _pair(self.stride)
        c, g = x.shape[1], self.num_groups
        if g == 1:
            y = GF.blur_pool3d(x, self.blur.unsqueeze(0), (1, sh, sw), (0, *self.padding), c)
        else:
            # conv2d(groups = g): every output channel of a group is the strided blur of the SUM of that group's input channels -- the one-group
            # kernels per group, on a view of the slice where it is 16-byte aligned and on a copy of it otherwise (as BlurPooling3d, video.py)
            if c % g:
                raise ValueError(f'BlurPooling2d: {c} channels are not divisible into {g} groups (conv2d refuses this in the reference)')
            cg = c // g
            piece = (lambda i: x[:, i * cg:(i + 1) * cg]) if cg % 8 == 0 else (lambda i: to_cl(x[:, i * cg:(i + 1) * cg]))
            y = to_cl(torch.cat([GF.blur_pool3d(piece(i), self.blur.unsqueeze(0), (1, sh, sw), (0, *self.padding), cg) for i in range(g)], dim=1))
        return like_input(y, was_4d)


class _PixelUnshuffle(nn.Module):
    """Stand-in for the reference's Rearrange('b c (h p) (w q) -> b (c p q) h w') at ``go_up.0`` (keeps Sequential indices)."""

    def forward(self, x):
        return x


class SpaceDownsample(nn.Module):
    """reference image.py:86-103: pixel-unshuffle by `factor` then Conv2d(C * f^2 -> C, 1).  The rearrange followed by a 1x1
    convolution IS an f x f convolution of stride f whose weight is the same tensor viewed as (C, C, f, f) -- no data is moved."""

    def __init__(self, in_dim: int, factor: int = 2) -> None:
        super().__init__()
        self.in_dim, self.factor = in_dim, factor
        conv = Conv2d(in_dim * factor ** 2, in_dim, kernel_size=1)               # parameters in the reference's shape
        self.go_up = nn.Sequential(_PixelUnshuffle(), conv)
        self.spec = ConvSpec(in_dim, in_dim, (1, factor, factor), (1, factor, factor), (1, 1, 1), (0, 0, 0), (0, 0, 0), None)
        self.op = GF.ConvOp(self.spec)

    def forward(self, inp: Tensor, resid: Optional[Tensor] = None) -> Tensor:
        x, was_4d = as_frames(inp)
        f, conv = self.factor, self.go_up[1]
        if x.shape[3] % f or x.shape[4] % f:
            raise ValueError(f'SpaceDownsample: image size {tuple(x.shape[3:])} is not divisible by the factor {f}')
        w = conv.weight.view(self.in_dim, self.in_dim, f, f).unsqueeze(2)           # [o][(c p q)] -> [o][c][1][p][q]
        return like_input(GF.conv3d(x, w, conv.bias, self.op, None if resid is None else as_frames(resid)[0]), was_4d)


class _GnLeaky(nn.GroupNorm):
    """``nn.GroupNorm`` whose forward also applies the LeakyReLU that follows it in the reference's Sequential (one pass)."""

    def forward(self, inp: Tensor) -> Tensor:
        return GF.group_norm(inp, self.num_groups, self.weight, self.bias, self.eps, act=2)


class _Fused(nn.Module):
    """Place-holder for an activation that the preceding module already applied (keeps the reference's Sequential indices)."""

    def forward(self, x):
        return x


class ImageResidualBlock(nn.Module):
    """reference image.py:105-163: main(GN, LeakyReLU, Conv2d, GN, LeakyReLU, Conv2d[, SpaceDownsample]) + res(Conv2d 1x1, stride =
    downsample) -- Identity shortcut when out_channel is None.  state_dict keys: main.0/2/3/5(/6.go_up.1), res."""

    def __init__(self, inp_channel: int, out_channel: int | None = None, kernel_size=3, padding=1, num_groups: int = 1,
                 downsample: int | None = None) -> None:
        super().__init__()
        self.res = Conv2d(inp_channel, out_channel, kernel_size=1, stride=default(downsample, 1)) if exists(out_channel) else nn.Identity()
        out_channel = default(out_channel, inp_channel)
        self.main = nn.Sequential(
            _GnLeaky(num_groups, inp_channel), _Fused(),
            Conv2d(inp_channel, out_channel, kernel_size=kernel_size, padding=padding),
            _GnLeaky(num_groups, out_channel), _Fused(),
            Conv2d(out_channel, out_channel, kernel_size=kernel_size, padding=padding),
            *([SpaceDownsample(out_channel, downsample)] if exists(downsample) and downsample else []),
        )
        self.downsample = downsample

    def forward(self, inp: Tensor) -> Tensor:
        x, was_4d = as_frames(inp)
        r = self.res(x)
        y = x
        last = len(self.main) - 1
        for i, layer in enumerate(self.main):
            y = layer(y, resid=r) if i == last else layer(y)      # the residual sum rides in the last GEMM's epilogue
        return like_input(y, was_4d)
