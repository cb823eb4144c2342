"""GAN critic of the tokenizer on the HIP kernels (drop-in for reference genie/module/discriminator.py:17-114).

``FrameDiscriminator``: Conv2d stem -> ImageResidualBlocks (GroupNorm + LeakyReLU + 3x3 convs, pixel-unshuffle downsampling) ->
Conv2d + LeakyReLU -> Linear(latent -> 1).  Same constructor, sub-module names and ``state_dict`` as the reference.  The optional
attention pair (``use_attn=True``) cannot run in the reference either -- ``SpatialAttention(d_inp=out_dim)`` with
``n_head * d_head != out_dim`` fails in its LayerNorm (test_discriminator 2/4) -- and raises here at
construction.  ``VideoDiscriminator`` (discriminator.py:116-222) is not on the path SURVEY.md section 8f-2 names and raises.
"""
from __future__ import annotations

from itertools import pairwise
from math import prod
from typing import Tuple

import torch
import torch.nn as nn
from torch import Tensor

from .. import functional as GF
from .image import Conv2d, ImageResidualBlock, LeakyReLU, as_frames


class _Flatten(nn.Module):
    """Rearrange('b ... -> b (...)') of the reference, on the LOGICAL (c, h, w) order (the CL memory order is (h, w, c))."""

    def forward(self, x: Tensor) -> Tensor:
        return x


class _Squeeze(nn.Module):
    def forward(self, x: Tensor) -> Tensor:
        return x


class FrameDiscriminator(nn.Module):
    def __init__(self, inp_size: int | Tuple[int, int], model_dim: int = 64, dim_mults: Tuple[int, ...] = (1, 2, 4),
                 down_step: Tuple[int | None, ...] = (None, 2, 2), inp_channels: int = 3, kernel_size: int | Tuple[int, int] = 3,
                 num_groups: int = 1, num_heads: int = 4, dim_head: int = 32, use_attn: bool = False, use_blur: bool = True,
                 act_fn: str = 'leaky') -> None:
        super().__init__()
        if isinstance(inp_size, int):
            inp_size = (inp_size, inp_size)
        dims = [model_dim * mult for mult in dim_mults]
        assert len(dims) == len(down_step), 'Dimension and downsample steps must match.'
        if use_attn:
            raise NotImplementedError('FrameDiscriminator(use_attn=True): the attention pair cannot run in the reference either '
                                      '(SpatialAttention(d_inp=out_dim) fails in its LayerNorm, SURVEY.md section 4); not implemented')
        self.proj_in = Conv2d(inp_channels, model_dim, kernel_size=3, padding=1)
        self.core = nn.ModuleList([])
        out_dim = model_dim
        for (inp_dim, out_dim), down in zip(pairwise(dims), down_step):          # zip stops at the shorter: the last down_step is unused (as in the reference)
            res_block = ImageResidualBlock(inp_dim, out_dim, downsample=down, num_groups=num_groups, kernel_size=kernel_size)
            self.core.append(nn.ModuleList([res_block, nn.ModuleList([nn.Identity(), nn.Identity()])]))
            inp_size = tuple(map(lambda x: x // (down or 1), inp_size))
        self.latent_size = (out_dim, *inp_size)
        latent_dim = out_dim * prod(inp_size)
        self.to_logits = nn.Sequential(Conv2d(out_dim, out_dim, kernel_size=3, padding=1), LeakyReLU(), _Flatten(), nn.Linear(latent_dim, 1), _Squeeze())

    def forward(self, image: Tensor) -> Tensor:
        x, _ = as_frames(image)
        out = self.proj_in(x)
        for res, (attn, ff) in self.core:
            out = res(out)
            out = out + out                  # `attn(out) + out` with attn = Identity (discriminator.py:108) ...
            out = out + out                  # ... and `ff(out) + out` with ff = Identity (:109): each doubles the features
        out = self.to_logits[1](self.to_logits[0](out))
        n, c, _, h, w = out.shape
        if (c, h, w) != self.latent_size:
            raise RuntimeError(f'FrameDiscriminator: features {(c, h, w)} do not match the latent size {self.latent_size} fixed by inp_size')
        lin = self.to_logits[3]
        # 'b c h w -> b (c h w)' @ W^T: the CL buffer is ordered (h, w, c), so the weight's columns are permuted instead of the features
        w_perm = lin.weight.view(1, c, h, w).permute(0, 2, 3, 1).reshape(1, -1)
        feats = out.permute(0, 2, 3, 4, 1).reshape(n, h * w * c)
        return (feats.float() @ w_perm.t().float() + lin.bias.float()).reshape(n)


class VideoDiscriminator(nn.Module):
    def __init__(self, *args, **kwargs) -> None:
        super().__init__()
        raise NotImplementedError("VideoDiscriminator (reference discriminator.py:116-222) is outside the implemented path "
                                  "(SURVEY.md section 8f-2 names the frame critic); use gan_discriminate='frames'")
