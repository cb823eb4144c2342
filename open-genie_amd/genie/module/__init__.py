"""String -> class registry and blueprint parser: THE drop-in seam (reference genie/module/__init__.py:23-93).

Same names, same return contract (``parse_blueprint`` -> ``(nn.ModuleList, List[bool])``), same
``ValueError`` for unknown names, and -- like the reference -- ``has_ext`` / ``n_rep`` are popped from the
caller's kwargs dicts."""
from typing import List, Tuple

import torch.nn as nn

from ..utils import Blueprint, default, exists
from .norm import GELU, AdaptiveGroupNorm, GroupNorm, SiLU
from .video import (CausalConv3d, CausalConvTranspose3d, DepthToSpaceTimeUpsample, DepthToSpaceUpsample,
                    DepthToTimeUpsample, SpaceTimeDownsample, VideoResidualBlock)


def get_module(name: str):
    match name:
        case 'space_attn':
            from .attention import SpatialAttention
            return SpatialAttention
        case 'time_attn':
            from .attention import TemporalAttention
            return TemporalAttention
        case 'space-time_attn':
            from .attention import SpaceTimeAttention
            return SpaceTimeAttention
        case 'blur_pool':
            from .image import BlurPooling2d                  # the 2-D form, as in the reference's registry (__init__.py:33-34)
            return BlurPooling2d
        case 'space_downsample':
            from .image import SpaceDownsample
            return SpaceDownsample
        case 'image-residual':
            from .image import ImageResidualBlock
            return ImageResidualBlock
        case 'video-residual':
            return VideoResidualBlock
        case 'causal-conv3d':
            return CausalConv3d
        case 'causal-conv3d-transpose':
            return CausalConvTranspose3d
        case 'depth2space_upsample':
            return DepthToSpaceUpsample
        case 'depth2time_upsample':
            return DepthToTimeUpsample
        case 'depth2spacetime_upsample':
            return DepthToSpaceTimeUpsample
        case 'spacetime_downsample':
            return SpaceTimeDownsample
        case 'group_norm':
            return GroupNorm
        case 'adaptive_group_norm':
            return AdaptiveGroupNorm
        case 'gelu':
            return GELU                       # nn.GELU subclass: same constructor, same (empty) state_dict
        case 'relu':
            return nn.ReLU
        case 'leaky_relu':
            return nn.LeakyReLU
        case 'silu':
            return SiLU
        case _:
            raise ValueError(f'Unknown module name: {name}')


def parse_blueprint(blueprint: Blueprint) -> Tuple[nn.ModuleList, List[bool]]:
    layers, ext_kw = [], []
    for desc in blueprint:
        if isinstance(desc, str):
            desc = (desc, {})
        name, kwargs = default(desc, (None, {}))
        ext_kw.extend([kwargs.pop('has_ext', False)] * kwargs.get('n_rep', 1))
        layers.extend([get_module(name)(**kwargs) for _ in range(kwargs.pop('n_rep', 1)) if exists(name) and exists(kwargs)])
    return nn.ModuleList(layers), ext_kw
