"""ForwardBlock (drop-in for reference genie/module/misc.py:71-104): GroupNorm then a stack of blocks."""
from __future__ import annotations

from itertools import pairwise
from typing import Tuple

import torch.nn as nn
from torch import Tensor

from ..conv import same_spec
from ..utils import default
from .norm import GELU, GroupNorm
from .video import Conv3d, _triple


class ForwardBlock(nn.Module):
    def __init__(self, in_dim: int, out_dim: int | None = None, hid_dim: int | Tuple[int, ...] | None = 256,
                 block=nn.Linear, act_fn=nn.GELU, num_groups: int = 1, last_act: bool = False, **kwargs) -> None:
        super().__init__()
        if act_fn is nn.GELU:
            act_fn = GELU                                     # same module, HIP kernel on video tensors
        out_dim = default(out_dim, in_dim)
        if isinstance(hid_dim, int):
            hid_dim = (hid_dim,)
        hid_dim = default(hid_dim, ())
        dims = (in_dim,) + tuple(hid_dim) + (out_dim,)

        def make(i, o):
            if block is nn.Conv3d or block is Conv3d:
                k = _triple(kwargs.get('kernel_size', 1))
                pad = _triple(kwargs.get('padding', 0))
                if pad != tuple((kk - 1) // 2 for kk in k):
                    raise NotImplementedError('ForwardBlock: only "same" padded Conv3d blocks are implemented on the HIP path')
                return Conv3d(i, o, k, same_spec(i, o, k), bias=kwargs.get('bias', True))
            return block(i, o, **kwargs)

        self.net = nn.Sequential(
            GroupNorm(num_groups, in_dim),
            *[nn.Sequential(make(i, o), act_fn() if l < len(dims) - 2 or last_act else nn.Identity())
              for l, (i, o) in enumerate(pairwise(dims))],
        )

    def forward(self, inp: Tensor) -> Tensor:
        return self.net(inp)
