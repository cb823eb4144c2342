"""GroupNorm family on the HIP streaming kernels (drop-in for reference genie/module/norm.py and for the
``nn.GroupNorm`` / ``nn.SiLU`` entries of the reference registry, genie/module/__init__.py:55-67)."""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch import Tensor

from .. import functional as GF


class GroupNorm(nn.GroupNorm):
    """``nn.GroupNorm`` (same ctor, same ``weight``/``bias`` keys) computed by ``genie_groupnorm_fwd/bwd``."""

    def forward(self, inp: Tensor) -> Tensor:
        return GF.group_norm(inp, self.num_groups, self.weight, self.bias, self.eps)


class SiLU(nn.SiLU):
    def forward(self, inp: Tensor) -> Tensor:
        return GF.silu(inp)


class GELU(nn.GELU):
    """nn.GELU() on the HIP path (5-D video tensors, exact erf form; the reference's ForwardBlock activation, misc.py:78).  Other ranks
    and the tanh approximation stay with torch."""

    def forward(self, inp: Tensor) -> Tensor:
        if inp.dim() == 5 and inp.is_cuda and self.approximate == 'none':
            return GF.gelu(inp)
        return super().forward(inp)


class AdaptiveGroupNorm(nn.Module):
    """reference norm.py:8-69: group_norm(x) * Linear(mean_{t,h,w} cond) + Linear(mean cond).
    The scale/shift are folded into the normalisation pass; the two (B, dim_cond) x (dim_cond, C) products are
    plain library GEMMs."""

    def __init__(self, dim_cond: int, num_groups: int, num_channels: int, cond_bias: bool = True, affine: bool = True,
                 eps: float = 1e-5, device=None, dtype=None) -> None:
        super().__init__()
        if num_channels % num_groups != 0:
            raise ValueError('num_channels must be divisible by num_groups')
        self.num_groups, self.num_channels, self.eps, self.affine = num_groups, num_channels, eps, affine
        kw = {'device': device, 'dtype': dtype}
        if affine:
            self.weight = nn.Parameter(torch.empty(num_channels, **kw))
            self.bias = nn.Parameter(torch.empty(num_channels, **kw))
        else:
            self.register_parameter('weight', None)
            self.register_parameter('bias', None)
        self.std = nn.Linear(dim_cond, num_channels)
        self.avg = nn.Linear(dim_cond, num_channels) if cond_bias else None
        self.reset_parameters()

    def reset_parameters(self) -> None:
        if self.affine:
            nn.init.ones_(self.weight)
            nn.init.zeros_(self.bias)
        nn.init.ones_(self.std.bias)
        nn.init.zeros_(self.std.weight)
        if self.avg is not None:
            nn.init.zeros_(self.avg.bias)
            nn.init.zeros_(self.avg.weight)

    def forward(self, inp: Tensor, cond: Tensor) -> Tensor:
        c = cond.float().reshape(cond.shape[0], cond.shape[1], -1).mean(-1)        # norm.py:62
        std = GF.linear(c, self.std.weight, self.std.bias)                         # (B, dim_cond) -> (B, C): csrc/linear_small.hip
        avg = GF.linear(c, self.avg.weight, self.avg.bias) if self.avg is not None else None
        return GF.group_norm(inp, self.num_groups, self.weight, self.bias, self.eps, ada_scale=std, ada_shift=avg)
