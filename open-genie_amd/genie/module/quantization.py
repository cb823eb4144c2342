"""Lookup-free quantisation (drop-in for reference genie/module/quantization.py:32-133)."""
from __future__ import annotations

from typing import Tuple

import torch
import torch.nn as nn
from torch import Tensor

from .. import _hip
from .. import functional as GF
from ..cl import is_cl
from ..utils import default


def entropy(p: Tensor, eps: float = 1e-6) -> Tensor:
    """reference quantization.py:17-28 (kept for API completeness; the training loss does not call it)."""
    return -(p * torch.log(p.clamp(min=eps))).sum(dim=-1)


class LookupFreeQuantization(nn.Module):
    """quant = sign(x), idx = MSB-first bit pack, straight-through gradient; in training mode also the
    entropy + commitment loss over all 2^d codes, computed (with its gradient) by ``genie_lfq_loss`` without
    materialising the N x 2^d probability matrix.  ``proj_inp`` / ``proj_out`` run on the skinny-linear kernels (functional.linear)."""

    def __init__(self, codebook_dim: int, num_codebook: int = 1, input_dim: int | None = None, use_bias: bool = True,
                 frac_sample: float = 1., commit_weight: float = 0.25, entropy_weight: float = 0.1,
                 diversity_weight: float = 1.) -> None:
        super().__init__()
        codebook_size = (2 ** codebook_dim) * num_codebook
        input_dim = default(input_dim, codebook_size)
        project = input_dim != codebook_dim * num_codebook
        self.proj_inp = nn.Linear(input_dim, codebook_dim * num_codebook, bias=use_bias) if project else nn.Identity()
        self.proj_out = nn.Linear(codebook_dim * num_codebook, input_dim, bias=use_bias) if project else nn.Identity()
        self.frac_sample = frac_sample               # stored, unused -- as in the reference (quantization.py:60)
        self.codebook_dim, self.num_codebooks, self.codebook_size = codebook_dim, num_codebook, codebook_size
        self.commit_weight, self.entropy_weight, self.diversity_weight = commit_weight, entropy_weight, diversity_weight
        self.register_buffer('bit_mask', 2 ** torch.arange(codebook_dim - 1, -1, -1))
        self._codebook = None

    @property
    def codebook(self) -> Tensor:
        """(codebook_size, d) table of {-1,+1} codes; non-persistent in the reference (quantization.py:74-75), built
        lazily here because nothing on the hot path reads it (LatentAction.sample does, action.py:107-109)."""
        cb = self._codebook
        if cb is None or cb.device != self.bit_mask.device:
            codes = torch.arange(self.codebook_size, device=self.bit_mask.device)[:, None] & self.bit_mask
            cb = self._codebook = 2 * (codes != 0).float() - 1
        return cb

    def forward(self, inp: Tensor, beta: float = 100., transpose: bool = False) -> Tuple[Tuple[Tensor, Tensor], Tensor | None]:
        _hip.require_gpu(inp, 'LookupFreeQuantization')
        d, c = self.codebook_dim, self.num_codebooks
        project = not isinstance(self.proj_inp, nn.Identity)
        # rows view: 'b d ... -> b ... d' (transpose) then pack 'b * d'
        x = inp.movedim(1, -1) if transpose else inp
        lead = x.shape[:-1]
        if project:
            z = GF.linear(x, self.proj_inp.weight, self.proj_inp.bias, out_dtype=torch.float32)      # fp32 arithmetic, fp32 z (csrc/linear_small.hip)
            rows = z.reshape(-1, c * d)
        elif transpose and inp.dim() == 5 and is_cl(inp):
            rows = x.reshape(-1, x.shape[-1])              # zero-copy view of the CL latent (pitch = channel pitch)
        else:
            rows = x.reshape(-1, c * d)
        if rows.dtype not in (torch.float32, torch.bfloat16):
            rows = rows.float()
        if rows.stride(-1) != 1 or (rows.shape[0] > 1 and rows.stride(0) < c * d):
            rows = rows.contiguous()
        quant, idxs, loss4 = GF.lfq_rows(rows, c, d, self.training, float(beta), self.commit_weight, self.entropy_weight,
                                          self.diversity_weight)
        out = GF.linear(quant, self.proj_out.weight, self.proj_out.bias, out_dtype=torch.float32) if project else quant
        out = out.reshape(*lead, out.shape[-1])
        if transpose:
            out = out.movedim(-1, 1)
        idxs = idxs.reshape(*lead, c).squeeze()            # drops EVERY size-1 dim, like the reference (:110)
        if not self.training:
            return (out, idxs), None
        return (out, idxs), loss4[0]
