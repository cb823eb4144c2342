"""LatentAction VQ-VAE (drop-in for reference genie/action.py:31-176) on the HIP hot path.

The reference class cannot be constructed or run with any blueprint it ships: its blueprints pass
an ``n_embd`` keyword the ST block rejects, name a ``'spacetime_upsample'`` module the registry lacks, and build the LFQ
without ``input_dim`` so that it projects from 2^d features.  This class follows action.py:111-176 line by line and applies
the minimal repair R-lam: blueprints must use ``'depth2spacetime_upsample'`` and ST blocks with
``transpose=True`` and ``n_head * d_head == n_embd`` (see ``genie.LATENT_ACT_ENC/DEC``), and the LFQ is built with
``input_dim = d_codebook * n_codebook``.
"""
from __future__ import annotations

from math import prod
from typing import Tuple

import torch
import torch.nn as nn
from torch import Tensor

from . import functional as GF
from .cl import is_cl, to_cl
from .module import parse_blueprint
from .module.quantization import LookupFreeQuantization
from .module.video import CausalConv3d, Downsample, Upsample
from .utils import Blueprint


class _ToActRearrange(nn.Module):
    """'b c t ... -> b t (c ...)' (reference action.py:84); folded into the column order of the projection."""

    def forward(self, x: Tensor) -> Tensor:
        return x


class LatentAction(nn.Module):
    def __init__(self, enc_desc: Blueprint, dec_desc: Blueprint, d_codebook: int, inp_channels: int = 3,
                 inp_shape: int | Tuple[int, int] = (64, 64), ker_size: int | Tuple[int, int] = 3, n_embd: int = 256,
                 n_codebook: int = 1, lfq_bias: bool = True, lfq_frac_sample: float = 1., lfq_commit_weight: float = 0.25,
                 lfq_entropy_weight: float = 0.1, lfq_diversity_weight: float = 1., quant_loss_weight: float = 1.) -> None:
        super().__init__()
        if isinstance(inp_shape, int):
            inp_shape = (inp_shape, inp_shape)
        copy = lambda desc: tuple(d if isinstance(d, str) else (d[0], dict(d[1])) for d in desc)
        self.proj_in = CausalConv3d(inp_channels, out_channels=n_embd, kernel_size=ker_size)
        self.proj_out = CausalConv3d(n_embd, out_channels=inp_channels, kernel_size=ker_size)
        self.enc_layers, self.enc_ext = parse_blueprint(copy(enc_desc))
        self.dec_layers, self.dec_ext = parse_blueprint(copy(dec_desc))
        enc_fact = prod(enc.factor for enc in self.enc_layers if isinstance(enc, (Downsample, Upsample)))
        dec_fact = prod(dec.factor for dec in self.dec_layers if isinstance(dec, (Downsample, Upsample)))
        assert enc_fact * dec_fact == 1, 'The product of the space-time up/down factors must be 1.'
        self.to_act = nn.Sequential(_ToActRearrange(), nn.Linear(int(n_embd * enc_fact * prod(inp_shape)), d_codebook, bias=False))
        self.quant = LookupFreeQuantization(codebook_dim=d_codebook, num_codebook=n_codebook, input_dim=d_codebook * n_codebook,   # R-lam (3)
                                            use_bias=lfq_bias, frac_sample=lfq_frac_sample, commit_weight=lfq_commit_weight,
                                            entropy_weight=lfq_entropy_weight, diversity_weight=lfq_diversity_weight)
        self.d_codebook, self.n_codebook, self.quant_loss_weight = d_codebook, n_codebook, quant_loss_weight

    def sample(self, idxs: Tensor) -> Tensor:
        return self.quant.codebook[idxs]

    def forward_order(self):
        """Sub-modules in execution order (trainer.execution_order lays the parameter arena out this way): the action head and
        its quantiser run after the encoder; their gradients are complete once the decoder's backward has delivered the condition
        gradients, i.e. before the encoder's backward starts."""
        return [self.proj_in, self.enc_layers, self.to_act, self.quant, self.dec_layers, self.proj_out]

    def _project_to_action(self, video: Tensor) -> Tensor:
        """Linear over 'b c t h w -> b t (c h w)' features.  The CL memory order of a frame is (h, w, c), so the weight's
        columns are permuted instead of the activation (a (d, C*H*W) tensor vs. B*T*C*H*W elements)."""
        x = to_cl(video)
        b, c, t, h, w = x.shape
        wt = self.to_act[1].weight
        if wt.shape[1] != c * h * w:
            raise RuntimeError(f'to_act expects {wt.shape[1]} features per frame, the encoder produced {c}x{h}x{w}')
        w_perm = wt.reshape(-1, c, h, w).permute(0, 2, 3, 1).reshape(wt.shape[0], -1)
        frames = x.permute(0, 2, 3, 4, 1).reshape(b * t, h * w * c)            # zero-copy view of the CL buffer
        # (B T, h w c) x (d, h w c)^T in fp32 arithmetic (csrc/linear_small.hip: the 2^18 .. 2^20-long reduction in register-resident weight
        # slices, partials summed in a fixed order); rounds 1-5: one bf16 library GEMM
        return GF.linear(frames, w_perm.contiguous(), None, out_dtype=torch.float32).reshape(b, t, -1)

    def encode(self, video: Tensor, mask: Tensor | None = None, transpose: bool = False):
        video = self.proj_in(video)
        for enc in self.enc_layers:
            video = enc(video, mask=mask)
        act = self._project_to_action(video)
        (act, idxs), q_loss = self.quant(act, transpose=transpose)
        return (act, idxs, video), q_loss

    def decode(self, video: Tensor, q_act: Tensor) -> Tensor:
        for dec, has_ext in zip(self.dec_layers, self.dec_ext):
            video = dec(video, cond=(None, q_act if has_ext else None))
        return self.proj_out(video)

    def forward(self, video: Tensor, mask: Tensor | None = None):
        (act, idxs, enc_video), q_loss = self.encode(video, mask=mask)
        recon = self.decode(enc_video, act)
        rec_loss = GF.mse_loss(recon, video)
        loss = rec_loss + (q_loss * self.quant_loss_weight if q_loss is not None else 0)
        return idxs, loss, (rec_loss, q_loss)
