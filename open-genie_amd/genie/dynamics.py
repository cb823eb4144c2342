"""DynamicsModel (MaskGIT) on the HIP hot path (drop-in for reference genie/dynamics.py:14-195).

Token + action embedding lookups are index gathers (torch); the N space-time blocks and the vocabulary head run
on the HIP kernels -- the head ``Linear(D -> V)`` is the 1x1x1 case of the gather-GEMM, so the MaskGIT token
logits come off the bf16 MFMA path.  Reference quirks kept: ``compute_loss`` reads its targets AFTER the masked
fill (so they all equal ``fill``) and needs batch >= 2 (``mask.squeeze()``); ``generate`` never feeds painted
codes back into the context and ignores ``topk``.
"""
from __future__ import annotations

from math import inf, pi, prod
from typing import Literal, Optional

import torch
import torch.nn as nn
from torch import Tensor

from . import functional as GF
from .cl import to_cl
from .conv import ConvSpec
from .module import parse_blueprint
from .utils import Blueprint, default


class _ActRearrange(nn.Module):
    """'b t d -> b t 1 1 d' (reference dynamics.py:36)."""

    def forward(self, x: Tensor) -> Tensor:
        return x[:, :, None, None, :]


class DynamicsModel(nn.Module):
    def __init__(self, desc: Blueprint, tok_vocab: int, act_vocab: int, embed_dim: int) -> None:
        super().__init__()
        self.dec_layers, self.ext_kw = parse_blueprint(tuple(d if isinstance(d, str) else (d[0], dict(d[1])) for d in desc))
        self.head = nn.Linear(embed_dim, tok_vocab)
        self.tok_emb = nn.Embedding(tok_vocab, embed_dim)
        self.act_emb = nn.Sequential(nn.Embedding(act_vocab, embed_dim), _ActRearrange())
        self.tok_vocab, self.act_vocab, self.embed_dim = tok_vocab, act_vocab, embed_dim
        self._head_op = GF.ConvOp(ConvSpec(embed_dim, tok_vocab, (1, 1, 1)))

    def forward_order(self):
        """Sub-modules in execution order (trainer.execution_order lays the parameter arena out this way)."""
        return [self.tok_emb, self.act_emb, self.dec_layers, self.head]

    def _trunk(self, tokens: Tensor, act_id: Tensor) -> Tensor:
        # (B, T, H, W, D); both lookups and their sparse backward are genie_embedding_fwd / _bwd (act_emb[1] is the 'b t d -> b t 1 1 d' rearrange)
        x = GF.embedding(tokens, self.tok_emb.weight) + self.act_emb[1](GF.embedding(act_id, self.act_emb[0].weight))
        for dec in self.dec_layers:
            x = dec(x)
        return x

    def _head(self, x: Tensor) -> Tensor:
        """x: (B, T, H, W, D) -> logits (B, T, H, W, V) bf16 via the 1x1x1 gather-GEMM."""
        w = self.head.weight
        y = GF.conv3d(to_cl(x.permute(0, 4, 1, 2, 3)), w[:, :, None, None, None], self.head.bias, self._head_op)
        return y.permute(0, 2, 3, 4, 1)

    def forward(self, tokens: Tensor, act_id: Tensor):
        logits = self._head(self._trunk(tokens, act_id))
        return logits, logits[:, -1]

    def compute_loss(self, tokens: Tensor, act_id: Tensor, mask: Tensor | None = None, fill: float = 0., fixed_rows: bool = False) -> Tensor:
        """`fixed_rows`: the same loss with SHAPES that do not depend on the mask -- the masked rows are moved to the front by a stable
        device-side sort, the vocabulary head runs over all B*T*h*w rows and the cross-entropy switches the others off.  Costs the head
        GEMMs of the unmasked rows (a quarter of them on average) and buys a step without host round trips or data-dependent launches,
        i.e. one that a hipGraph can replay (genie/graph.py; the mask is then an INPUT of the step, drawn by the caller)."""
        b, t, h, w = tokens.shape
        mask = default(mask, torch.distributions.Bernoulli(torch.empty(1).uniform_(0.5, 1).item()).sample((b, t, h, w)).bool())
        host_mask = mask if mask.device.type == 'cpu' and not fixed_rows else None
        mask = mask.to(tokens.device)
        tokens = torch.masked_fill(tokens, mask, fill)
        # (fixed_rows: the shape-stable form never needs the reference's squeeze(), so a batch of ONE clip stays capturable)
        m = mask.reshape(tokens.shape) if fixed_rows else mask.squeeze()
        if m.shape != tokens.shape:
            # batch 1: the reference's mask.squeeze() drops the batch axis and its boolean indexing misbehaves; same code, same fate
            logits, _ = self(tokens, act_id.detach())
            logits = logits[m]
            target = tokens[m]
            return torch.nn.functional.cross_entropy(logits.reshape(-1, logits.shape[-1]).float(), target.reshape(-1))
        # logits[m] / tokens[m] / cross_entropy(mean) of the reference (dynamics.py:89-97).  Only the masked rows enter the loss, the head
        # is row-wise, and the logits are not returned: gather those rows of the trunk output FIRST and run the vocabulary head, the
        # cross-entropy and their backward on them alone (a Bernoulli(0.5 .. 1) mask drops a quarter of the 2^18-wide rows on average);
        # the fused genie_masked_ce_fwd / _bwd then never materialise the gathered rows' fp32 copy or their softmax.
        # A mask that lives on the host (the default one does) gives the row list without a device round trip.
        if fixed_rows:
            flat = m.reshape(-1)
            total = flat.numel()
            order = torch.argsort((~flat).to(torch.uint8), stable=True)        # masked rows first, in their original order
            gt, gh, rp = self._compact_grid(total)
            rows = order if rp == total else torch.cat([order, order.new_zeros(rp - total)])
            valid = torch.arange(rp, device=rows.device) < flat.sum()          # (no masked row at all: 0 / 0 = NaN, as F.cross_entropy)
            x = self._trunk(tokens, act_id.detach())
            d = x.shape[-1]
            xr = x.reshape(-1, d).index_select(0, rows)
            if GF.linear_ce_supported(xr, self.head.weight):
                return self._fused_loss(xr, tokens.reshape(-1).index_select(0, rows), valid)
            return GF.masked_cross_entropy(self._head(xr.view(1, gt, gh, 256, d)), tokens.reshape(-1).index_select(0, rows).view(1, gt, gh, 256), valid)
        flat = (host_mask.squeeze() if host_mask is not None else m).reshape(-1)
        rows = flat.nonzero().squeeze(1).to(tokens.device)
        x = self._trunk(tokens, act_id.detach())
        d = x.shape[-1]
        if rows.numel() == 0:
            return x.sum() * float('nan')                                      # F.cross_entropy over zero rows (mean) is NaN
        xr = x.reshape(-1, d).index_select(0, rows)
        if GF.linear_ce_supported(xr, self.head.weight):
            # Linear(D -> V) + cross-entropy as ONE operator: the (rows, V) logits and their gradient never reach HBM (csrc/linear_ce.hip)
            return self._fused_loss(xr, tokens.reshape(-1).index_select(0, rows), None)
        # the compact rows are laid out as a (t, h, 256) grid for the gather-GEMM (every axis < 1024); the few pad rows re-read row 0
        # and are switched off in the cross-entropy (zero loss, zero gradient)
        r = rows.numel()
        gt, gh, rp = self._compact_grid(r)
        if rp != r:
            rows = torch.cat([rows, rows.new_zeros(rp - r)])
        valid = None if rp == r else (torch.arange(rp, device=rows.device) < r)
        xc = x.reshape(-1, d).index_select(0, rows).view(1, gt, gh, 256, d)
        logits = self._head(xc)                                                # (1, gt, gh, 256, V)
        return GF.masked_cross_entropy(logits, tokens.reshape(-1).index_select(0, rows).view(1, gt, gh, 256), valid)

    def _fused_loss(self, xr: Tensor, target: Tensor, valid: Optional[Tensor]) -> Tensor:
        w = self.head.weight
        return GF.linear_cross_entropy(xr, w, self.head.bias, self._head_op.pack_fwd(w[:, :, None, None, None]), target, valid)

    @staticmethod
    def _compact_grid(r: int):
        """(t, h, padded rows) of the (t, h, 256) grid that holds r >= 1 gathered rows: every axis stays below the 1024 the gather-GEMM's
        row decomposition allows, at most 255 pad rows up to 131072 rows (512 * 256 - 1 beyond)."""
        k = (r + 255) // 256
        gh = k if k <= 512 else 512
        gt = (k + gh - 1) // gh
        return gt, gh, gt * gh * 256

    def _last_frame_logits(self, tokens: Tensor, act_id: Tensor) -> Tensor:
        """logits[:, -1] of ``forward`` -- (B, h, w, V) bf16 -- with the vocabulary head applied to the last frame only (the head is
        row-wise, so these are the same numbers; at V = 2^18 the other T frames are 94 % of the head GEMM)."""
        x = self._trunk(tokens, act_id)
        return self._head(x[:, -1:])[:, 0]

    @torch.no_grad()
    def generate(self, tokens: Tensor, act_id: Tensor, steps: int = 10, which: Literal['linear', 'cosine', 'arccos'] = 'linear',
                 temp: float = 1., topk: int = 50, masked_tok: int = 0, uniforms: Optional[Tensor] = None,
                 feedback: bool = False, trace: Optional[list] = None) -> Tensor:
        """MaskGIT sampling (reference dynamics.py:101-165) with the whole per-step chain on the device:
        softmax -> categorical draw -> confidence -> top-k -> scatter are two HIP kernels (genie_maskgit_sample / _paint) and the
        loop never synchronises with the host (the reference syncs at ``mask.sum() == 0`` every step).

        ``uniforms`` (steps, B*h*w) injects the noise of the categorical draws (inverse CDF), which makes token ids reproducible
        and comparable with the CPU oracle; without it the uniforms come from the device RNG (``torch.rand``), which is the same
        distribution as the reference's ``torch.multinomial``.  Reference quirks kept by default: painted codes are never fed back
        into the context (dynamics.py:128,136 -- the logits are therefore the same at every step and are computed ONCE here) and
        ``topk`` is unused.  ``feedback=True`` is the opt-in repair: the painted codes replace the masked frame before each step."""
        b, t, h, w = tokens.shape
        n = h * w
        dev = tokens.device
        schedule = self.get_schedule(steps, shape=(h, w), which=which)
        mask = torch.ones(b, n, dtype=torch.uint8, device=dev)
        code = torch.full((b, n), masked_tok, dtype=torch.int64, device=dev)
        act = torch.cat([act_id, torch.zeros(b, 1, dtype=act_id.dtype, device=dev)], dim=1)
        tok_id = torch.cat([tokens, code.reshape(b, 1, h, w).to(tokens.dtype)], dim=1)
        logits = None
        remaining = n              # masked positions per sample, tracked on the HOST: every step paints exactly `num_tokens` of them
        for step, num_tokens in enumerate(schedule.tolist()):     # (the schedule is a CPU tensor: no device sync here either)
            if remaining <= 0:     # the reference's `if mask.sum() == 0: break` (dynamics.py:133) without its device->host sync
                break
            remaining -= num_tokens
            if logits is None or feedback:
                logits = self._last_frame_logits(tok_id, act)                       # (B, h, w, V) bf16
            u = uniforms[step] if uniforms is not None else torch.rand(b * n, device=dev)
            pred, conf = GF.maskgit_sample(logits, u, temp)
            if trace is not None:
                trace.append({'logits': logits, 'pred': pred.clone(), 'conf': conf.clone(), 'mask_before': mask.clone(), 'k': num_tokens})
            GF.maskgit_paint(conf, pred, num_tokens, code, mask)
            if feedback:
                tok_id[:, -1] = code.reshape(b, h, w).to(tok_id.dtype)
        assert mask.sum() == 0, f'Not all tokens were predicted. {mask.sum()} tokens left.'
        return torch.cat([tokens, code.reshape(b, 1, h, w).to(tokens.dtype)], dim=1)

    def get_schedule(self, steps: int, shape, which: Literal['linear', 'cosine', 'arccos'] = 'linear') -> Tensor:
        n = prod(shape)
        t = torch.linspace(1, 0, steps)
        match which:
            case 'linear':
                s = 1 - t
            case 'cosine':
                s = torch.cos(t * pi * .5)
            case 'arccos':
                s = torch.acos(t) / (pi * .5)
            case _:
                raise ValueError(f'Unknown schedule type: {which}')
        schedule = ((s / s.sum()) * n).round().int().clamp(min=1)
        schedule[-1] += n - schedule.sum()
        return schedule
