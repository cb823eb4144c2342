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


def sample_from_uniform(prob: Tensor, u: Tensor) -> Tensor:
    """Inverse-CDF categorical draw from injected uniforms: #(cumsum(prob) <= u * total), clamped.  Used instead of
    torch.multinomial when ``generate(..., uniforms=...)`` is given so token ids are reproducible across devices."""
    cdf = prob.double().cumsum(-1)
    thr = (u.double() * cdf[:, -1])[:, None]
    return (cdf <= thr).sum(-1).clamp(max=prob.shape[-1] - 1)


class DynamicsModel(nn.Module):
    def __init__(self, desc: Blueprint, tok_vocab: int, act_vocab: int, embed_dim: int) -> None:
        super().__init__()
        self.dec_layers, self.ext_kw = parse_blueprint(tuple(d if isinstance(d, str) else (d[0], dict(d[1])) for d in desc))
        self.head = nn.Linear(embed_dim, tok_vocab)
        self.tok_emb = nn.Embedding(tok_vocab, embed_dim)
        self.act_emb = nn.Sequential(nn.Embedding(act_vocab, embed_dim), _ActRearrange())
        self.tok_vocab, self.act_vocab, self.embed_dim = tok_vocab, act_vocab, embed_dim
        self._head_op = GF.ConvOp(ConvSpec(embed_dim, tok_vocab, (1, 1, 1)))

    def _trunk(self, tokens: Tensor, act_id: Tensor) -> Tensor:
        x = self.tok_emb(tokens) + self.act_emb(act_id)                 # (B, T, H, W, D)
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

    def compute_loss(self, tokens: Tensor, act_id: Tensor, mask: Tensor | None = None, fill: float = 0.) -> Tensor:
        b, t, h, w = tokens.shape
        mask = default(mask, torch.distributions.Bernoulli(torch.empty(1).uniform_(0.5, 1).item()).sample((b, t, h, w)).bool())
        mask = mask.to(tokens.device)
        tokens = torch.masked_fill(tokens, mask, fill)
        logits, _ = self(tokens, act_id.detach())
        m = mask.squeeze()
        if m.shape != tokens.shape:
            # batch 1: the reference's mask.squeeze() drops the batch axis and its boolean indexing misbehaves; same code, same fate
            logits = logits[m]
            target = tokens[m]
            return torch.nn.functional.cross_entropy(logits.reshape(-1, logits.shape[-1]).float(), target.reshape(-1))
        # logits[m] / tokens[m] / cross_entropy(mean) of the reference, without materialising the gathered rows, their fp32 copy,
        # the softmax and its backward: one fused pass forward, one backward (genie_masked_ce_fwd / _bwd)
        return GF.masked_cross_entropy(logits, tokens, m)

    @torch.no_grad()
    def generate(self, tokens: Tensor, act_id: Tensor, steps: int = 10, which: Literal['linear', 'cosine', 'arccos'] = 'linear',
                 temp: float = 1., topk: int = 50, masked_tok: int = 0, uniforms: Optional[Tensor] = None) -> Tensor:
        b, t, h, w = tokens.shape
        schedule = self.get_schedule(steps, shape=(h, w), which=which)
        mask = torch.ones(b, h * w, dtype=torch.bool, device=tokens.device)
        code = torch.full((b, h * w), masked_tok, device=tokens.device, dtype=tokens.dtype)
        mock = torch.zeros(b, 1, dtype=act_id.dtype, device=tokens.device)
        tok_id = torch.cat([tokens, code.reshape(b, 1, h, w)], dim=1)
        act = torch.cat([act_id, mock], dim=1)
        pred_tok = tok_id
        remaining = h * w          # masked positions per sample, tracked on the HOST: every step paints exactly `num_tokens` of them
        for step, num_tokens in enumerate(schedule.tolist()):     # (the schedule is a CPU tensor: no device sync here either)
            if remaining <= 0:     # the reference's `if mask.sum() == 0: break` (dynamics.py:133) without its device->host sync
                break
            remaining -= num_tokens
            _, logits = self(tok_id, act)
            prob = torch.softmax(logits.float() / temp, dim=-1).reshape(b * h * w, -1)
            pred = sample_from_uniform(prob, uniforms[step].to(prob.device)) if uniforms is not None else torch.multinomial(prob, num_samples=1).squeeze(-1)
            conf = prob.gather(-1, pred[:, None]).reshape(b, h * w).masked_fill(~mask, -inf)      # (boolean index_put would sync)
            idxs = torch.topk(conf, k=num_tokens, dim=-1).indices
            vals = pred.reshape(b, -1).gather(-1, idxs).to(code.dtype)
            code.scatter_(1, idxs, vals)
            mask.scatter_(1, idxs, False)
            pred_tok = torch.cat([tokens, code.reshape(b, 1, h, w)], dim=1)
        assert mask.sum() == 0, f'Not all tokens were predicted. {mask.sum()} tokens left.'
        return pred_tok

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
