"""Space-time transformer attention on the HIP kernels (drop-in for reference genie/module/attention.py).

Constructor signatures, attributes and ``state_dict`` keys follow the reference.  The reference's numerics
quirks are reproduced on purpose: the attention scale is ``n_head * d_head**-0.5``
(attention.py:195); rotary embedding is applied to the block input BEFORE LayerNorm and the rotated+normed
tensor is Q, K and V (attention.py:219-226); with ``d_inp == n_head * d_head`` there are no projections at all.

What runs where:  rotary + LayerNorm -> ``genie_rotary_layernorm_fwd``;  head split / SDPA / head merge / the
rearranges around them -> ``genie_attention_fwd`` (address arithmetic, nothing is moved);  FFN = GroupNorm +
Conv3d(+ residual in the epilogue) -> the conv/norm kernels.  The optional condition projections
(``to_k`` / ``to_v``, attention.py:128-129) run on the skinny-linear kernels (functional.linear); ``to_q`` / ``to_out`` of non-default
blueprints (d_inp != n_head * d_head) are plain library GEMMs.
"""
from __future__ import annotations

import math
from typing import Literal, Optional, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch import Tensor

from .. import _hip
from .. import conv as _conv
from .. import functional as GF
from ..cl import empty_like_cl, is_cl, pitch_of, to_cl
from ..conv import same_spec
from ..utils import default, exists
from .misc import ForwardBlock
from .video import Conv3d


class RotaryEmbedding(nn.Module):
    """reference attention.py:17-103.  Only the frequency table lives here; the rotation itself is fused into the
    LayerNorm prologue kernel.  Angles are formed in fp32 (they reach thousands of radians for the '2d' kind)."""

    def __init__(self, dim: int, kind: Literal['1d', '2d', 'const'] = '1d', theta=10000, max_freq=10, num_freq=1,
                 learned_freq=False, interpolate_factor=1., theta_rescale_factor=1.) -> None:
        super().__init__()
        theta *= theta_rescale_factor ** (dim / (dim - 2))
        match kind:
            case '1d':
                freq = 1. / (theta ** (torch.arange(0, dim, 2)[:(dim // 2)].float() / dim))
            case '2d':
                freq = torch.linspace(1., max_freq / 2, dim // 2) * math.pi
            case 'const':
                freq = torch.ones(num_freq).float()
        self.freq = nn.Parameter(freq, requires_grad=learned_freq)
        if learned_freq:
            raise NotImplementedError('RotaryEmbedding: learned_freq is not implemented on the HIP path')
        assert interpolate_factor >= 1.
        self.interpolate_factor = interpolate_factor
        self.default_seq_dim = -2
        self._table = (None, None)

    def table(self, npos: int, feat: int) -> Tensor:
        """fp32 [npos][feat]: (cos, sin) of pos * freq_i in slots (2i, 2i + 1)."""
        rot_dim = 2 * self.freq.numel()
        assert rot_dim <= feat, f'feature dimension {feat} is not of sufficient size to rotate in all the positions {rot_dim}'
        if rot_dim != feat:
            raise NotImplementedError('RotaryEmbedding: partial rotation (rot_dim < features) is not implemented on the HIP path')
        key = (npos, feat, self.freq._version, self.freq.data_ptr())
        if self._table[0] != key:
            pos = torch.arange(npos, device=self.freq.device) / self.interpolate_factor
            ang = pos.float()[:, None] * self.freq.detach().float()[None, :]                  # attention.py:60-62
            tab = torch.stack((ang.cos(), ang.sin()), dim=-1).reshape(npos, feat).contiguous()
            self._table = (key, tab)
        return self._table[1]


class Adapter(nn.Module):
    """reference attention.py:105-149 (projection holder; the head split is address arithmetic in the kernel)."""

    def __init__(self, qry_dim: int, n_head: int, d_head: int, key_dim: int | None = None, val_dim: int | None = None,
                 block=nn.Linear, qry_kwargs: dict = {}, key_kwargs: dict = {}, val_kwargs: dict = {}, bias: bool = False) -> None:
        super().__init__()
        key_dim = default(key_dim, qry_dim)
        val_dim = default(val_dim, key_dim)
        if isinstance(block, type) and issubclass(block, nn.Module):
            block = (block, block, block)
        hid = n_head * d_head
        self.to_q = block[0](qry_dim, hid, bias=bias, **qry_kwargs) if qry_dim != hid else nn.Identity()
        self.to_k = block[1](key_dim, hid, bias=bias, **key_kwargs) if key_dim != hid else nn.Identity()
        self.to_v = block[2](val_dim, hid, bias=bias, **val_kwargs) if val_dim != hid else nn.Identity()
        self.n_head = n_head


class _Merge(nn.Module):
    """placeholder for the reference's Rearrange('b h n d -> b n (h d)') at ``to_out.0`` (no parameters)."""

    def forward(self, x):
        return x


# ------------------------------------------------------------------------------------------------
# fused block function:  out = Attn(LN(rot(x))) [+ x]
# ------------------------------------------------------------------------------------------------
def _attn_variant(direction: str, S: int, d_head: int) -> str:
    """Name under which conv.LaunchProfiler files an attention call: the kernel family the C side picks for it (attention.hip /
    attention_lean.hip / attention_narrow.hip) and the roofline that bounds it (MFMA from S >= 1024, traffic below)."""
    fam = 'attn_narrow' if d_head < 32 else ('attn_small' if S <= 32 else 'attn')
    return f'{fam}_{direction}[{"mfma" if S >= 1024 and d_head >= 32 else "hbm"}]'


COND_SMALL = __import__('os').environ.get('GENIE_ATTN_COND_SMALL', '1') != '0'      # A/B: 0 = conditioned short sequences through the general kernels + torch sums


class _AttnFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x: Tensor, gamma: Tensor, beta: Tensor, table: Optional[Tensor], kext: Optional[Tensor], vext: Optional[Tensor],
                mode: str, n_head: int, d_head: int, scale: float, causal: bool, add_resid: bool, eps: float, dropout_p: float = 0.0, seed: int = 0):
        lib = _hip.load_library()
        b, c, t, h, w = x.shape
        assert pitch_of(x) == c and c == n_head * d_head
        hw, ntok = h * w, b * t * h * w
        if mode == 'space':
            nseq, S = b * t, hw
            qmap = (1, hw * c, 0, c)
            pos_div, pos_mod = 1, hw
        else:
            nseq, S = b * hw, t
            qmap = (hw, t * hw * c, c, hw * c)
            pos_div, pos_mod = hw, t
        u = empty_like_cl(x)
        stats = torch.empty(ntok * 2, dtype=torch.float32, device=x.device)
        g32, b32 = GF._f32(gamma), GF._f32(beta)
        P = _hip.ptr
        _hip.check(lib.genie_rotary_layernorm_fwd(P(x), P(u), ntok, c, c, P(table), pos_div, pos_mod, P(g32), P(b32), eps, P(stats),
                                                  _hip.stream_ptr()), 'genie_rotary_layernorm_fwd')
        if kext is None:
            k, v, kvmap, Sk = u, u, qmap, S
        else:
            # condition rows: (B, Sk, C) contiguous; broadcast over the other axis of the video
            k, v, Sk = kext, vext, kext.shape[1]
            inner = t if mode == 'space' else hw
            kvmap = (inner, Sk * c, 0, c)
        out = empty_like_cl(x)
        keep = any(ctx.needs_input_grad)
        oattn = empty_like_cl(x) if (keep and add_resid) else None
        lse = torch.empty(ntok * n_head, dtype=torch.float32, device=x.device)
        prof = _conv.PROFILER if _conv.PROFILER is not None and not _conv.PROFILER.only_triple else None
        t0 = prof.begin() if prof is not None else None
        if dropout_p > 0.0:                               # general kernels with the counter-based mask (include/genie_hip.h, ABI 13); backward re-derives it
            _hip.check(lib.genie_attention_fwd_dropout(P(u), P(k), P(v), P(x) if add_resid else None, P(out), P(oattn), P(lse), nseq, n_head, d_head, S, Sk,
                                                       _hip.i64(qmap), _hip.i64(kvmap), _hip.i64(qmap), scale, 1 if causal else 0, c, dropout_p, seed,
                                                       _hip.stream_ptr()), 'genie_attention_fwd_dropout')
        else:
            _hip.check(lib.genie_attention_fwd(P(u), P(k), P(v), P(x) if add_resid else None, P(out), P(oattn), P(lse), nseq, n_head, d_head, S, Sk,
                                               _hip.i64(qmap), _hip.i64(kvmap), _hip.i64(qmap), scale, 1 if causal else 0, c, _hip.stream_ptr()),
                       'genie_attention_fwd')
        if prof is not None:                              # dense count 4 S Sk C per sequence; traffic: read u (q, k / v), resid, write out
            prof.end(_attn_variant('fwd', S, d_head), f'attention fwd {mode} S={S} Sk={Sk} C={c} nseq={nseq}', 4.0 * S * Sk * c * nseq, t0,
                     bytes_=(4.0 if add_resid else 3.0) * ntok * c * 2)
        ctx.cfg = (mode, n_head, d_head, scale, causal, add_resid, eps, qmap, kvmap, nseq, S, Sk, pos_div, pos_mod)
        ctx.drop = (float(dropout_p), int(seed))
        ctx.save_for_backward(x, gamma, beta, table, kext, vext, u, oattn if oattn is not None else out, lse, stats)
        return out

    @staticmethod
    def backward(ctx, dout: Tensor):
        x, gamma, beta, table, kext, vext, u, out, lse, stats = ctx.saved_tensors
        mode, n_head, d_head, scale, causal, add_resid, eps, qmap, kvmap, nseq, S, Sk, pos_div, pos_mod = ctx.cfg
        lib = _hip.load_library()
        b, c, t, h, w = x.shape
        ntok = b * t * h * w
        dout = to_cl(dout)
        du = empty_like_cl(x)
        D = GF.workspace(3 * ntok * n_head, x.device, 'attn_D')            # D, lse * log2 e, -D (ABI 10)
        P = _hip.ptr
        dk = dv = dk32 = dv32 = None
        inner = t if mode == 'space' else h * w
        # conditioned SHORT sequences (the LatentAction decoder's temporal attention over the per-clip action codes): the packed kernel sums the
        # condition rows' gradients over the pixels of a clip itself (genie_attention_bwd_cond) -- rounds 1-5 took them per sequence (two tensors of the
        # activation's size) and summed them with torch
        tp = 8 if S <= 8 else (16 if S <= 16 else 32)
        dropout_p, seed = ctx.drop
        cond_small = (kext is not None and S == Sk and S <= 32 and d_head in (32, 64) and inner % (32 // tp) == 0 and COND_SMALL and dropout_p == 0.0)
        if kext is None:
            k, v, dkvmap = u, u, None
        else:
            k, v = kext, vext
            if cond_small:
                dk32 = torch.zeros((b, Sk, c), dtype=torch.float32, device=x.device)
                dv32 = torch.zeros_like(dk32)
            else:
                dk = torch.empty((nseq, Sk, c), dtype=torch.bfloat16, device=x.device)
                dv = torch.empty_like(dk)
            dkvmap = _hip.i64((1, Sk * c, 0, c))
        prof = _conv.PROFILER if _conv.PROFILER is not None and not _conv.PROFILER.only_triple else None
        t0 = prof.begin() if prof is not None else None
        if cond_small:
            _hip.check(lib.genie_attention_bwd_cond(P(u), P(k), P(v), P(out), None, P(dout), P(lse), P(D), P(du), P(dk32), P(dv32), nseq, n_head, d_head, S,
                                                    _hip.i64(qmap), _hip.i64(kvmap), _hip.i64(qmap), scale, 1 if causal else 0, c, c, ntok, _hip.stream_ptr()),
                       'genie_attention_bwd_cond')
        elif dropout_p > 0.0:
            _hip.check(lib.genie_attention_bwd_dropout(P(u), P(k), P(v), P(out), None, P(dout), P(lse), P(D), P(du), P(dk), P(dv),
                                                       nseq, n_head, d_head, S, Sk, _hip.i64(qmap), _hip.i64(kvmap), _hip.i64(qmap), dkvmap, scale,
                                                       1 if causal else 0, c, ntok, dropout_p, seed, _hip.stream_ptr()), 'genie_attention_bwd_dropout')
        else:
            _hip.check(lib.genie_attention_bwd(P(u), P(k), P(v), P(out), None, P(dout), P(lse), P(D), P(du), P(dk), P(dv),
                                               nseq, n_head, d_head, S, Sk, _hip.i64(qmap), _hip.i64(kvmap), _hip.i64(qmap), dkvmap, scale,
                                               1 if causal else 0, c, ntok, _hip.stream_ptr()), 'genie_attention_bwd')
        if prof is not None:                              # 2.5 x the forward count (five GEMM units of the maths); traffic: u, out, dout read, du written (+ dq re-read)
            prof.end(_attn_variant('bwd', S, d_head), f'attention bwd {mode} S={S} Sk={Sk} C={c} nseq={nseq}', 10.0 * S * Sk * c * nseq, t0,
                     bytes_=5.0 * ntok * c * 2)
        dx = empty_like_cl(x)
        need_g, need_b = ctx.needs_input_grad[1], ctx.needs_input_grad[2]
        direct = GF._direct(gamma) and GF._direct(beta) and gamma.is_leaf and beta.is_leaf and gamma.dtype == torch.float32
        dgamma = (GF._grad_buffer(gamma) if direct else torch.zeros(c, dtype=torch.float32, device=x.device)) if need_g else None
        dbeta = (GF._grad_buffer(beta) if direct else torch.zeros(c, dtype=torch.float32, device=x.device)) if need_b else None
        _hip.check(lib.genie_rotary_layernorm_bwd(P(x), P(du), P(dout) if add_resid else None, P(dx), ntok, c, c, P(table), pos_div, pos_mod,
                                                  P(GF._f32(gamma)), P(stats), P(dgamma), P(dbeta), _hip.stream_ptr()), 'genie_rotary_layernorm_bwd')
        dkext = dvext = None
        if kext is not None:
            if cond_small:
                dkext, dvext = dk32.to(kext.dtype), dv32.to(vext.dtype)
            else:
                dkext = dk.reshape(b, inner, Sk, c).float().sum(1).to(kext.dtype)
                dvext = dv.reshape(b, inner, Sk, c).float().sum(1).to(vext.dtype)
        return (dx, None if direct else dgamma, None if direct else dbeta, None, dkext, dvext, None, None, None, None, None, None, None, None, None)


class Attention(nn.Module):
    """reference attention.py:154-239 (base class; see SpatialAttention / TemporalAttention for the video forms)."""

    _mode = None      # 'space' | 'time'
    _kind = None      # rotary kind

    def __init__(self, n_head: int, d_head: int, d_inp: int | None = None, d_out: int | None = None, bias: bool = False,
                 scale: float | None = None, causal: bool = False, dropout: float = 0.0, **kwargs) -> None:
        super().__init__()
        hid = n_head * d_head
        self.d_inp = default(d_inp, hid)
        self.d_out = default(d_out, self.d_inp)
        self.norm = nn.LayerNorm(hid)
        self.embed = nn.Identity()
        self.to_qkv = Adapter(qry_dim=self.d_inp, n_head=n_head, d_head=d_head, bias=bias, **kwargs)
        self.to_out = nn.Sequential(_Merge(), nn.Linear(hid, self.d_out, bias=bias) if self.d_out != hid else nn.Identity())
        self.scale = default(scale, n_head * d_head ** -0.5)          # QUIRK: operator precedence (attention.py:195)
        self.causal, self.dropout = causal, dropout
        self.n_head, self.d_head = n_head, d_head
        if not 0.0 <= float(dropout) < 1.0:
            raise ValueError(f'Attention: dropout={dropout} not in [0, 1)')
        self.last_dropout_seed = None                               # the seed of the most recent forward (tests rebuild the mask from it)

    def _video_forward(self, video: Tensor, cond: Optional[Tensor], mask, transpose: bool, add_resid: bool) -> Tensor:
        if mask is not None:
            raise NotImplementedError('attention masks are not supported (the reference raises on them too, SURVEY.md section 0)')
        hid = self.n_head * self.d_head
        if not isinstance(self.to_qkv.to_q, nn.Identity):
            # the reference applies LayerNorm(n_head * d_head) to a d_inp-wide tensor and raises here
            raise RuntimeError(f'Attention: d_inp={self.d_inp} != n_head * d_head={hid} is not runnable (nor in the reference)')
        x = video if transpose else video.permute(0, 4, 1, 2, 3)           # logical (B, C, T, H, W)
        x = to_cl(x)
        b, c, t, h, w = x.shape
        if c != hid:
            if isinstance(self.embed, RotaryEmbedding) and c < 2 * self.embed.freq.numel():
                # what the reference's rotary embedding says first (attention.py:87) -- e.g. the shipped config/tokenize.yaml, whose
                # channels-first 64-feature stem meets channels-last 512-wide blocks
                raise AssertionError(f'feature dimension {c} is not of sufficient size to rotate in all the positions {2 * self.embed.freq.numel()}')
            raise RuntimeError(f'Attention: input has {c} features, expected n_head * d_head = {hid}')
        npos = h * w if self._mode == 'space' else t
        table = self.embed.table(npos, c) if isinstance(self.embed, RotaryEmbedding) else None
        kext = vext = None
        if cond is not None:
            kc = cond.to(torch.float32)
            lin = lambda mod, t: GF.linear(t, mod.weight, mod.bias, out_dtype=torch.bfloat16) if isinstance(mod, nn.Linear) else mod(t)
            kext = lin(self.to_qkv.to_k, kc).to(torch.bfloat16).contiguous()       # Linear(key_dim -> C): csrc/linear_small.hip when key_dim <= 32
            vext = lin(self.to_qkv.to_v, kc).to(torch.bfloat16).contiguous()
        p_drop, seed = float(self.dropout), 0
        if p_drop > 0.0:
            # QUIRK kept: the reference passes `dropout_p=self.dropout` to the FUNCTIONAL sdpa (attention.py:229), which has no training switch --
            # the weights are dropped in eval mode too.  The seed comes from torch's CPU generator (torch.manual_seed reproduces a run; no device
            # sync) and is baked into the launch arguments, so a captured step would replay one mask for ever: refused.
            if x.is_cuda and torch.cuda.is_current_stream_capturing():
                raise RuntimeError('Attention: dropout > 0 cannot be captured in a hipGraph (the mask seed is a launch argument)')
            seed = int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).item())
            self.last_dropout_seed = seed
        out = _AttnFn.apply(x, self.norm.weight, self.norm.bias, table, kext, vext, self._mode, self.n_head, self.d_head, float(self.scale),
                            bool(self.causal), add_resid, self.norm.eps, p_drop, seed)
        if not isinstance(self.to_out[1], nn.Identity):
            o = self.to_out[1](out.permute(0, 2, 3, 4, 1).float())
            out = to_cl(o.permute(0, 4, 1, 2, 3))
        return out if transpose else out.permute(0, 2, 3, 4, 1)


class SpatialAttention(Attention):
    """reference attention.py:241-307: self-attention over the H*W pixels of each frame, '2d' rotary on the flattened index."""
    _mode = 'space'

    def __init__(self, n_head: int, d_head: int, d_inp: int | None = None, d_out: int | None = None, bias: bool = False,
                 embed: bool = True, scale: float | None = None, causal: bool = False, dropout: float = 0.0, transpose: bool = False,
                 **kwargs) -> None:
        super().__init__(n_head, d_head, d_inp, d_out, bias, scale, causal, dropout, **kwargs)
        self.embed = RotaryEmbedding(self.d_inp, kind='2d') if embed else nn.Identity()
        self.transpose = transpose

    def forward(self, video: Tensor, cond: Tensor | None = None, mask: Tensor | None = None, transpose: bool | None = None,
                _add_resid: bool = False) -> Tensor:
        return self._video_forward(video, cond, mask, default(transpose, self.transpose), _add_resid)


class TemporalAttention(Attention):
    """reference attention.py:309-371: causal self-attention over the T frames of each pixel, '1d' rotary."""
    _mode = 'time'

    def __init__(self, n_head: int, d_head: int, d_inp: int | None = None, d_out: int | None = None, bias: bool = False,
                 embed: bool = True, scale: float | None = None, causal: bool = False, dropout: float = 0.0, transpose: bool = False,
                 **kwargs) -> None:
        super().__init__(n_head, d_head, d_inp, d_out, bias, scale, causal, dropout, **kwargs)
        self.embed = RotaryEmbedding(self.d_inp, kind='1d') if embed else nn.Identity()
        self.transpose = transpose

    def forward(self, video: Tensor, cond: Tensor | None = None, mask: Tensor | None = None, transpose: bool | None = None,
                _add_resid: bool = False) -> Tensor:
        return self._video_forward(video, cond, mask, default(transpose, self.transpose), _add_resid)


class SpaceTimeAttention(nn.Module):
    """reference attention.py:373-474:  x = space(x) + x;  x = temp(x) + x;  x = ffn(x) + ffn_skip(x)  with
    ffn = ForwardBlock(GroupNorm(n_head) -> Conv3d layers over (C, *hid_dim, d_out) with GELU between them, misc.py:71-104); the shipped
    blueprints use hid_dim = None, d_out = None: one Conv3d(C, C, k, padding=(k-1)//2, bias=bias)."""

    def __init__(self, n_head, d_head, d_inp: int | None = None, d_out: int | None = None, hid_dim=None, bias: bool = False,
                 embed=True, scale: float | None = None, dropout: float = 0.0, kernel_size: int = 3, transpose: bool = False,
                 time_attn_kw: dict = {}, space_attn_kw: dict = {}) -> None:
        super().__init__()
        if isinstance(n_head, int):
            n_head = (n_head, n_head)
        if isinstance(d_head, int):
            d_head = (d_head, d_head)
        if isinstance(embed, bool):
            embed = (embed, embed)
        self.space_attn = SpatialAttention(n_head=n_head[0], d_head=d_head[0], d_inp=d_inp, d_out=None, bias=bias, scale=scale,
                                           embed=embed[0], causal=False, dropout=dropout, transpose=transpose, **space_attn_kw)
        self.temp_attn = TemporalAttention(n_head=n_head[1], d_head=d_head[1], d_inp=None, d_out=None, bias=bias, scale=scale,
                                           embed=embed[1], causal=True, dropout=dropout, transpose=transpose, **time_attn_kw)
        ffn = ForwardBlock(n_head[1] * d_head[1], out_dim=d_out, hid_dim=hid_dim, num_groups=n_head[1], bias=bias, block=nn.Conv3d,
                           kernel_size=kernel_size, padding=(kernel_size - 1) // 2)
        self.ffn = nn.Sequential(_Merge(), ffn, _Merge())          # indices 0 / 2 are the reference's Rearrange layers
        self.in_channels = default(d_inp, n_head[0] * d_head[0])
        self.out_channels = default(d_out, n_head[1] * d_head[1])
        space_hid, time_hid = d_head[0] * n_head[0], d_head[1] * n_head[1]
        if exists(d_inp) and d_inp != space_hid:
            raise NotImplementedError('SpaceTimeAttention: d_inp different from n_head * d_head is not implemented on the HIP path (the '
                                      'reference cannot run it either: its Attention normalises d_inp features and splits n_head * d_head, SURVEY.md section 0)')
        if space_hid != time_hid:
            raise NotImplementedError('SpaceTimeAttention: the spatial and the temporal sub-layer must have the same width on the HIP path')
        self.time_skip, self.space_skip = nn.Identity(), nn.Identity()
        # reference attention.py:453: a 1x1x1 projection of the skip path when the feed-forward changes the width
        self.ffn_skip = Conv3d(time_hid, d_out, (1, 1, 1), same_spec(time_hid, d_out, (1, 1, 1)), bias=True) if exists(d_out) and time_hid != d_out else nn.Identity()
        self.transpose = transpose

    def forward(self, video: Tensor, cond=None, mask: Tensor | None = None) -> Tensor:
        if not isinstance(cond, tuple):
            cond = (cond, cond)
        space_cond, time_cond = cond
        tr = self.transpose
        x = to_cl(video if tr else video.permute(0, 4, 1, 2, 3))
        x = self.space_attn(x, cond=space_cond, mask=mask, transpose=True, _add_resid=True)
        x = self.temp_attn(x, cond=time_cond, mask=mask, transpose=True, _add_resid=True)
        # feed-forward (reference misc.py:86-98): GroupNorm, then Conv3d layers with the activation between them (none after the last unless
        # `last_act`); hid_dim = None is the shipped blueprints' single conv.  The skip connection rides in the last conv's GEMM epilogue
        # whenever nothing follows that conv and the skip is the identity.
        net = self.ffn[1].net
        gn, layers = net[0], list(net)[1:]
        skip = x if isinstance(self.ffn_skip, nn.Identity) else self.ffn_skip(x)
        y = GF.group_norm(x, gn.num_groups, gn.weight, gn.bias, gn.eps)
        for i, layer in enumerate(layers):
            conv, act = layer[0], layer[1]
            fuse = i == len(layers) - 1 and isinstance(act, nn.Identity)
            y = conv(y, resid=skip if fuse else None)
            if not isinstance(act, nn.Identity):
                y = act(y)
            if i == len(layers) - 1 and not fuse:
                y = y + skip
        x = y
        return x if tr else x.permute(0, 2, 3, 4, 1)
