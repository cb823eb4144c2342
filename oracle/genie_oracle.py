"""CPU oracle for the open-genie hot path.

TEST INFRASTRUCTURE -- NOT PRODUCT CODE.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import this file; the product path (``open-genie_amd/``)
never does and fails loudly when its HIP library is missing.

What it is: a *functional* restatement (no ``nn.Module``; explicit ``state_dict`` + blueprint in,
tensors out) of the reference's algorithm, in fp32 (or fp64 on request) on the CPU.  The reference
delegates all arithmetic to PyTorch ATen (pinned ``torch==2.3.0``, requirements.txt:1-4; this image
has torch 2.10) -- conv3d, group_norm, layer_norm, softmax, scaled-dot-product attention.  ATen *is*
the reference's arithmetic provider, so the restatement calls the same ATen CPU primitives
(``torch.nn.functional``) where the reference does and spells out everything the reference builds
on top of them (padding arithmetic, rotary angles, the attention-scale quirk, LFQ bit packing, the
loss expressions, MaskGIT scheduling) from its documented behaviour.  Every function cites the
reference ``file:line`` it follows (paths relative to ``/root/reference``).

Parity pinning: the reference holds NO golden vectors for this path -- its tests
assert shapes only.  The oracle is therefore pinned against outputs of the reference itself:
``tests/golden/make_golden.py`` imports the real reference in the build container and commits
seeded input/output fixtures under ``tests/golden/``; ``tests/test_oracle_golden.py`` checks this
file against them on every run, and ``tests/test_oracle_vs_reference.py`` checks it against the
live reference whenever ``/root/reference`` is present.

Known reference quirks reproduced on purpose are marked QUIRK.
"""
from __future__ import annotations

import copy
import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F
from torch import Tensor

SD = Dict[str, Tensor]


def _triple(v) -> Tuple[int, int, int]:
    return (v, v, v) if isinstance(v, int) else tuple(v)


def _get(sd: SD, key: str) -> Optional[Tensor]:
    return sd.get(key, None)


# ----------------------------------------------------------------------------------------------
# Optional bf16 emulation of the HIP path's STORES (test infrastructure; off by default)
# ----------------------------------------------------------------------------------------------
# The reference computes in fp32 throughout.  The HIP path computes every operator in fp32 too, but keeps activations and
# activation gradients in HBM as bf16: one rounding where a kernel stores its result.  With ``set_rounding('bf16_at_stores')``
# the oracle rounds at exactly those places -- forward values at every operator output the HIP path stores (`_st`), and the
# gradient that an operator's backward kernel stores for its input (`_gr`) -- and nowhere else (statistics, softmax, losses and
# parameter gradients stay fp32, as in the kernels).  The model-level parity tests compare the HIP path with THIS mode so that
# their tolerance measures implementation error, not the (expected, much larger) bf16-vs-fp32 representation error; the fp32
# numbers are still reported next to it.  ``None`` (default) is the reference's arithmetic, bit for bit what it was.
_ROUNDING: Optional[str] = None


def set_rounding(mode: Optional[str]) -> Optional[str]:
    """mode: None (fp32 reference arithmetic) or 'bf16_at_stores'.  Returns the previous mode."""
    global _ROUNDING
    if mode not in (None, 'bf16_at_stores'):
        raise ValueError(f'unknown rounding mode {mode!r}')
    old, _ROUNDING = _ROUNDING, mode
    return old


class rounding:
    """``with rounding('bf16_at_stores'): ...``"""

    def __init__(self, mode: Optional[str]):
        self.mode = mode

    def __enter__(self):
        self.old = set_rounding(self.mode)
        return self

    def __exit__(self, *exc):
        set_rounding(self.old)
        return False


def _bf16(x: Tensor) -> Tensor:
    return x.to(torch.bfloat16).to(x.dtype)


class _StoreFn(torch.autograd.Function):
    """A tensor the HIP path keeps in HBM: value rounded to bf16 on the way forward, its gradient on the way back."""

    @staticmethod
    def forward(ctx, x):
        return _bf16(x)

    @staticmethod
    def backward(ctx, g):
        return _bf16(g)


class _GradFn(torch.autograd.Function):
    """Identity forward; the gradient stored for this operator input is bf16."""

    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return _bf16(g)


def _st(x: Tensor) -> Tensor:
    return _StoreFn.apply(x) if _ROUNDING else x


def _gr(x: Optional[Tensor]) -> Optional[Tensor]:
    return _GradFn.apply(x) if (_ROUNDING and x is not None and x.requires_grad) else x


# ----------------------------------------------------------------------------------------------
# a1  CausalConv3d                                   genie/module/video.py:106-200
# ----------------------------------------------------------------------------------------------
def causal_pad_amounts(kernel_size, stride=(1, 1, 1), dilation=(1, 1, 1), padding=None):
    """(time_front, height, width) zero padding.  video.py:154-164.

    time pad = (kt-1)*dil_t + (1 - stride_t), all of it in FRONT (causal); h/w pad = (k-1)//2 on
    both sides unless given.
    """
    kt, kh, kw = _triple(kernel_size)
    st, _, _ = _triple(stride)
    dt, _, _ = _triple(dilation)
    if padding is None or isinstance(padding, int):
        padding = (padding, padding)
    tp = (kt - 1) * dt + (1 - st)
    hp = padding[0] if padding[0] is not None else (kh - 1) // 2
    wp = padding[1] if padding[1] is not None else (kw - 1) // 2
    return tp, hp, wp


def causal_conv3d(x: Tensor, w: Tensor, b: Optional[Tensor], stride=(1, 1, 1), dilation=(1, 1, 1),
                  padding=None, resid: Optional[Tensor] = None, round_dx: bool = True) -> Tensor:
    """video.py:178-192: F.pad(x, (wp, wp, hp, hp, tp, 0)) then conv3d without padding.
    (`resid`: an addend the caller sums with the result -- video.py:648 -- passed in so that the rounding mode can round the SUM once,
    as the GEMM epilogue does.)"""
    stride, dilation = _triple(stride), _triple(dilation)
    tp, hp, wp = causal_pad_amounts(w.shape[2:], stride, dilation, padding)
    x = F.pad(_gr(x) if round_dx else x, (wp, wp, hp, hp, tp, 0))       # (a negative tp crops, as F.pad does in the reference)
    y = F.conv3d(x, w, b, stride=stride, dilation=dilation)
    return _st(y if resid is None else y + resid)


def causal_conv_transpose3d(x: Tensor, w: Tensor, b: Optional[Tensor], stride=(1, 1, 1), dilation=(1, 1, 1), space_pad=None) -> Tensor:
    """video.py:202-277: nn.ConvTranspose3d(padding=(0, kh // 2, kw // 2)) (weight (in, out, kt, kh, kw)), output cropped to
    (t * T, h * H, w * W) (video.py:263-267)."""
    stride, dilation = _triple(stride), _triple(dilation)
    if space_pad is None or isinstance(space_pad, int):
        space_pad = (space_pad, space_pad)
    hp = space_pad[0] if space_pad[0] is not None else w.shape[3] // 2
    wp = space_pad[1] if space_pad[1] is not None else w.shape[4] // 2
    t, h, ww = x.shape[2:]
    y = F.conv_transpose3d(_gr(x), w, b, stride=stride, padding=(0, hp, wp), dilation=dilation)
    return _st(y[..., :t * stride[0], :h * stride[1], :ww * stride[2]])


def spacetime_upsample(x: Tensor, w: Tensor, b: Optional[Tensor], time_factor: int = 2, space_factor: int = 2) -> Tensor:
    """video.py:432-455: nn.ConvTranspose3d(kernel = stride = (tf, sf, sf))."""
    return _st(F.conv_transpose3d(_gr(x), w, b, stride=(time_factor, space_factor, space_factor)))


def conv3d_same(x: Tensor, w: Tensor, b: Optional[Tensor], resid: Optional[Tensor] = None, round_dx: bool = True) -> Tensor:
    """nn.Conv3d(k, padding=(k-1)//2) as used by VideoResidualBlock (video.py:580-586, 614-620)
    and the ST-block FFN (attention.py:429-438).  Symmetric zero padding: NOT causal (QUIRK 5)."""
    pad = tuple((k - 1) // 2 for k in w.shape[2:])
    y = F.conv3d(_gr(x) if round_dx else x, w, b, padding=pad)
    return _st(y if resid is None else y + resid)


# ----------------------------------------------------------------------------------------------
# a3  GroupNorm / SiLU                               torch.nn.GroupNorm, torch.nn.SiLU
# ----------------------------------------------------------------------------------------------
def group_norm(x: Tensor, groups: int, w: Optional[Tensor], b: Optional[Tensor], eps: float = 1e-5) -> Tensor:
    """Spelled-out group norm: per (sample, group) biased variance over (C/G, *spatial)."""
    n, c = x.shape[:2]
    xg = x.reshape(n, groups, -1)
    mean = xg.mean(dim=-1, keepdim=True)
    var = xg.var(dim=-1, unbiased=False, keepdim=True)
    y = ((xg - mean) * torch.rsqrt(var + eps)).reshape(x.shape)
    shape = (1, c) + (1,) * (x.dim() - 2)
    if w is not None:
        y = y * w.reshape(shape)
    if b is not None:
        y = y + b.reshape(shape)
    return y


def silu(x: Tensor) -> Tensor:
    return x * torch.sigmoid(x)


# ----------------------------------------------------------------------------------------------
# a6  AdaptiveGroupNorm                              genie/module/norm.py:55-69
# ----------------------------------------------------------------------------------------------
def adaptive_group_norm(x: Tensor, cond: Tensor, sd: SD, prefix: str, num_groups: int, eps: float = 1e-5) -> Tensor:
    """norm.py:55-69: group_norm(x) * Linear_std(mean_{t,h,w} cond) + Linear_avg(mean cond)."""
    y = group_norm(_gr(x), num_groups, _get(sd, prefix + 'weight'), _get(sd, prefix + 'bias'), eps)
    c = _gr(cond).reshape(cond.shape[0], cond.shape[1], -1).mean(-1)     # norm.py:62
    std = F.linear(c, sd[prefix + 'std.weight'], sd[prefix + 'std.bias'])
    shape = std.shape + (1,) * (x.dim() - 2)
    y = y * std.reshape(shape)
    if prefix + 'avg.weight' in sd:
        avg = F.linear(c, sd[prefix + 'avg.weight'], sd[prefix + 'avg.bias'])
        y = y + avg.reshape(shape)
    return _st(y)


# ----------------------------------------------------------------------------------------------
# a7  BlurPooling3d                                  genie/module/video.py:22-56, 487-537
# ----------------------------------------------------------------------------------------------
def blur_kernel(kernel_size) -> Tensor:
    """video.py:22-56.  Pascal-triangle taps, outer product over (t, h, w), normalised to sum 1.
    QUIRK: the w taps use comb(kw-1, i) for i in range(kt) (video.py:47) -- identical for cubes."""
    kt, kh, kw = _triple(kernel_size)
    t = torch.tensor([math.comb(kt - 1, i) for i in range(kt)], dtype=torch.float32)
    h = torch.tensor([math.comb(kt - 1, i) for i in range(kt)], dtype=torch.float32)
    w = torch.tensor([math.comb(kw - 1, i) for i in range(kt)], dtype=torch.float32)
    k = t[:, None, None] * h[None, :, None] * w[None, None, :]
    return k / k.sum()


def blur_pool3d(x: Tensor, kernel_size, time_factor: int, space_factor, num_groups: int = 1,
                out_channels: Optional[int] = None) -> Tensor:
    """video.py:516-534.  QUIRK 6: with num_groups=1 this is a DENSE conv whose every (out, in) tap
    is the same blur kernel -> each output channel is the blur of the SUM over input channels."""
    kt, kh, kw = _triple(kernel_size)
    sf = (space_factor, space_factor) if isinstance(space_factor, int) else tuple(space_factor)
    c = x.shape[1]
    o = out_channels if out_channels is not None else c
    ker = blur_kernel((kt, kh, kw)).to(x.dtype)
    ker = ker[None, None].expand(o, c // num_groups, kt, kh, kw)
    pad = ((kt - 1) // 2, (kh - 1) // 2, (kw - 1) // 2)
    return _st(F.conv3d(_gr(x), ker, stride=(time_factor, *sf), padding=pad, groups=num_groups))


# ----------------------------------------------------------------------------------------------
# a2  VideoResidualBlock                             genie/module/video.py:539-656
# ----------------------------------------------------------------------------------------------
def _act(name: str, x: Tensor) -> Tensor:
    if name in ('swish', 'silu'):
        return silu(x)
    if name == 'relu':
        return F.relu(x)
    if name == 'gelu':
        return F.gelu(x)
    if name == 'leaky':
        return F.leaky_relu(x)
    raise ValueError(name)


def video_residual_block(x: Tensor, sd: SD, prefix: str, in_channels: int, out_channels: Optional[int] = None,
                         kernel_size=3, num_groups: int = 1, downsample=None, use_causal: bool = False,
                         use_norm: bool = True, use_blur: bool = True, act_fn: str = 'swish', **_) -> Tensor:
    """video.py:588-648: main = [GN, act, conv k, (down), GN, act, conv k]; res = [(down), conv 1].
    QUIRK 5: plain (non-causal) convs and num_groups=1 by default; res always has the 1x1x1 conv."""
    if isinstance(downsample, int):
        downsample = (downsample, downsample)
    ks = _triple(kernel_size)

    def conv(t: Tensor, key: str, resid: Optional[Tensor] = None, round_dx: bool = True) -> Tensor:
        if use_causal:   # CausalConv3d wraps the conv as .conv3d and ignores the passed `padding`
            w, b = sd[prefix + key + '.conv3d.weight'], _get(sd, prefix + key + '.conv3d.bias')
            pad = tuple((k - 1) // 2 for k in w.shape[2:])
            # video.py:580-586 passes padding=(pt, ph, pw); CausalConv3d reads padding[0], padding[1]
            # as (height, width) pads (video.py:157-158): for the 1x1x1 res conv padding is None.
            if w.shape[2:] == (1, 1, 1):
                return causal_conv3d(t, w, b, resid=resid, round_dx=round_dx)
            return causal_conv3d(t, w, b, padding=(pad[0], pad[1]), resid=resid, round_dx=round_dx)
        return conv3d_same(t, sd[prefix + key + '.weight'], _get(sd, prefix + key + '.bias'), resid=resid, round_dx=round_dx)

    def down(t: Tensor, key: str, ch: int) -> Tensor:
        if downsample is None:
            return t
        tf, sf = downsample
        if use_blur:
            return blur_pool3d(t, ks, tf, sf, num_groups=num_groups)
        w, b = sd[prefix + key + '.go_down.conv3d.weight'], _get(sd, prefix + key + '.go_down.conv3d.bias')
        return causal_conv3d(t, w, b, stride=(tf, sf, sf))

    def norm_act(t: Tensor, key: str) -> Tensor:       # (one pass, one store on the HIP path)
        t = _gr(t)
        if use_norm:
            t = group_norm(t, num_groups, sd[prefix + key + '.weight'], sd[prefix + key + '.bias'])
        return _st(_act(act_fn, t))

    out_channels = out_channels if out_channels is not None else in_channels
    # (fused block on the HIP path: the shortcut's backward-data GEMM takes the main branch's stored input gradient as its epilogue
    # addend -- dx = bf16(dgrad_res(dy) + bf16(d_main)) -- so the shortcut's own dx is never rounded separately)
    res = conv(down(x, 'res.0', in_channels), 'res.1', round_dx=downsample is not None)
    h = conv(norm_act(x, 'main.0'), 'main.2')
    h = down(h, 'main.3', out_channels)
    return conv(norm_act(h, 'main.4'), 'main.6', resid=res)       # video.py:648: main(x) + res(x)


# ----------------------------------------------------------------------------------------------
# a4 / a5  SpaceTimeDownsample, DepthToSpaceTimeUpsample   video.py:457-483, 379-430
# ----------------------------------------------------------------------------------------------
def spacetime_downsample(x: Tensor, sd: SD, prefix: str, time_factor: int = 2, space_factor: int = 2, **_) -> Tensor:
    w, b = sd[prefix + 'go_down.conv3d.weight'], _get(sd, prefix + 'go_down.conv3d.bias')
    return causal_conv3d(x, w, b, stride=(time_factor, space_factor, space_factor))


def depth_to_spacetime(y: Tensor, time_factor: int, space_factor: int) -> Tensor:
    """'b (c p q r) t h w -> b c (t p) (h q) (w r)'  (video.py:403-408)."""
    b, cpqr, t, h, w = y.shape
    p, q, r = time_factor, space_factor, space_factor
    c = cpqr // (p * q * r)
    y = y.reshape(b, c, p, q, r, t, h, w).permute(0, 1, 5, 2, 6, 3, 7, 4)
    return y.reshape(b, c, t * p, h * q, w * r)


def depth2spacetime_upsample(x: Tensor, sd: SD, prefix: str, time_factor: int = 2, space_factor: int = 2, **_) -> Tensor:
    w, b = sd[prefix + 'go_up.0.conv3d.weight'], _get(sd, prefix + 'go_up.0.conv3d.bias')
    return depth_to_spacetime(causal_conv3d(x, w, b), time_factor, space_factor)


# ----------------------------------------------------------------------------------------------
# a8  LookupFreeQuantization                          genie/module/quantization.py:77-133
# ----------------------------------------------------------------------------------------------
def lfq_bit_mask(d: int) -> Tensor:
    """quantization.py:72: MSB first (QUIRK 7)."""
    return 2 ** torch.arange(d - 1, -1, -1)


def lfq_codebook(d: int, num_codebook: int = 1) -> Tensor:
    """quantization.py:74-75: code k -> {-1,+1}^d, bit i of k (MSB first) set -> +1."""
    codes = torch.arange((2 ** d) * num_codebook)[:, None] & lfq_bit_mask(d)
    return 2 * (codes != 0).float() - 1


def lfq_indices_numpy(x: np.ndarray) -> np.ndarray:
    """Integer part of LFQ in plain numpy: idx = sum_i (x[..., i] > 0) << (d-1-i)   (quantization.py:97-98).
    x: (..., d) float.  Returns int64 (...)."""
    d = x.shape[-1]
    weights = (1 << np.arange(d - 1, -1, -1)).astype(np.int64)
    return ((x > 0).astype(np.int64) * weights).sum(-1)


def entropy(p: Tensor, eps: float = 1e-6) -> Tensor:
    """quantization.py:17-28: the clamp is INSIDE the log only (QUIRK 7)."""
    return -(p * torch.log(p.clamp(min=eps))).sum(dim=-1)


def lfq_forward(x: Tensor, sd: SD, prefix: str, codebook_dim: int, num_codebook: int = 1, training: bool = False,
                beta: float = 100., transpose: bool = False, commit_weight: float = 0.25,
                entropy_weight: float = 0.1, diversity_weight: float = 1.):
    """quantization.py:77-133.  Returns ((out, idxs), loss-or-None)."""
    if transpose:
        x = x.movedim(1, -1)                                   # 'b d ... -> b ... d'
    lead = x.shape[:-1]
    z = x.reshape(x.shape[0], -1, x.shape[-1])                 # pack 'b * d'
    if prefix + 'proj_inp.weight' in sd:
        z = F.linear(z, sd[prefix + 'proj_inp.weight'], _get(sd, prefix + 'proj_inp.bias'))
    b, n, _ = z.shape
    z = z.reshape(b, n, num_codebook, codebook_dim)
    quant = torch.sign(z)                                      # sign(0) = 0 ...
    mask = lfq_bit_mask(codebook_dim)
    idxs = ((z > 0).to(torch.int32) * mask.to(torch.int32)).sum(-1)   # ... but the bit uses > 0
    idxs = idxs.to(torch.int64)                                # einops-reduce sum of int32 -> int64
    code = (z + (quant - z).detach()) if training else quant   # straight-through estimator
    code = code.reshape(b, n, num_codebook * codebook_dim)
    out = code
    if prefix + 'proj_out.weight' in sd:
        out = F.linear(code, sd[prefix + 'proj_out.weight'], _get(sd, prefix + 'proj_out.bias'))
    out = out.reshape(*lead, out.shape[-1])
    if transpose:
        out = out.movedim(-1, 1)
    idxs = idxs.reshape(*lead, num_codebook).squeeze()         # QUIRK 7: drops EVERY size-1 dim
    if not training:
        return (out, idxs), None
    cb = lfq_codebook(codebook_dim, num_codebook).to(z.dtype)  # (K, d)
    logits = 2 * torch.einsum('bncd,jd->bncj', z, cb)
    prob = (logits * beta).softmax(dim=-1).reshape(b * n, num_codebook, -1)
    avg_prob = prob.mean(dim=0)
    inp_ent = entropy(prob).mean()
    avg_ent = entropy(avg_prob).mean()
    ent_loss = inp_ent + diversity_weight * avg_ent            # QUIRK 7: plus, not minus
    commit = F.mse_loss(z, quant.detach(), reduction='mean')
    loss = ent_loss * entropy_weight + commit * commit_weight
    return (out, idxs), loss


def lfq_entropy_terms_factored(z: Tensor, beta: float = 100., eps: float = 1e-6, split: Optional[int] = None):
    """Exact factorisation of the LFQ entropy terms used by the HIP kernel (LFQ section).

    softmax(2*beta * z.c) over c in {-1,+1}^d is a product of d independent two-point distributions,
    so with the bits split into a high group (first `split` dims, MSB side) and a low group,
    p[token, code] = A[token, hi(code)] * B[token, lo(code)].  The clamp inside the log prevents a
    closed form, so the per-code terms are still enumerated -- but from A (x) B, never from an
    N x 2^d logits matrix.  z: (N, d) fp.  Returns (inp_ent, avg_ent) as the reference defines them
    for num_codebook == 1 (quantization.py:116-123)."""
    n, d = z.shape
    split = d // 2 if split is None else split
    zz = z.double()

    def part(zs: Tensor) -> Tensor:
        cb = lfq_codebook(zs.shape[1]).double()
        return ((2 * beta) * (zs @ cb.T)).softmax(-1)           # logits 2*z.c, times beta -> softmax

    a, bb = part(zz[:, :split]), part(zz[:, split:])
    inp = torch.zeros((), dtype=torch.float64)
    for i in range(n):
        p = torch.outer(a[i], bb[i]).reshape(-1)
        inp = inp - (p * torch.log(p.clamp(min=eps))).sum()
    inp_ent = inp / n
    avg = (a.T @ bb / n).reshape(-1)
    avg_ent = -(avg * torch.log(avg.clamp(min=eps))).sum()
    return inp_ent, avg_ent


# ----------------------------------------------------------------------------------------------
# a9  RotaryEmbedding                                 genie/module/attention.py:17-103
# ----------------------------------------------------------------------------------------------
def rotary_freq(dim: int, kind: str, theta: float = 10000., max_freq: float = 10.) -> Tensor:
    """attention.py:33-39."""
    if kind == '1d':
        return 1. / (theta ** (torch.arange(0, dim, 2)[: dim // 2].float() / dim))
    if kind == '2d':
        return torch.linspace(1., max_freq / 2, dim // 2) * math.pi
    raise ValueError(kind)


def rotary_apply(x: Tensor, freq: Tensor) -> Tensor:
    """attention.py:48-100 for seq_dim=-2: x (..., S, C), freq (C/2,).

    angle[s, 2i] = angle[s, 2i+1] = s * freq[i]; out[2i] = x[2i] cos - x[2i+1] sin,
    out[2i+1] = x[2i+1] cos + x[2i] sin.  Rotation covers ALL C features (QUIRK 2); the position is
    the flattened index along S -- for the '2d' kind too (the flattened h*w index)."""
    s, c = x.shape[-2], x.shape[-1]
    pos = torch.arange(s, dtype=freq.dtype)
    ang = torch.repeat_interleave(pos[:, None] * freq[None, :], 2, dim=-1)       # (S, C)
    assert ang.shape[-1] <= c, f'feature dimension {c} is not of sufficient size to rotate in all the positions {ang.shape[-1]}'
    x1, x2 = x[..., 0::2], x[..., 1::2]
    rot = torch.stack((-x2, x1), dim=-1).reshape(x.shape)
    return (x * ang.cos() + rot * ang.sin()).to(x.dtype)


# ----------------------------------------------------------------------------------------------
# a10  Attention core                                  genie/module/attention.py:154-239
# ----------------------------------------------------------------------------------------------
def attention_core(x: Tensor, sd: SD, prefix: str, n_head: int, d_head: int, causal: bool, rotary_kind: Optional[str],
                   cond: Optional[Tensor] = None, scale: Optional[float] = None, drop_keep: Optional[Tensor] = None, drop_p: float = 0.0) -> Tensor:
    """x: (B', S, C) with C == n_head*d_head (QUIRK 3: to_q/to_k/to_v/to_out are Identity then).

    attention.py:219-236: rotary (before the norm, QUIRK 2) -> LayerNorm -> q = k = v = that tensor
    unless `cond` is given, in which case k = to_k(cond) and v = to_v(k) ... see below.
    QUIRK 1: scale = n_head * d_head**-0.5 (attention.py:195).
    `drop_keep` (B', n_head, Sq, Sk; 1 = kept) with `drop_p`: attention.py:229 `dropout_p=self.dropout` -- sdpa's documented form
    `torch.dropout(softmax(S), p) @ V` with the Bernoulli draws GIVEN (torch's Philox stream is backend-specific; the HIP path's counter-based
    mask is exported by genie_attention_dropout_mask and applied here)."""
    c = n_head * d_head
    # (no separate gradient store for x: the rotary+LayerNorm backward kernel adds the skip branch's gradient in fp32 and stores the sum)
    if rotary_kind is not None:
        x = rotary_apply(x, sd[prefix + 'embed.freq'])
    q = _st(F.layer_norm(x, (c,), sd[prefix + 'norm.weight'], sd[prefix + 'norm.bias']))     # one pass, one store on the HIP path
    key = q if cond is None else cond
    val = key                                                       # attention.py:222-223
    # Adapter.forward attention.py:133-149: key = default(key, qry); val = default(val, key)
    k = _st(F.linear(key, sd[prefix + 'to_qkv.to_k.weight'], _get(sd, prefix + 'to_qkv.to_k.bias'))) \
        if prefix + 'to_qkv.to_k.weight' in sd else key
    v = _st(F.linear(val, sd[prefix + 'to_qkv.to_v.weight'], _get(sd, prefix + 'to_qkv.to_v.bias'))) \
        if prefix + 'to_qkv.to_v.weight' in sd else val

    def heads(t: Tensor) -> Tensor:
        return t.reshape(t.shape[0], t.shape[1], n_head, d_head).transpose(1, 2)

    qh, kh, vh = heads(q), heads(k), heads(v)
    scale = scale if scale is not None else n_head * d_head ** -0.5

    def sdpa(qc: Tensor, kc: Tensor, vc: Tensor, keep: Optional[Tensor] = None) -> Tensor:
        att = torch.matmul(qc, kc.transpose(-1, -2)) * scale
        if causal:
            sq, sk = att.shape[-2:]
            m = torch.ones(sq, sk, dtype=torch.bool).tril()          # SDPA is_causal: top-left aligned
            att = att.masked_fill(~m, float('-inf'))
        w = att.softmax(dim=-1)
        if keep is not None:
            w = w * keep.to(w.dtype) / (1.0 - drop_p)
        return torch.matmul(w, vc)

    nseq, sq, sk = qh.shape[0], qh.shape[2], kh.shape[2]
    if drop_keep is not None:
        assert tuple(drop_keep.shape) == (nseq, n_head, sq, sk) and 0.0 < drop_p < 1.0
        o = sdpa(qh, kh, vh, drop_keep)
    elif nseq > 1 and nseq * n_head * sq * sk > (1 << 27):
        # long sequences (LAM at 64x64: 16 x 4 x 4096 x 4096 scores = 4.3 GB per tensor): same arithmetic one sequence at a time, the score
        # matrices recomputed in backward instead of kept
        from torch.utils.checkpoint import checkpoint
        outs = [checkpoint(sdpa, qh[i:i + 1], kh[i:i + 1], vh[i:i + 1], use_reentrant=False) if torch.is_grad_enabled() and
                (qh.requires_grad or kh.requires_grad or vh.requires_grad) else sdpa(qh[i:i + 1], kh[i:i + 1], vh[i:i + 1]) for i in range(nseq)]
        o = torch.cat(outs, dim=0)
    else:
        o = sdpa(qh, kh, vh)
    return o.transpose(1, 2).reshape(x.shape[0], x.shape[1], c)


def spatial_attention(video: Tensor, sd: SD, prefix: str, n_head: int, d_head: int, transpose: bool, embed: bool = True,
                      cond: Optional[Tensor] = None, scale=None, drop_keep: Optional[Tensor] = None, drop_p: float = 0.0) -> Tensor:
    """attention.py:279-307.  video (B, C, T, H, W) if transpose else (B, T, H, W, C)."""
    x = video.permute(0, 2, 3, 4, 1) if transpose else video
    b, t, h, w, c = x.shape
    seq = x.reshape(b * t, h * w, c)
    if cond is not None:
        cond = cond.repeat_interleave(t, dim=0)                     # 'b hw c -> (b t) hw c'
    out = attention_core(seq, sd, prefix, n_head, d_head, False, '2d' if embed else None, cond, scale, drop_keep, drop_p)
    out = out.reshape(b, t, h, w, c)
    return out.permute(0, 4, 1, 2, 3) if transpose else out


def temporal_attention(video: Tensor, sd: SD, prefix: str, n_head: int, d_head: int, transpose: bool, embed: bool = True,
                       cond: Optional[Tensor] = None, scale=None, drop_keep: Optional[Tensor] = None, drop_p: float = 0.0) -> Tensor:
    """attention.py:347-371.  Causal over T, one sequence per (b, h, w)."""
    x = video.permute(0, 2, 3, 4, 1) if transpose else video           # b t h w c
    b, t, h, w, c = x.shape
    seq = x.permute(0, 2, 3, 1, 4).reshape(b * h * w, t, c)
    if cond is not None:
        cond = cond.repeat_interleave(h * w, dim=0)                 # 'b t c -> (b h w) t c'
    out = attention_core(seq, sd, prefix, n_head, d_head, True, '1d' if embed else None, cond, scale, drop_keep, drop_p)
    out = out.reshape(b, h, w, t, c).permute(0, 3, 1, 2, 4)
    return out.permute(0, 4, 1, 2, 3) if transpose else out


# ----------------------------------------------------------------------------------------------
# a12  SpaceTimeAttention block                        genie/module/attention.py:373-474
# ----------------------------------------------------------------------------------------------
def space_time_block(video: Tensor, sd: SD, prefix: str, n_head: int, d_head: int, transpose: bool = False,
                     embed=True, kernel_size: int = 3, cond=None, scale=None, **_) -> Tensor:
    """attention.py:456-474: x = space(x)+x; x = temp(x)+x; x = ffn(x)+x with
    ffn = GroupNorm(n_head, C) -> Conv3d(C, C, k, padding=(k-1)//2, bias=False)  (QUIRK 4)."""
    if not isinstance(cond, tuple):
        cond = (cond, cond)
    if isinstance(embed, bool):
        embed = (embed, embed)
    # (rounding mode: each sub-layer's `f(x) + x` leaves its kernel as ONE bf16 store; the skip branch's gradient is the stored dout itself)
    video = _st(spatial_attention(video, sd, prefix + 'space_attn.', n_head, d_head, transpose, embed[0], cond[0], scale) + video)
    video = _st(temporal_attention(video, sd, prefix + 'temp_attn.', n_head, d_head, transpose, embed[1], cond[1], scale) + video)
    x = video if transpose else video.permute(0, 4, 1, 2, 3)
    y = _st(group_norm(_gr(x), n_head, sd[prefix + 'ffn.1.net.0.weight'], sd[prefix + 'ffn.1.net.0.bias']))
    y = conv3d_same(y, sd[prefix + 'ffn.1.net.1.0.weight'], _get(sd, prefix + 'ffn.1.net.1.0.bias'), resid=x)     # ffn(x) + x
    return y if transpose else y.permute(0, 2, 3, 4, 1)


# ----------------------------------------------------------------------------------------------
# blueprint walking                                    genie/module/__init__.py:71-93
# ----------------------------------------------------------------------------------------------
def expand_blueprint(blueprint) -> List[Tuple[str, dict, bool]]:
    """One (name, kwargs, has_ext) per instantiated layer; pops has_ext / n_rep like the reference
    (on a deep copy: the reference mutates the caller's dicts)."""
    out = []
    for desc in copy.deepcopy(list(blueprint)):
        if isinstance(desc, str):
            desc = (desc, {})
        name, kw = desc
        kw = dict(kw)
        has_ext = kw.pop('has_ext', False)
        n_rep = kw.pop('n_rep', 1)
        out.extend([(name, kw, has_ext)] * n_rep)
    return out


def run_layer(name: str, kw: dict, x: Tensor, sd: SD, prefix: str, cond=None, use_cond: bool = False) -> Tensor:
    if name == 'causal-conv3d':
        return causal_conv3d(x, sd[prefix + 'conv3d.weight'], _get(sd, prefix + 'conv3d.bias'),
                             stride=kw.get('stride', (1, 1, 1)), dilation=kw.get('dilation', (1, 1, 1)),
                             padding=kw.get('padding', None))
    if name == 'video-residual':
        return video_residual_block(x, sd, prefix, **kw)
    if name == 'spacetime_downsample':
        return spacetime_downsample(x, sd, prefix, **kw)
    if name == 'depth2spacetime_upsample':
        return depth2spacetime_upsample(x, sd, prefix, **kw)
    if name == 'group_norm':
        return _st(group_norm(_gr(x), kw['num_groups'], _get(sd, prefix + 'weight'), _get(sd, prefix + 'bias'), kw.get('eps', 1e-5)))
    if name == 'adaptive_group_norm':
        return adaptive_group_norm(x, cond, sd, prefix, kw['num_groups'], kw.get('eps', 1e-5))
    if name == 'silu':
        return _st(silu(_gr(x)))
    if name == 'space-time_attn':
        return space_time_block(x, sd, prefix, cond=cond if use_cond else None, **kw)
    raise ValueError(f'Unknown module name: {name}')


# ----------------------------------------------------------------------------------------------
# a13  VideoTokenizer                                  genie/tokenizer.py:307-387
# ----------------------------------------------------------------------------------------------
def _run_layers(x: Tensor, sd: SD, desc, prefix: str, cond: Optional[Tensor], lo: int = 0, hi: Optional[int] = None) -> Tensor:
    """The layer loop of tokenizer.py:314-315 / 326-328 (layers [lo, hi) of the expanded blueprint; default: all).  In the rounding mode
    a 'group_norm' directly followed by 'silu' is ONE stored tensor (the HIP path runs the pair as one pass); the arithmetic is the same
    either way."""
    layers = expand_blueprint(desc)
    if hi is not None:
        layers = layers[:hi]
    i = lo
    while i < len(layers):
        name, kw, has_ext = layers[i]
        if _ROUNDING and name == 'group_norm' and not has_ext and i + 1 < len(layers) and layers[i + 1][0] == 'silu' and not layers[i + 1][2]:
            p = f'{prefix}{i}.'
            x = _st(silu(group_norm(_gr(x), kw['num_groups'], _get(sd, p + 'weight'), _get(sd, p + 'bias'), kw.get('eps', 1e-5))))
            i += 2
            continue
        x = run_layer(name, kw, x, sd, f'{prefix}{i}.', cond=cond if has_ext else None, use_cond=has_ext)
        i += 1
    return x


def tokenizer_encode(video: Tensor, sd: SD, enc_desc) -> Tensor:
    return _run_layers(video, sd, enc_desc, 'enc_layers.', None)


def tokenizer_decode(quant: Tensor, sd: SD, dec_desc, cond: Optional[Tensor] = None) -> Tensor:
    cond = quant if cond is None else cond                           # tokenizer.py:324
    return _run_layers(quant, sd, dec_desc, 'dec_layers.', cond)


def tokenizer_tokenize(video: Tensor, sd: SD, enc_desc, d_codebook: int, n_codebook: int = 1, beta: float = 100.):
    """tokenizer.py:332-350 (eval-mode LFQ, transpose=True)."""
    enc = tokenizer_encode(video, sd, enc_desc)
    (q, idx), _ = lfq_forward(enc, sd, 'quant.', d_codebook, n_codebook, training=False, beta=beta, transpose=True)
    return q, idx


def tokenizer_forward_hotpath(video: Tensor, sd: SD, enc_desc, dec_desc, d_codebook: int, n_codebook: int = 1,
                              beta: float = 100., quant_loss_weight: float = 1., **lfq_kw):
    """R-fwd: tokenizer.py:352-387 with the GAN and perceptual terms omitted (they
    cannot execute offline): loss = mse(rec, video) + quant_loss * w.  Training-mode LFQ."""
    enc = tokenizer_encode(video, sd, enc_desc)
    (q, idx), q_loss = lfq_forward(enc, sd, 'quant.', d_codebook, n_codebook, training=True, beta=beta,
                                   transpose=True, **lfq_kw)
    rec = tokenizer_decode(q, sd, dec_desc)
    rec_loss = F.mse_loss(rec, video)
    loss = rec_loss + q_loss * quant_loss_weight
    return loss, (rec_loss, q_loss), rec, idx


# ----------------------------------------------------------------------------------------------
# a14-a16  DynamicsModel                               genie/dynamics.py:44-195
# ----------------------------------------------------------------------------------------------
def dynamics_forward(tokens: Tensor, act_id: Tensor, sd: SD, desc) -> Tuple[Tensor, Tensor]:
    """dynamics.py:44-64.  tokens (B,T,H,W) int64, act_id (B,T) int64 -> logits (B,T,H,W,V)."""
    x = _st(F.embedding(tokens, sd['tok_emb.weight']) + F.embedding(act_id, sd['act_emb.0.weight'])[:, :, None, None, :])
    for i, (name, kw, has_ext) in enumerate(expand_blueprint(desc)):
        x = run_layer(name, kw, x, sd, f'dec_layers.{i}.')             # dynamics.py:59: no cond passed
    logits = _st(F.linear(_gr(x), sd['head.weight'], sd['head.bias']))
    return logits, logits[:, -1]


def dynamics_loss(tokens: Tensor, act_id: Tensor, mask: Tensor, sd: SD, desc, fill: int = 0) -> Tensor:
    """dynamics.py:66-99 with an explicit mask.  QUIRK 9: targets are read AFTER masked_fill, so every
    target equals `fill`; `mask.squeeze()` requires batch >= 2."""
    tokens = torch.masked_fill(tokens, mask, fill)
    logits, _ = dynamics_forward(tokens, act_id, sd, desc)
    m = mask.squeeze()
    return F.cross_entropy(logits[m].reshape(-1, logits.shape[-1]), tokens[m].reshape(-1))


def linear_cross_entropy(h: Tensor, weight: Tensor, bias: Optional[Tensor], target: Tensor, valid: Optional[Tensor] = None) -> Tensor:
    """The tail of DynamicsModel.compute_loss as one function of the gathered rows: ``self.head`` (dynamics.py:62, nn.Linear built at
    :32) followed by ``cross_entropy(logits[mask], tokens[mask])`` (dynamics.py:89-97, mean reduction).  h (M, D), weight (V, D),
    bias (V,) or None, target (M,) int64; `valid` (M,) bool selects the rows that enter the mean (None = all) -- the boolean gather of
    dynamics.py:92-93 applied to rows that were already gathered."""
    logits = F.linear(h, weight, bias)
    if valid is not None:
        logits, target = logits[valid], target[valid]
    return F.cross_entropy(logits, target)


def maskgit_schedule(steps: int, shape: Sequence[int], which: str = 'linear') -> Tensor:
    """dynamics.py:167-195."""
    n = int(np.prod(shape))
    t = torch.linspace(1, 0, steps)
    if which == 'linear':
        s = 1 - t
    elif which == 'cosine':
        s = torch.cos(t * math.pi * .5)
    elif which == 'arccos':
        s = torch.acos(t) / (math.pi * .5)
    else:
        raise ValueError(f'Unknown schedule type: {which}')
    sch = ((s / s.sum()) * n).round().int().clamp(min=1)
    sch[-1] += n - sch.sum()
    return sch


def sample_from_uniform(prob: Tensor, u: Tensor) -> Tensor:
    """Inverse-CDF categorical sampling with an injected uniform ('hard parts'):
    pred = #(cumsum(prob) <= u * total), clamped.  Stands in for torch.multinomial (dynamics.py:141),
    whose RNG stream differs between devices; both the oracle and the HIP path use THIS rule so token
    ids can be compared bit for bit.  prob (N, V) fp32, u (N,) in [0, 1)."""
    cdf = prob.double().cumsum(-1)
    thr = (u.double() * cdf[:, -1])[:, None]
    return (cdf <= thr).sum(-1).clamp(max=prob.shape[-1] - 1)


def maskgit_sample_step(logits: Tensor, u: Tensor, temp: float = 1.) -> Tuple[Tensor, Tensor]:
    """dynamics.py:138-143 for one step: prob = softmax(logits / temp) packed to (rows, V); pred = categorical draw (inverse CDF
    with the injected uniforms instead of torch.multinomial); conf = prob[pred].  Returns (pred int64 (rows,), conf fp32 (rows,))."""
    prob = torch.softmax(logits / temp, dim=-1).reshape(-1, logits.shape[-1])
    pred = sample_from_uniform(prob, u)
    return pred, prob.gather(-1, pred[:, None])[:, 0]


def maskgit_paint_step(conf: Tensor, pred: Tensor, mask: Tensor, code: Tensor, k: int) -> None:
    """dynamics.py:146-158 for one step, in place: conf[~mask] = -inf; idxs = topk(conf, k); code[idxs] = pred[idxs];
    mask[idxs] = False.  conf / pred / mask / code: (B, h*w)."""
    conf = conf.clone()
    conf[~mask] = -math.inf
    idxs = torch.topk(conf, k=k, dim=-1).indices
    code.scatter_(1, idxs, pred.gather(-1, idxs).to(code.dtype))
    mask.scatter_(1, idxs, False)


def dynamics_generate(tokens: Tensor, act_id: Tensor, sd: SD, desc, uniforms: Tensor, steps: int = 10,
                      which: str = 'linear', temp: float = 1., masked_tok: int = 0, trace: Optional[list] = None) -> Tensor:
    """dynamics.py:101-165 with injected noise.  uniforms: (steps, B*H*W).  QUIRK 9: painted codes are
    never fed back into tok_id (dynamics.py:128,136).  `trace` (optional list) receives per-step dicts with the probabilities'
    source logits, the draws and the mask before painting (used by the parity tests to explain every mismatch)."""
    b, t, h, w = tokens.shape
    schedule = maskgit_schedule(steps, (h, w), which)
    mask = torch.ones(b, h * w, dtype=torch.bool)
    code = torch.full((b, h * w), masked_tok, dtype=tokens.dtype)
    tok_id = torch.cat([tokens, code.reshape(b, 1, h, w)], dim=1)
    act = torch.cat([act_id, torch.zeros(b, 1, dtype=act_id.dtype)], dim=1)
    pred_tok = tok_id
    for step, k in enumerate(schedule.tolist()):
        if mask.sum() == 0:
            break
        _, logits = dynamics_forward(tok_id, act, sd, desc)
        pred, conf = maskgit_sample_step(logits, uniforms[step], temp)
        pred, conf = pred.reshape(b, -1), conf.reshape(b, -1)
        if trace is not None:
            trace.append({'logits': logits, 'pred': pred.clone(), 'conf': conf.clone(), 'mask_before': mask.clone(), 'k': k})
        maskgit_paint_step(conf, pred, mask, code, k)
        pred_tok = torch.cat([tokens, code.reshape(b, 1, h, w)], dim=1)
    assert mask.sum() == 0
    return pred_tok


# ----------------------------------------------------------------------------------------------
# a17  LatentAction, repaired (R-lam)    genie/action.py:111-176
# ----------------------------------------------------------------------------------------------
def latent_action_encode(video: Tensor, sd: SD, enc_desc) -> Tensor:
    """action.py:111-122 up to the encoder output: proj_in then the encoder blueprint (no condition, no mask)."""
    x = causal_conv3d(video, sd['proj_in.conv3d.weight'], _get(sd, 'proj_in.conv3d.bias'))
    for i, (name, kw, _) in enumerate(expand_blueprint(enc_desc)):
        x = run_layer(name, kw, x, sd, f'enc_layers.{i}.')
    return x


def latent_action_to_act(enc_video: Tensor, sd: SD) -> Tensor:
    """action.py:83-90, 124: 'b c t ... -> b t (c ...)' then Linear(no bias) -> (B, T, d) pre-quantisation action latent."""
    b, c, t = enc_video.shape[:3]
    return _st(F.linear(_gr(enc_video).permute(0, 2, 1, 3, 4).reshape(b, t, -1), _gr(sd['to_act.1.weight'])))


def latent_action_decode(enc_video: Tensor, q_act: Tensor, sd: SD, dec_desc) -> Tensor:
    """action.py:136-160: decoder blueprint with the quantised action as TEMPORAL condition of the `has_ext` layers, then proj_out."""
    y = enc_video
    for i, (name, kw, has_ext) in enumerate(expand_blueprint(dec_desc)):
        y = run_layer(name, kw, y, sd, f'dec_layers.{i}.', cond=(None, q_act if has_ext else None), use_cond=True)
    return causal_conv3d(y, sd['proj_out.conv3d.weight'], _get(sd, 'proj_out.conv3d.bias'))


def latent_action_forward(video: Tensor, sd: SD, enc_desc, dec_desc, d_codebook: int, training: bool = True,
                          quant_loss_weight: float = 1., beta: float = 100., trace: Optional[dict] = None):
    """action.py:111-176 with the three R-lam repairs applied by the CALLER's blueprint (transpose=True ST
    blocks with n_head*d_head == n_embd, 'depth2spacetime_upsample', LFQ input_dim = d_codebook)."""
    enc_video = latent_action_encode(video, sd, enc_desc)
    act = latent_action_to_act(enc_video, sd)                                             # 'b c t ... -> b t (c ...)' + Linear
    if trace is not None:
        trace.update(enc_video=enc_video, act=act)
    (q_act, idxs), q_loss = lfq_forward(act, sd, 'quant.', d_codebook, training=training, beta=beta, transpose=False)
    recon = latent_action_decode(enc_video, q_act, sd, dec_desc)
    rec_loss = F.mse_loss(recon, video)
    loss = rec_loss + (q_loss * quant_loss_weight if q_loss is not None else 0)
    return idxs, loss, (rec_loss, q_loss), recon


# ----------------------------------------------------------------------------------------------
# f2  GAN critic      genie/module/image.py:105-163, discriminator.py:17-114, loss.py:109-164, utils.py:30-56
# ----------------------------------------------------------------------------------------------
def image_residual_block(x: Tensor, sd: SD, prefix: str, inp_channel: int, out_channel: Optional[int] = None, kernel_size=3,
                         padding=1, num_groups: int = 1, downsample: Optional[int] = None) -> Tensor:
    """image.py:105-163.  main = GN, LeakyReLU, Conv2d, GN, LeakyReLU, Conv2d[, SpaceDownsample]; res = Conv2d(1x1, stride = downsample)
    when out_channel is given, else identity.  SpaceDownsample (image.py:86-103): 'b c (h p) (w q) -> b (c p q) h w' then Conv2d 1x1."""
    res = x
    if out_channel is not None:
        res = F.conv2d(x, sd[prefix + 'res.weight'], _get(sd, prefix + 'res.bias'), stride=downsample or 1)
    h = F.leaky_relu(F.group_norm(x, num_groups, sd[prefix + 'main.0.weight'], sd[prefix + 'main.0.bias']))
    h = F.conv2d(h, sd[prefix + 'main.2.weight'], _get(sd, prefix + 'main.2.bias'), padding=padding)
    h = F.leaky_relu(F.group_norm(h, num_groups, sd[prefix + 'main.3.weight'], sd[prefix + 'main.3.bias']))
    h = F.conv2d(h, sd[prefix + 'main.5.weight'], _get(sd, prefix + 'main.5.bias'), padding=padding)
    if downsample:
        b, c, hh, ww = h.shape
        f = downsample
        h = h.reshape(b, c, hh // f, f, ww // f, f).permute(0, 1, 3, 5, 2, 4).reshape(b, c * f * f, hh // f, ww // f)
        h = F.conv2d(h, sd[prefix + 'main.6.go_up.1.weight'], _get(sd, prefix + 'main.6.go_up.1.bias'))
    return h + res


def frame_discriminator(image: Tensor, sd: SD, prefix: str = '', model_dim: int = 64, dim_mults=(1, 2, 4), down_step=(None, 2, 2),
                        kernel_size=3, num_groups: int = 1, **_) -> Tensor:
    """discriminator.py:99-114 with use_attn=False (the attention variants cannot run in the reference): Conv2d stem;
    per stage a residual block, then `attn(out) + out` and `ff(out) + out` with attn = ff = Identity, i.e. the features are
    doubled twice; Conv2d + LeakyReLU + flatten + Linear -> one logit per image."""
    dims = [model_dim * m for m in dim_mults]
    out = F.conv2d(image, sd[prefix + 'proj_in.weight'], _get(sd, prefix + 'proj_in.bias'), padding=1)
    for i, ((ci, co), down) in enumerate(zip(zip(dims[:-1], dims[1:]), down_step)):
        out = image_residual_block(out, sd, f'{prefix}core.{i}.0.', ci, co, kernel_size=kernel_size, num_groups=num_groups, downsample=down)
        out = out + out
        out = out + out
    out = F.leaky_relu(F.conv2d(out, sd[prefix + 'to_logits.0.weight'], _get(sd, prefix + 'to_logits.0.bias'), padding=1))
    return F.linear(out.flatten(1), sd[prefix + 'to_logits.3.weight'], sd[prefix + 'to_logits.3.bias'])[:, 0]


def pick_frames(video: Tensor, frame_idxs: Tensor) -> Tensor:
    """utils.py:30-56 with explicit indices: frame_idxs (b * k,) -> (b * k, c, h, w), k consecutive entries per clip."""
    b = video.shape[0]
    batch_idxs = torch.repeat_interleave(torch.arange(b), frame_idxs.numel() // b)
    return video[batch_idxs, :, frame_idxs]


def gan_loss(rec_video: Tensor, inp_video: Tensor, train_gen: bool, frame_idxs: Tensor, sd: SD, prefix: str = 'gan_crit.disc.', **disc_kw) -> Tensor:
    """loss.py:147-164 (frames critic): hinge loss; the fakes are detached when the critic is being trained."""
    fake, real = pick_frames(rec_video, frame_idxs), pick_frames(inp_video, frame_idxs)
    if train_gen:
        return -frame_discriminator(fake, sd, prefix, **disc_kw).mean()
    fs = frame_discriminator(fake.detach(), sd, prefix, **disc_kw)
    rs = frame_discriminator(real, sd, prefix, **disc_kw)
    return (F.relu(1 + fs) + F.relu(1 - rs)).mean()


def tokenizer_forward_gan(video: Tensor, sd: SD, enc_desc, dec_desc, d_codebook: int, idx_gen: Tensor, idx_dis: Tensor, disc_kw: dict,
                          gan_loss_weight: float = 1., quant_loss_weight: float = 1., beta: float = 100., **lfq_kw):
    """tokenizer.py:352-387 with the GAN critic on and the perceptual term omitted (VGG16 weights do not exist offline):
    loss = mse + gen * w + dis * w + quant * w_q; idx_gen / idx_dis are the frame choices of the two gan_crit calls."""
    enc = tokenizer_encode(video, sd, enc_desc)
    (q, idx), q_loss = lfq_forward(enc, sd, 'quant.', d_codebook, 1, training=True, beta=beta, transpose=True, **lfq_kw)
    rec = tokenizer_decode(q, sd, dec_desc)
    rec_loss = F.mse_loss(rec, video)
    gen = gan_loss(rec, video, True, idx_gen, sd, **disc_kw)
    dis = gan_loss(rec, video, False, idx_dis, sd, **disc_kw)
    loss = rec_loss + gen * gan_loss_weight + dis * gan_loss_weight + q_loss * quant_loss_weight
    return loss, (rec_loss, gen, dis, q_loss), rec
