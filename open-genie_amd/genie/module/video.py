"""3-D convolution building blocks on the HIP gather-GEMM (drop-in for reference genie/module/video.py).

Same constructor signatures, attributes and ``state_dict`` layout as the reference classes, so reference
checkpoints load; ``forward`` takes any (N, C, T, H, W) CUDA tensor and returns a CL tensor (logical NCTHW,
bf16, channels-last strides).  There is no CPU path.
"""
from __future__ import annotations

import math
from abc import ABC
from typing import Optional, Tuple

import torch
import torch.nn as nn
from torch import Tensor

from .. import functional as GF
from ..cl import to_cl
from ..conv import ConvSpec, causal_spec, causal_time_crop, same_spec
from ..utils import default, exists


def _triple(v) -> Tuple[int, int, int]:
    return (v, v, v) if isinstance(v, int) else tuple(v)


class Conv3d(nn.Module):
    """Conv3d whose arithmetic is ``genie_conv_igemm``; parameters laid out like ``nn.Conv3d`` (keys
    ``weight`` (Cout, Cin, kt, kh, kw) / ``bias``), stored channels_last_3d so that packing is a cast.

    ``spec`` fixes the padding rule: symmetric (``nn.Conv3d(padding=(k-1)//2)``, reference video.py:580-586)
    or causal (reference video.py:154-164)."""

    def __init__(self, in_channels: int, out_channels: int, kernel_size, spec: ConvSpec, bias: bool = True) -> None:
        super().__init__()
        ks = _triple(kernel_size)
        w = torch.empty(out_channels, in_channels, *ks)
        nn.init.kaiming_uniform_(w, a=math.sqrt(5))                 # nn.Conv3d.reset_parameters
        self.weight = nn.Parameter(w.contiguous(memory_format=torch.channels_last_3d))
        if bias:
            bound = 1 / math.sqrt(in_channels * ks[0] * ks[1] * ks[2])
            self.bias = nn.Parameter(torch.empty(out_channels).uniform_(-bound, bound))
        else:
            self.register_parameter('bias', None)
        self.in_channels, self.out_channels, self.kernel_size = in_channels, out_channels, ks
        self.spec = spec
        self.op = GF.ConvOp(spec)

    def forward(self, inp: Tensor, resid: Optional[Tensor] = None) -> Tensor:
        return GF.conv3d(inp, self.weight, self.bias, self.op, resid)

    def extra_repr(self) -> str:
        s = self.spec
        return f'{s.cin}, {s.cout}, kernel_size={s.kernel}, stride={s.stride}, pad_front={s.pad_front}, pad_back={s.pad_back}, shuffle={s.shuffle}'


class GroupedConv3d(nn.Module):
    """``nn.Conv3d(groups=G)`` (reference video.py:168-175 hands ``groups`` on to nn.Conv3d): output channels [g co, (g + 1) co) see input
    channels [g ci, (g + 1) ci) only.  One parameter pair laid out like nn.Conv3d's (``weight`` (Cout, Cin / G, kt, kh, kw), ``bias`` (Cout)),
    G launches of the dense conv on channel slices (a CL copy of the slice in, a channel concat out -- plumbing; no shipped blueprint
    uses groups, so there is no grouped kernel)."""

    def __init__(self, in_channels: int, out_channels: int, kernel_size, spec: ConvSpec, groups: int, bias: bool = True) -> None:
        super().__init__()
        if groups < 1 or in_channels % groups or out_channels % groups:
            raise ValueError(f'in_channels {in_channels} and out_channels {out_channels} must be divisible by groups {groups}')   # nn.Conv3d's rule
        if spec.shuffle is not None:
            raise NotImplementedError('GroupedConv3d: a depth-to-space store pattern together with groups is not implemented')
        ks = _triple(kernel_size)
        ci, co = in_channels // groups, out_channels // groups
        w = torch.empty(out_channels, ci, *ks)
        nn.init.kaiming_uniform_(w, a=math.sqrt(5))
        self.weight = nn.Parameter(w.contiguous(memory_format=torch.channels_last_3d))
        if bias:
            bound = 1 / math.sqrt(ci * ks[0] * ks[1] * ks[2])
            self.bias = nn.Parameter(torch.empty(out_channels).uniform_(-bound, bound))
        else:
            self.register_parameter('bias', None)
        self.in_channels, self.out_channels, self.kernel_size, self.groups = in_channels, out_channels, ks, groups
        self.spec = ConvSpec(ci, co, spec.kernel, spec.stride, spec.dilation, spec.pad_front, spec.pad_back, None)
        self.ops = [GF.ConvOp(self.spec) for _ in range(groups)]

    def forward(self, inp: Tensor, resid: Optional[Tensor] = None) -> Tensor:
        """`resid` (output-shaped, optional) is added to the result -- the same call signature as Conv3d.forward, so that callers which hand a
        residual to `.conv3d` (VideoResidualBlock's tail) work with either."""
        inp = to_cl(inp)
        ci, co = self.spec.cin, self.spec.cout
        outs = []
        for g, op in enumerate(self.ops):
            xg = inp[:, g * ci:(g + 1) * ci]              # ci % 8 == 0: a CL view (same pitch, 16-byte aligned); else a dense copy with zeroed pad channels
            if ci % 8:
                xg = xg.clone(memory_format=torch.contiguous_format)
            outs.append(GF.conv3d(xg, self.weight[g * co:(g + 1) * co], None if self.bias is None else self.bias[g * co:(g + 1) * co], op))
        out = to_cl(torch.cat(outs, dim=1))
        return out if resid is None else to_cl(out + to_cl(resid))

    def extra_repr(self) -> str:
        return f'{self.in_channels}, {self.out_channels}, kernel_size={self.kernel_size}, groups={self.groups}, stride={self.spec.stride}'


def get_blur_kernel(kernel_size, device=None, dtype=None, norm: bool = True) -> Tensor:
    """Pascal-triangle blur taps (reference video.py:22-56, including its use of kernel_size[0] for the
    h taps and for the length of the w taps)."""
    if isinstance(kernel_size, int):
        kernel_size = (kernel_size, kernel_size)
    k0, k1 = kernel_size[0], kernel_size[1]
    t = torch.tensor([math.comb(k0 - 1, i) for i in range(k0)], device=device, dtype=dtype)
    h = torch.tensor([math.comb(k0 - 1, i) for i in range(k0)], device=device, dtype=dtype)
    w = torch.tensor([math.comb(k1 - 1, i) for i in range(k0)], device=device, dtype=dtype)
    k = t[:, None, None] * h[None, :, None] * w[None, None, :]
    return k / k.sum() if norm else k


class Upsample(nn.Module, ABC):
    def __init__(self, time_factor: int = 1, space_factor: int = 1) -> None:
        super().__init__()
        self.time_factor, self.space_factor = time_factor, space_factor
        self.go_up = None

    @property
    def factor(self):
        return self.time_factor * (self.space_factor ** 2)

    def forward(self, inp: Tensor, **kwargs) -> Tensor:
        return self.go_up(inp)


class Downsample(nn.Module, ABC):
    def __init__(self, time_factor: int = 1, space_factor: int = 1) -> None:
        super().__init__()
        self.time_factor, self.space_factor = time_factor, space_factor
        self.go_down = None

    @property
    def factor(self):
        return self.time_factor * (self.space_factor ** 2)

    def forward(self, inp: Tensor, **kwargs) -> Tensor:
        return self.go_down(inp)


class CausalConv3d(nn.Module):
    """reference video.py:106-200.  The causal front padding is a predicate in the gather, never a copy."""

    def __init__(self, in_channels: int, out_channels: int, kernel_size, stride=(1, 1, 1), dilation=(1, 1, 1),
                 padding=None, pad_mode: str = 'constant', _shuffle=None, **kwargs) -> None:
        super().__init__()
        stride, dilation, kernel_size = _triple(stride), _triple(dilation), _triple(kernel_size)
        if isinstance(padding, int) or padding is None:
            padding = (padding, padding)
        if pad_mode not in ('constant', 'reflect', 'replicate', 'circular'):
            raise ValueError(f"CausalConv3d: unknown pad_mode '{pad_mode}'")
        bias = kwargs.pop('bias', True)
        groups = int(kwargs.pop('groups', 1))
        if kwargs:
            raise TypeError(f'CausalConv3d: unexpected arguments {sorted(kwargs)}')
        spec = causal_spec(in_channels, out_channels, kernel_size, stride, dilation, padding, shuffle=_shuffle)
        # zero padding ('constant', every shipped blueprint) is a predicate in the conv kernels' gather.  The other F.pad modes of the
        # reference (video.py:160-164) have no such form: the padded tensor is materialised once (torch's pad kernel on the bf16 tensor, a
        # copy of the activation -- plumbing, not arithmetic) and the conv then runs WITHOUT padding on it.
        self.pad_mode, self._pads = pad_mode, None
        if pad_mode != 'constant':
            if causal_time_crop(kernel_size, stride, dilation):
                raise NotImplementedError("CausalConv3d: a negative causal pad (cropping) together with a non-constant pad_mode is not implemented")
            self._pads = (spec.pad_front[2], spec.pad_back[2], spec.pad_front[1], spec.pad_back[1], spec.pad_front[0], spec.pad_back[0])
            spec = ConvSpec(in_channels, spec.cout, kernel_size, stride, dilation, (0, 0, 0), (0, 0, 0), _shuffle)
        self.conv3d = (Conv3d(in_channels, out_channels, kernel_size, spec, bias=bias) if groups == 1
                       else GroupedConv3d(in_channels, out_channels, kernel_size, spec, groups, bias=bias))
        self.in_channels, self.out_channels = in_channels, out_channels
        # a NEGATIVE causal pad (kt = 1, time stride 2: (kt - 1) dil + 1 - stride = -1) crops leading frames in the reference (F.pad with a
        # negative amount, video.py:154-164, 189)
        self.time_crop = causal_time_crop(kernel_size, stride, dilation)

    def forward(self, inp: Tensor, resid: Optional[Tensor] = None) -> Tensor:
        """`resid` (output-shaped, optional): added in the conv's epilogue.  Callers that want the fused add MUST come through here and not
        through `.conv3d`: the time crop and the non-zero pad modes live in this method (VideoResidualBlock called `.conv3d`
        directly and ran reflect / replicate / circular convs unpadded)."""
        if self.time_crop:
            if inp.shape[2] <= self.time_crop:
                raise ValueError(f'CausalConv3d: {inp.shape[2]} frames, the causal padding of this layer removes {self.time_crop}')
            inp = to_cl(inp)[:, :, self.time_crop:].contiguous(memory_format=torch.channels_last_3d)     # a dense CL copy of the kept frames
        if self._pads is not None:
            inp = torch.nn.functional.pad(to_cl(inp), self._pads, mode=self.pad_mode)
        return self.conv3d(inp) if resid is None else self.conv3d(inp, resid=resid)

    @property
    def plain(self) -> bool:
        """Is this layer exactly its inner dense Conv3d (zero padding as a gather predicate, no crop, no groups)?  Only then may a caller
        unwrap `.conv3d` into a fused node."""
        return self._pads is None and not self.time_crop and isinstance(self.conv3d, Conv3d)

    @property
    def inp_dim(self) -> int:
        return self.in_channels

    @property
    def out_dim(self) -> int:
        return self.out_channels


class _Shuffled(nn.Module):
    """Stand-in for the reference's einops ``Rearrange`` child (no parameters; keeps Sequential indices)."""

    def forward(self, x):
        return x


class DepthToSpaceTimeUpsample(Upsample):
    """reference video.py:379-430: CausalConv3d(C -> C' * tf * sf^2) then
    'b (c p q r) t h w -> b c (t p) (h q) (w r)'.  The rearrange is the store pattern of the GEMM epilogue."""

    def __init__(self, in_channels: int, out_channels: Optional[int] = None, time_factor: int = 2, space_factor: int = 2,
                 kernel_size=1) -> None:
        super().__init__(time_factor=time_factor, space_factor=space_factor)
        out_channels = default(out_channels, in_channels)
        self.go_up = nn.Sequential(
            CausalConv3d(in_channels, out_channels * time_factor * space_factor ** 2, kernel_size=kernel_size,
                         _shuffle=(time_factor, space_factor, space_factor)),
            _Shuffled(),
        )
        self.in_channels, self.out_channels = in_channels, out_channels

    def forward(self, inp: Tensor, **kwargs) -> Tensor:
        return self.go_up(inp)

    @property
    def inp_dim(self) -> int:
        return self.in_channels

    @property
    def out_dim(self) -> int:
        return self.out_channels


class SpaceTimeDownsample(Downsample):
    """reference video.py:457-483: strided CausalConv3d."""

    def __init__(self, in_channels: int, kernel_size, out_channels: Optional[int] = None, time_factor: int = 2,
                 space_factor: int = 2, **kwargs) -> None:
        super().__init__(time_factor=1 / time_factor, space_factor=1 / space_factor)
        self.go_down = CausalConv3d(in_channels, default(out_channels, in_channels), kernel_size=_triple(kernel_size),
                                    stride=(time_factor, space_factor, space_factor), **kwargs)


class BlurPooling3d(nn.Module):
    """reference video.py:487-537.  With num_groups=1 the reference's dense conv sums ALL input channels into
    every output channel; computed as what it is (channel sum -> strided Pascal stencil ->
    broadcast, ``genie_blur_pool3d_fwd``) instead of a dense GEMM."""

    def __init__(self, in_channels: int, kernel_size, out_channels: Optional[int] = None, time_factor: int = 2,
                 space_factor=2, num_groups: int = 1, **kwargs) -> None:
        super().__init__()
        ks = _triple(kernel_size)
        if isinstance(space_factor, int):
            space_factor = (space_factor, space_factor)
        self.register_buffer('blur', get_blur_kernel(ks))
        self.stride = (time_factor, *space_factor)
        self.kwargs, self.num_groups, self.out_channels = kwargs, num_groups, out_channels
        self.padding = tuple((k - 1) // 2 for k in ks)

    def forward(self, inp: Tensor) -> Tensor:
        inp = to_cl(inp)
        c, o, g = inp.shape[1], default(self.out_channels, inp.shape[1]), self.num_groups
        if g == 1:
            return GF.blur_pool3d(inp, self.blur, self.stride, self.padding, o)
        # num_groups = g (reference video.py:520-533: F.conv3d(..., groups=g) with the Pascal kernel repeated over (o, c / g)): every output
        # channel of group i is the strided blur of the SUM of group i's input channels -- the one-group kernels on channel-slice views
        # (a CL tensor with a pitch wider than its channel count), one launch pair per group, concatenated.  VideoResidualBlock hands its
        # GroupNorm's num_groups to this module (video.py:592-597), so `video-residual` blueprints with groups AND a downsample land here.
        if c % g or o % g:
            raise ValueError(f'BlurPooling3d: {c} -> {o} channels are not divisible into {g} groups (F.conv3d refuses this in the reference)')
        cg, og = c // g, o // g
        # groups of a multiple of 8 channels are views of the input (16-byte aligned slices); any other width is first copied into a tensor of its own
        # (`to_cl` of the slice: differentiable, one extra pass over 1 / g of the tensor per group -- round 6; rounds 1-5 raised)
        piece = (lambda i: inp[:, i * cg:(i + 1) * cg]) if cg % 8 == 0 else (lambda i: to_cl(inp[:, i * cg:(i + 1) * cg]))
        outs = [GF.blur_pool3d(piece(i), self.blur, self.stride, self.padding, og) for i in range(g)]
        return to_cl(torch.cat(outs, dim=1))

    def __repr__(self):
        return f'BlurPooling3d({self.out_channels}, kernel_size={tuple(self.blur.shape)}, stride={self.stride}, padding={self.padding})'


class VideoResidualBlock(nn.Module):
    """reference video.py:539-656: main = [GN, act, conv k, (down), GN, act, conv k] + res = [(down), conv 1].

    Fused here: GN + SiLU in one pass each; the final `main + res` add rides in the epilogue of the second
    main conv (the reference allocates a separate sum)."""

    def __init__(self, in_channels: int, out_channels: Optional[int] = None, kernel_size=3, num_groups: int = 1,
                 pad_mode: str = 'constant', downsample=None, use_causal: bool = False, use_norm: bool = True,
                 use_blur: bool = True, act_fn: str = 'swish') -> None:
        super().__init__()
        from .image import LeakyReLU
        from .norm import GELU, GroupNorm, SiLU
        if isinstance(downsample, int):
            downsample = (downsample, downsample)
        ks = _triple(kernel_size)
        # reference video.py:581-585.  SiLU and LeakyReLU ride in the GroupNorm pass (genie_groupnorm_fwd act 1 / 2); ReLU (a LeakyReLU of slope
        # 0) and GELU are one element-wise pass behind it
        acts = {'swish': (SiLU, 1), 'silu': (SiLU, 1), 'leaky': (LeakyReLU, 2), 'relu': (lambda: LeakyReLU(0.0), 0), 'gelu': (GELU, 0)}
        if act_fn not in acts:
            raise ValueError(f"VideoResidualBlock: unknown act_fn '{act_fn}' (relu, gelu, leaky, swish / silu)")
        Act, self.act_code = acts[act_fn]
        if exists(downsample) and not use_blur:
            # the reference raises TypeError here too (SpaceTimeDownsample gets an unexpected num_groups)
            raise TypeError("VideoResidualBlock: downsample with use_blur=False is unsupported (the reference raises as well)")
        out_channels = default(out_channels, in_channels)
        tf, sf = downsample if exists(downsample) else (None, None)

        def conv(ci, co, k):
            k = _triple(k)
            if use_causal:   # CausalConv3d reads padding[0], padding[1] as the (h, w) pads (video.py:157-158)
                pad = None if k == (1, 1, 1) else ((k[0] - 1) // 2, (k[1] - 1) // 2)
                return CausalConv3d(ci, co, k, padding=pad, pad_mode=pad_mode)
            return Conv3d(ci, co, k, same_spec(ci, co, k))

        def down(ch):
            return BlurPooling3d(ch, ks, time_factor=tf, space_factor=sf, num_groups=num_groups) if exists(downsample) else nn.Identity()

        norm = (lambda ch: GroupNorm(num_groups, ch)) if use_norm else (lambda ch: nn.Identity())
        self.res = nn.Sequential(down(in_channels), conv(in_channels, out_channels, 1))
        self.main = nn.Sequential(norm(in_channels), Act(), conv(in_channels, out_channels, ks), down(out_channels),
                                  norm(out_channels), Act(), conv(out_channels, out_channels, ks))
        self.inp_channels, self.out_channels = in_channels, out_channels
        self.use_norm, self.use_causal = use_norm, use_causal

    def _norm_act(self, x: Tensor, norm: nn.Module, act: nn.Module) -> Tensor:
        if not self.use_norm:
            return act(x)
        if self.act_code:
            return GF.group_norm(x, norm.num_groups, norm.weight, norm.bias, norm.eps, act=self.act_code)
        return act(GF.group_norm(x, norm.num_groups, norm.weight, norm.bias, norm.eps, act=0))

    def forward(self, inp: Tensor) -> Tensor:
        inp = to_cl(inp)
        convs = (self.main[2], self.main[6], self.res[1])
        # the fused node takes the inner dense convs: only when every CausalConv3d IS its inner conv (no F.pad mode, no crop, no groups)
        fusable = all(m.plain if isinstance(m, CausalConv3d) else isinstance(m, Conv3d) for m in convs)
        if fusable and self.use_norm and self.act_code == 1 and isinstance(self.res[0], nn.Identity) and isinstance(self.main[3], nn.Identity):
            unwrap = lambda m: m.conv3d if isinstance(m, CausalConv3d) else m
            out = GF.residual_block(inp, self.main[0], unwrap(self.main[2]), self.main[4], unwrap(self.main[6]), unwrap(self.res[1]))
            if out is not None:
                return out
        res = self.res[1](self.res[0](inp))
        h = self.main[2](self._norm_act(inp, self.main[0], self.main[1]))
        h = self.main[3](h)
        h = self._norm_act(h, self.main[4], self.main[5])
        return self.main[6](h, resid=res)                  # Conv3d / CausalConv3d / (inside it) GroupedConv3d all take `resid`

    @property
    def inp_dim(self) -> int:
        return self.inp_channels

    @property
    def out_dim(self) -> int:
        return self.out_channels


class DepthToSpaceUpsample(Upsample):
    """reference video.py:279-327: per-frame Conv2d 1x1 then 'b (c p q) h w -> b c (h p) (w q)'."""

    def __init__(self, in_channels: int, out_channels: Optional[int] = None, factor: int = 2) -> None:
        super().__init__(space_factor=factor)
        out_channels = default(out_channels, in_channels)
        spec = ConvSpec(in_channels, out_channels * factor ** 2, (1, 1, 1), shuffle=(1, factor, factor))
        conv = Conv3d(in_channels, out_channels * factor ** 2, 1, spec)
        conv.weight = nn.Parameter(conv.weight.detach().reshape(out_channels * factor ** 2, in_channels, 1, 1).clone())   # Conv2d key shape
        self.go_up = nn.Sequential(conv, _Shuffled())
        self.in_channels, self.out_channels = in_channels, out_channels

    def forward(self, inp: Tensor, **kwargs) -> Tensor:
        conv = self.go_up[0]
        return GF.conv3d(inp, conv.weight.unsqueeze(2), conv.bias, conv.op)

    @property
    def inp_dim(self) -> int:
        return self.in_channels

    @property
    def out_dim(self) -> int:
        return self.out_channels


class DepthToTimeUpsample(Upsample):
    """reference video.py:329-377: Conv1d 1 over time then 'b (c f) t -> b c (t f)'."""

    def __init__(self, in_channels: int, out_channels: Optional[int] = None, factor: int = 2) -> None:
        super().__init__(time_factor=factor)
        out_channels = default(out_channels, in_channels)
        spec = ConvSpec(in_channels, out_channels * factor, (1, 1, 1), shuffle=(factor, 1, 1))
        conv = Conv3d(in_channels, out_channels * factor, 1, spec)
        conv.weight = nn.Parameter(conv.weight.detach().reshape(out_channels * factor, in_channels, 1).clone())           # Conv1d key shape
        self.go_up = nn.Sequential(conv, _Shuffled())
        self.in_channels, self.out_channels = in_channels, out_channels

    def forward(self, inp: Tensor, **kwargs) -> Tensor:
        conv = self.go_up[0]
        return GF.conv3d(inp, conv.weight[..., None, None], conv.bias, conv.op)

    @property
    def inp_dim(self) -> int:
        return self.in_channels

    @property
    def out_dim(self) -> int:
        return self.out_channels


class CausalConvTranspose3d(nn.Module):
    """reference video.py:202-277: ``nn.ConvTranspose3d(padding=(0, kh // 2, kw // 2))`` whose output is cropped to (t T, h H, w W).
    Parameters are laid out like ``nn.ConvTranspose3d`` (keys ``weight`` (in, out, kt, kh, kw) / ``bias``), so reference checkpoints
    load.  A transposed convolution IS the backward-data pass of the convolution with the same weight tensor: forward runs the
    backward-data kernels (one launch per output parity class for strides > 1), backward the forward / weight-gradient kernels."""

    def __init__(self, in_channels: int, out_channels: int, kernel_size, stride=(1, 1, 1), dilation=(1, 1, 1), space_pad=None, **kwargs) -> None:
        super().__init__()
        ks, stride, dilation = _triple(kernel_size), _triple(stride), _triple(dilation)
        if isinstance(space_pad, int) or space_pad is None:
            space_pad = (space_pad, space_pad)
        bias = kwargs.pop('bias', True)
        if kwargs.pop('groups', 1) != 1 or kwargs.pop('output_padding', 0) not in (0, (0, 0, 0)):
            raise NotImplementedError('CausalConvTranspose3d: groups / output_padding are not implemented on the HIP path')
        if kwargs:
            raise TypeError(f'CausalConvTranspose3d: unexpected arguments {sorted(kwargs)}')
        hp = default(space_pad[0], ks[1] // 2)
        wp = default(space_pad[1], ks[2] // 2)
        w = torch.empty(in_channels, out_channels, *ks)
        nn.init.kaiming_uniform_(w, a=math.sqrt(5))                 # nn.ConvTranspose3d.reset_parameters
        self.weight = nn.Parameter(w)
        if bias:
            bound = 1 / math.sqrt(out_channels * ks[0] * ks[1] * ks[2])      # fan_in of a transposed conv counts weight.size(1)
            self.bias = nn.Parameter(torch.empty(out_channels).uniform_(-bound, bound))
        else:
            self.register_parameter('bias', None)
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride, self.dilation, self.padding = ks, stride, dilation, (0, hp, wp)
        # the convolution this is the backward-data pass of: (out_channels -> in_channels), same taps / stride / dilation / padding
        self.spec = ConvSpec(out_channels, in_channels, ks, stride, dilation, self.padding, self.padding, None)
        self.op = GF.ConvOp(self.spec)

    def forward(self, inp: Tensor) -> Tensor:
        t, h, w = inp.shape[2:]
        full = tuple((n - 1) * s - 2 * p + d * (k - 1) + 1 for n, s, p, d, k in zip((t, h, w), self.stride, self.padding, self.dilation, self.kernel_size))
        out = GF.conv_transpose3d(inp, self.weight, self.op, full)
        out = out[:, :, :t * self.stride[0], :h * self.stride[1], :w * self.stride[2]]            # video.py:263-267
        if self.bias is not None:                    # in fp32: one rounding of the sum, and the bias gradient is an fp32 reduction
            out = out.float() + self.bias[None, :, None, None, None]
        return to_cl(out)

    @property
    def inp_dim(self) -> int:
        return self.in_channels

    @property
    def out_dim(self) -> int:
        return self.out_channels


class _ConvTransposeWeights(nn.Module):
    """Parameter holder with ``nn.ConvTranspose3d``'s keys (``weight`` (in, out, kt, kh, kw), ``bias`` (out))."""

    def __init__(self, in_dim: int, out_dim: int, ks, bias: bool = True) -> None:
        super().__init__()
        w = torch.empty(in_dim, out_dim, *ks)
        nn.init.kaiming_uniform_(w, a=math.sqrt(5))
        self.weight = nn.Parameter(w)
        if bias:
            bound = 1 / math.sqrt(out_dim * ks[0] * ks[1] * ks[2])
            self.bias = nn.Parameter(torch.empty(out_dim).uniform_(-bound, bound))
        else:
            self.register_parameter('bias', None)


class SpaceTimeUpsample(Upsample):
    """reference video.py:432-455: ``nn.ConvTranspose3d(kernel = stride = (tf, sf, sf))``.  With kernel == stride every input pixel writes
    its own (tf, sf, sf) block of the output -- a 1x1x1 convolution to Cout * tf * sf * sf channels followed by the depth-to-space-time
    rearrange '(c p q r)': the shuffle-epilogue conv kernels with the weight read as (out, p, q, r | in).  Keys ``go_up.weight`` /
    ``go_up.bias`` as in the reference."""

    def __init__(self, in_dim: int, out_dim: int, time_factor: int = 2, space_factor: int = 2, **kwargs) -> None:
        super().__init__(time_factor=time_factor, space_factor=space_factor)
        bias = kwargs.pop('bias', True)
        if kwargs:
            raise NotImplementedError(f'SpaceTimeUpsample: ConvTranspose3d options {sorted(kwargs)} are not implemented on the HIP path')
        f = (time_factor, space_factor, space_factor)
        self.go_up = _ConvTransposeWeights(in_dim, out_dim, f, bias)
        self.in_channels, self.out_channels, self.fac = in_dim, out_dim, f
        self.op = GF.ConvOp(ConvSpec(in_dim, out_dim * f[0] * f[1] * f[2], (1, 1, 1), shuffle=f))

    def forward(self, inp: Tensor, **kwargs) -> Tensor:
        w = self.go_up.weight                                               # (in, out, p, q, r)
        # the conv kernels see a DERIVED copy of the parameter (fresh tensor every call: version 0, and the caching allocator tends to
        # hand back the same address), so ConvOp's (version, address) key cannot notice an optimiser step / load_state_dict
        # (ADVICE r3): key the packs on the PARAMETER and drop them whenever it has changed
        pkey = (w._version, w.data_ptr())
        if getattr(self, '_pack_key', None) != pkey:
            self.op._fwd = self.op._bwd = (None, None)
            self._pack_key = pkey
        rows = self.out_channels * self.fac[0] * self.fac[1] * self.fac[2]
        w_conv = w.permute(1, 2, 3, 4, 0).reshape(rows, self.in_channels)[:, :, None, None, None]      # row (c p q r), column in
        b = self.go_up.bias
        b_conv = None if b is None else b.repeat_interleave(rows // self.out_channels)
        return GF.conv3d(inp, w_conv, b_conv, self.op)

    @property
    def inp_dim(self) -> int:
        return self.in_channels

    @property
    def out_dim(self) -> int:
        return self.out_channels
