"""The internal activation layout ("CL") and its conversion at the model boundary.

A CL tensor is a bf16 tensor of LOGICAL shape (N, C, T, H, W) -- the shape every reference module
takes (reference genie/module/video.py:178) -- whose memory is channels-last with a channel pitch Cp
that is a multiple of 8: physical (N, T, H, W, Cp), pad channels [C, Cp) kept at zero.  For C % 8 == 0
this is exactly torch's ``channels_last_3d`` memory format, so drop-in users can hand one in directly.
"""
from __future__ import annotations

import torch
from torch import Tensor

from . import _hip


def cpitch(c: int) -> int:
    return (c + 7) & ~7


def is_cl(x: Tensor) -> bool:
    if x.dtype != torch.bfloat16 or x.dim() != 5 or not x.is_cuda:
        return False
    n, c, t, h, w = x.shape
    sn, sc, st, sh, sw = x.stride()
    if c > 1 and sc != 1:
        return False
    cp = sw
    if cp % 8 != 0 or cp < c:
        return False
    if cp != c and x.storage_offset() % 8 != 0:
        return False
    return sh == w * cp and st == h * w * cp and sn == t * h * w * cp and x.data_ptr() % 16 == 0


def pitch_of(x: Tensor) -> int:
    return x.stride(4)


def empty_cl(n: int, c: int, t: int, h: int, w: int, device, zero_pad: bool = True) -> Tensor:
    """Uninitialised CL tensor; pad channels (if any) are zeroed so the invariant holds once the
    producer kernel has written channels [0, C)."""
    cp = cpitch(c)
    buf = torch.empty((n, t, h, w, cp), dtype=torch.bfloat16, device=device)
    if cp != c and zero_pad:
        buf[..., c:].zero_()
    return buf[..., :c].permute(0, 4, 1, 2, 3)


def empty_like_cl(x: Tensor) -> Tensor:
    n, c, t, h, w = x.shape
    return empty_cl(n, c, t, h, w, x.device)


class _ToClFn(torch.autograd.Function):
    """Differentiable layout conversion: the gradient comes back as a dense tensor of the input's dtype."""

    @staticmethod
    def forward(ctx, x: Tensor) -> Tensor:
        ctx.dtype = x.dtype
        return _to_cl_impl(x)

    @staticmethod
    def backward(ctx, g: Tensor):
        g = _to_cl_impl(g)
        return from_cl(g, dtype=ctx.dtype if ctx.dtype in (torch.float32, torch.bfloat16) else torch.float32).to(ctx.dtype)


def to_cl(x: Tensor) -> Tensor:
    """Any (N, C, T, H, W) fp32/bf16 CUDA tensor -> CL (no-op if it already is)."""
    if is_cl(x):
        return x
    if x.requires_grad and torch.is_grad_enabled():
        return _ToClFn.apply(x)
    return _to_cl_impl(x)


def _to_cl_impl(x: Tensor) -> Tensor:
    if is_cl(x):
        return x
    if x.dim() != 5:
        raise ValueError(f'expected a 5-D (N, C, T, H, W) tensor, got shape {tuple(x.shape)}')
    _hip.require_gpu(x, 'to_cl')
    if x.dtype == torch.float32:
        dt = _hip.GENIE_F32
    elif x.dtype == torch.bfloat16:
        dt = _hip.GENIE_BF16
    else:
        x, dt = x.float(), _hip.GENIE_F32
    n, c, t, h, w = x.shape
    out = empty_cl(n, c, t, h, w, x.device, zero_pad=False)   # the kernel writes the pad channels itself
    lib = _hip.load_library()
    _hip.check(lib.genie_to_channels_last(x.data_ptr(), dt, _hip.i64(x.shape), _hip.i64(x.stride()), out.data_ptr(),
                                          pitch_of(out), _hip.stream_ptr()), 'genie_to_channels_last')
    return out


def from_cl(x: Tensor, dtype=torch.float32) -> Tensor:
    """CL -> contiguous (N, C, T, H, W) tensor of `dtype` (fp32 or bf16)."""
    assert is_cl(x)
    out = torch.empty(x.shape, dtype=dtype, device=x.device)
    dt = _hip.GENIE_F32 if dtype == torch.float32 else _hip.GENIE_BF16
    lib = _hip.load_library()
    _hip.check(lib.genie_from_channels_last(x.data_ptr(), pitch_of(x), _hip.i64(x.shape), out.data_ptr(), dt,
                                            _hip.i64(out.stride()), _hip.stream_ptr()), 'genie_from_channels_last')
    return out
