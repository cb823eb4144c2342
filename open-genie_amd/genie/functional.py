"""Autograd bindings of the HIP kernels (the only place torch.autograd meets libgenie_hip.so).

Every Function takes/returns CL tensors (genie/cl.py) and enqueues kernels on the current stream;
nothing here synchronises, so a whole training step can be captured in one hipGraph.

Weight gradients, two modes (``DIRECT_PARAM_GRADS``, env ``GENIE_DIRECT_PARAM_GRADS``):

* ``'arena'`` (default): parameters that a ``genie.trainer.ParamArena`` manages get their gradients accumulated straight into
  ``param.grad`` (a view of the flat gradient arena) by the wgrad kernels, with fp32 atomics, and the Function returns ``None`` for
  them -- no per-step allocation / memset / add per weight, one fused optimiser kernel, contiguous all-reduce ranges.  Every other
  parameter gets the classic behaviour: the gradient is RETURNED to autograd, so ``torch.autograd.grad`` w.r.t. parameters,
  parameter hooks, torch DDP's reducer and Lightning's ``strategy=ddp`` (the reference's entry point, config/tokenize.yaml:77)
  see it like any other gradient.
* ``'all'``: direct accumulation for every fp32 leaf parameter (what round 1 did unconditionally; bypasses AccumulateGrad).
* ``'off'``: never.
"""
from __future__ import annotations

from typing import Optional

import torch
from torch import Tensor

from . import _hip
from .cl import empty_like_cl, is_cl, pitch_of, to_cl
from . import conv as _conv
from .conv import ConvSpec, conv_dgrad, conv_forward, conv_wgrad, pack_weight_bwd, pack_weight_fwd

DIRECT_PARAM_GRADS = __import__('os').environ.get('GENIE_DIRECT_PARAM_GRADS', 'arena')


def _direct(p: Optional[Tensor]) -> bool:
    """May the kernels accumulate this parameter's gradient straight into ``p.grad``?"""
    mode = DIRECT_PARAM_GRADS
    if mode is True or mode == 'all':
        return True
    if mode is False or mode == 'off' or p is None:
        return False
    return getattr(p, '_genie_arena', False)

FUSED_RESBLOCK = __import__('os').environ.get('GENIE_FUSED_RESBLOCK', '1') != '0'    # VideoResidualBlock as one autograd node

# Weight-gradient kernels on a side stream (GENIE_ASYNC_WGRAD=1 or functional.ASYNC_WGRAD = True; arena-managed parameters only).
# A wgrad launch only feeds the optimiser: nothing in the rest of backward waits for it.  Issued on its own stream it runs
# CONCURRENTLY with whatever the main stream does next -- in particular the HBM-bound GroupNorm backward passes and shortcut
# convolutions, which leave the matrix pipes idle (12 + ~5 ms of an 87 ms step) while the MFMA-bound wgrad kernels (23 ms) leave
# HBM idle.  Contract: whoever consumes the gradients joins first -- ``join_wgrad()``; ParamArena.adamw_step / zero_grad and
# DataParallel do.  Off by default because a plain ``loss.backward(); torch_optimizer.step()`` would race.
# Mode 1 (what bench.py uses): every conv forward / backward-data launch on the main stream first waits for the side stream, so the
# MFMA-bound gather-GEMMs never share the chip with a wgrad kernel -- the wgrad kernels overlap the NON-conv work that follows them
# (GroupNorm backward, LFQ, element-wise passes) and nothing else; per-kernel timings of the gather-GEMMs stay clean.  Mode 2: no
# such waits (wgrad runs whenever the hardware finds room; measured equally fast end to end, but it inflates the wall time of the
# conv kernels it shares CUs with).
ASYNC_WGRAD = int(__import__('os').environ.get('GENIE_ASYNC_WGRAD', '0'))
_wgrad_streams = {}
_wgrad_pending = set()


def wgrad_stream(device=None):
    idx = torch.cuda.current_device() if device is None or device.index is None else device.index
    st = _wgrad_streams.get(idx)
    if st is None:
        st = _wgrad_streams[idx] = torch.cuda.Stream(device=idx)
    return st


def _wgrad(x: Tensor, dy: Tensor, spec, gw: Tensor, gb: Optional[Tensor], dy_unshuffled: bool = False) -> None:
    """conv_wgrad into arena-managed gradients, on the side stream when ASYNC_WGRAD is on."""
    if not ASYNC_WGRAD:
        conv_wgrad(x, dy, spec, gw, gb, dy_unshuffled)
        return
    side = wgrad_stream(x.device)
    side.wait_stream(torch.cuda.current_stream())              # x and dy are complete
    with torch.cuda.stream(side):
        conv_wgrad(x, dy, spec, gw, gb, dy_unshuffled)
    x.record_stream(side)                                      # the caching allocator must not hand these out again before the kernel ran
    dy.record_stream(side)
    _wgrad_pending.add(side.device.index)


_LFQ_DEFER = False


class deferred_lfq_loss:
    """Context: LFQ training losses computed inside it are issued on the side stream (when ASYNC_WGRAD == 2) -- the caller must call
    ``join_wgrad()`` before reading the loss or running backward."""

    def __enter__(self):
        global _LFQ_DEFER
        self._old = _LFQ_DEFER
        _LFQ_DEFER = ASYNC_WGRAD == 2
        return self

    def __exit__(self, *exc):
        global _LFQ_DEFER
        _LFQ_DEFER = self._old
        return False


def _conv_gate() -> None:
    """Called before a conv forward / backward-data launch: in mode 1 the main stream waits for outstanding wgrad kernels."""
    if ASYNC_WGRAD == 1 and _wgrad_pending:
        join_wgrad()


def join_wgrad(stream=None) -> None:
    """Make `stream` (default: the current stream) wait for every weight-gradient kernel issued on a side stream so far."""
    if not _wgrad_pending:
        return
    for idx in list(_wgrad_pending):
        (stream if stream is not None else torch.cuda.current_stream(idx)).wait_stream(_wgrad_streams[idx])
    if stream is None:
        _wgrad_pending.clear()


_ws_cache = {}


def workspace(nfloats: int, device, tag: str = 'ws') -> Tensor:
    """Grow-only fp32 scratch per (device, tag, current stream): reuse is stream-ordered, so every stream that launches kernels
    (the weight-gradient / LFQ side stream, a data-parallel comm stream) gets its own buffer."""
    key = (device.index if device.index is not None else torch.cuda.current_device(), tag, torch.cuda.current_stream(device).cuda_stream)
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nfloats:
        buf = torch.empty(max(int(nfloats), 1 << 16), dtype=torch.float32, device=device)
        _ws_cache[key] = buf
    return buf


def _grad_buffer(p: Tensor) -> Tensor:
    if p.grad is None:
        p.grad = torch.zeros_like(p, memory_format=torch.preserve_format)
    return p.grad


# ------------------------------------------------------------------------------------------------
# Conv3d
# ------------------------------------------------------------------------------------------------
class ConvOp:
    """Geometry + bf16 weight packs of one convolution (packs refresh when the parameter changes)."""

    def __init__(self, spec: ConvSpec):
        self.spec = spec
        self._fwd = (None, None)
        self._bwd = (None, None)

    def pack_fwd(self, weight: Tensor) -> Tensor:
        key = (weight._version, weight.data_ptr())
        if self._fwd[0] != key:
            self._fwd = (key, pack_weight_fwd(weight, self.spec))
        return self._fwd[1]

    def pack_bwd(self, weight: Tensor) -> Tensor:
        key = (weight._version, weight.data_ptr())
        if self._bwd[0] != key:
            self._bwd = (key, pack_weight_bwd(weight, self.spec))
        return self._bwd[1]

    def pack_narrow(self, weight: Tensor, bias: Optional[Tensor], backward: bool) -> Tensor:
        """Pack for the HBM-bound narrow kernel (conv_narrow.hip), rebuilt when the weight (or bias) changes."""
        key = (weight._version, weight.data_ptr(), None if bias is None or backward else (bias._version, bias.data_ptr()))
        slot = '_nbwd' if backward else '_nfwd'
        cur = getattr(self, slot, (None, None))
        if cur[0] != key:
            cur = (key, _conv.pack_narrow_bwd(weight) if backward else _conv.pack_narrow_fwd(weight, bias))
            setattr(self, slot, cur)
        return cur[1]

    def pack_narrow_out(self, weight: Tensor) -> Tensor:
        key = (weight._version, weight.data_ptr())
        cur = getattr(self, '_nout', (None, None))
        if cur[0] != key:
            cur = (key, _conv.pack_narrow_out(weight))
            self._nout = cur
        return cur[1]


class _Conv3dFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x: Tensor, weight: Tensor, bias: Optional[Tensor], op: ConvOp, resid: Optional[Tensor]):
        _conv_gate()
        if resid is None and _conv.narrow_fwd_ok(op.spec, x):
            out = _conv.conv_narrow_in(x, op.pack_narrow(weight, bias, False), -op.spec.pad_front[0], f'fwd {op.spec.cin}->128 k3 @{tuple(x.shape[2:])}')
        elif resid is None and _conv.narrow_out_ok(op.spec, x):
            out = _conv.conv_narrow_out(x, op.pack_narrow_out(weight), bias, op.spec.cout, -op.spec.pad_front[0], f'fwd 128->{op.spec.cout} k3 @{tuple(x.shape[2:])}')
        else:
            out = conv_forward(x, op.pack_fwd(weight), bias, op.spec, resid=resid)
        ctx.op = op
        ctx.in_size = tuple(x.shape[2:])
        ctx.has_resid = resid is not None
        ctx.save_for_backward(x, weight, bias)
        return out

    @staticmethod
    def backward(ctx, dy: Tensor):
        x, weight, bias = ctx.saved_tensors
        op: ConvOp = ctx.op
        dy = to_cl(dy)
        dx = dw = db = None
        # upsample conv: ONE un-shuffle of the output gradient serves the backward-data pass and the weight gradient (both then run the
        # plain kw-triple kernels on the conv's own row grid instead of gathering through the depth-to-space shuffle)
        dyu = _conv.unshuffle_dy(dy, op.spec) if _conv.wgrad_unshuffled_ok(op.spec, x) else None
        if ctx.needs_input_grad[0]:
            _conv_gate()
            if _conv.narrow_dgrad_ok(op.spec, dy):
                dx = _conv.conv_narrow_in(dy, op.pack_narrow(weight, None, True), op.spec.pad_front[0] - 2, f'dgrad 128->{op.spec.cout} k3 @{tuple(dy.shape[2:])}')
            else:
                dx = conv_dgrad(dy, op.pack_bwd(weight), op.spec, ctx.in_size, dy_unshuffled=dyu)
        need_w = ctx.needs_input_grad[1]
        need_b = bias is not None and ctx.needs_input_grad[2]
        if need_w or need_b:
            # a (V, D, 1, 1, 1) view of an nn.Linear weight (the Dynamics vocabulary head) accumulates into the Linear's gradient
            wleaf = weight if weight.is_leaf else getattr(weight, '_base', None)
            linear_view = (wleaf is not weight and wleaf is not None and wleaf.dim() == 2 and wleaf.is_contiguous()
                           and tuple(weight.shape) == (*wleaf.shape, 1, 1, 1))
            if wleaf is not None and _direct(wleaf) and wleaf.is_leaf and (wleaf is weight or linear_view) and (bias is None or (bias.is_leaf and _direct(bias))):
                gw = _grad_buffer(wleaf).view(weight.shape) if wleaf is not weight else _grad_buffer(weight)
                gb = _grad_buffer(bias) if need_b else None
                (_wgrad if getattr(wleaf, '_genie_arena', False) else conv_wgrad)(x, dy if dyu is None else dyu, op.spec, gw, gb, dyu is not None)
            else:
                dw = torch.zeros(weight.shape, dtype=torch.float32, device=weight.device)
                db = torch.zeros_like(bias) if need_b else None
                conv_wgrad(x, dy if dyu is None else dyu, op.spec, dw, db, dyu is not None)
        dres = dy if ctx.has_resid and ctx.needs_input_grad[4] else None
        return dx, dw, db, None, dres


def conv3d(x: Tensor, weight: Tensor, bias: Optional[Tensor], op: ConvOp, resid: Optional[Tensor] = None) -> Tensor:
    """x: any (N, C, T, H, W) CUDA tensor (converted to CL once); returns a CL tensor.
    `resid` (CL, output-shaped) is added in the GEMM epilogue."""
    return _Conv3dFn.apply(to_cl(x), weight, bias, op, None if resid is None else to_cl(resid))


class _ConvTranspose3dFn(torch.autograd.Function):
    """ConvTranspose3d as what it is -- the backward-data pass of the convolution with the same weight tensor: forward =
    ``conv_dgrad``, input gradient = ``conv_forward``, weight gradient = ``conv_wgrad`` with the roles of the two activations swapped.
    `spec` describes that convolution (cin = the transposed conv's OUTPUT channels, cout = its input channels); `full` is the
    un-cropped output size."""

    @staticmethod
    def forward(ctx, x: Tensor, weight: Tensor, op: ConvOp, full):
        _conv_gate()
        y = conv_dgrad(x, op.pack_bwd(weight), op.spec, tuple(full))
        ctx.op = op
        ctx.save_for_backward(x, weight, y.new_empty(0))
        ctx.full = tuple(full)
        return y

    @staticmethod
    def backward(ctx, dy: Tensor):
        x, weight, _ = ctx.saved_tensors
        op: ConvOp = ctx.op
        dy = to_cl(dy)
        dx = dw = None
        if ctx.needs_input_grad[0]:
            _conv_gate()
            dx = conv_forward(dy, op.pack_fwd(weight), None, op.spec)
        if ctx.needs_input_grad[1]:
            dw = torch.zeros(weight.shape, dtype=torch.float32, device=weight.device)
            conv_wgrad(dy, x, op.spec, dw, None)            # conv input = the transposed conv's output gradient, conv output gradient = its input
        return dx, dw, None, None


def conv_transpose3d(x: Tensor, weight: Tensor, op: ConvOp, full) -> Tensor:
    """x: (N, Cin, T, H, W); weight: nn.ConvTranspose3d layout (Cin, Cout, kt, kh, kw); returns the UN-CROPPED (N, Cout, *full) CL tensor
    (no bias)."""
    return _ConvTranspose3dFn.apply(to_cl(x), weight, op, tuple(full))


# ------------------------------------------------------------------------------------------------
# GroupNorm (+ adaptive scale/shift) (+ SiLU)
# ------------------------------------------------------------------------------------------------
def _f32(t: Optional[Tensor]) -> Optional[Tensor]:
    if t is None:
        return None
    return t if (t.dtype == torch.float32 and t.is_contiguous()) else t.float().contiguous()


class _GroupNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x: Tensor, gamma, beta, ada_s, ada_b, groups: int, eps: float, act: int):
        n, c, t, h, w = x.shape
        lib = _hip.load_library()
        y = empty_like_cl(x)
        mean = torch.empty(n * groups, dtype=torch.float32, device=x.device)
        rstd = torch.empty_like(mean)
        ws = workspace(lib.genie_groupnorm_ws_floats(n, c, groups), x.device, 'gn')
        g_, b_, as_, ab_ = _f32(gamma), _f32(beta), _f32(ada_s), _f32(ada_b)
        P = _hip.ptr
        _hip.check(lib.genie_groupnorm_fwd(P(x), P(y), n, t * h * w, c, pitch_of(x), groups, P(g_), P(b_), P(as_), P(ab_),
                                           eps, act, P(mean), P(rstd), P(ws), _hip.stream_ptr()), 'genie_groupnorm_fwd')
        ctx.groups, ctx.act = groups, act
        ctx.save_for_backward(x, gamma, beta, ada_s, ada_b, mean, rstd)
        return y

    @staticmethod
    def backward(ctx, dy: Tensor):
        x, gamma, beta, ada_s, ada_b, mean, rstd = ctx.saved_tensors
        n, c, t, h, w = x.shape
        dy = to_cl(dy)
        if pitch_of(dy) != pitch_of(x):
            raise RuntimeError('group_norm backward: gradient pitch mismatch')
        lib = _hip.load_library()
        dx = empty_like_cl(x)
        ws = workspace(lib.genie_groupnorm_ws_floats(n, c, ctx.groups), x.device, 'gn')
        need_g = gamma is not None and ctx.needs_input_grad[1]
        need_b = beta is not None and ctx.needs_input_grad[2]
        dgamma = dbeta = None
        ret_g = ret_b = None
        if need_g:
            if _direct(gamma) and gamma.is_leaf and gamma.dtype == torch.float32 and gamma.is_contiguous():
                dgamma = _grad_buffer(gamma)
            else:
                dgamma = ret_g = torch.zeros(c, dtype=torch.float32, device=x.device)
        if need_b:
            if _direct(beta) and beta.is_leaf and beta.dtype == torch.float32 and beta.is_contiguous():
                dbeta = _grad_buffer(beta)
            else:
                dbeta = ret_b = torch.zeros(c, dtype=torch.float32, device=x.device)
        das = torch.empty(n, c, dtype=torch.float32, device=x.device) if ada_s is not None else None
        dab = torch.empty(n, c, dtype=torch.float32, device=x.device) if ada_b is not None else None
        P = _hip.ptr
        _hip.check(lib.genie_groupnorm_bwd(P(x), P(dy), P(dx), n, t * h * w, c, pitch_of(x), ctx.groups, P(_f32(gamma)), P(_f32(beta)),
                                           P(_f32(ada_s)), P(_f32(ada_b)), ctx.act, P(mean), P(rstd), P(dgamma), P(dbeta), P(das), P(dab),
                                           P(ws), _hip.stream_ptr()), 'genie_groupnorm_bwd')
        return dx, ret_g, ret_b, das, dab, None, None, None


def group_norm(x: Tensor, groups: int, gamma: Optional[Tensor], beta: Optional[Tensor], eps: float = 1e-5,
               ada_scale: Optional[Tensor] = None, ada_shift: Optional[Tensor] = None, act=False) -> Tensor:
    """`act`: False / 0 none, True / 1 SiLU, 2 LeakyReLU(0.01) -- applied in the same pass."""
    x = to_cl(x)
    if x.shape[1] % groups != 0:
        raise ValueError(f'num_channels {x.shape[1]} must be divisible by num_groups {groups}')
    return _GroupNormFn.apply(x, gamma, beta, ada_scale, ada_shift, groups, eps, int(act))


class _SiluFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x: Tensor):
        y = empty_like_cl(x)
        numel = x.shape[0] * x.shape[2] * x.shape[3] * x.shape[4] * pitch_of(x)
        _hip.check(_hip.load_library().genie_silu_fwd(x.data_ptr(), y.data_ptr(), numel, _hip.stream_ptr()), 'genie_silu_fwd')
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, dy: Tensor):
        (x,) = ctx.saved_tensors
        dy = to_cl(dy)
        dx = empty_like_cl(x)
        numel = x.shape[0] * x.shape[2] * x.shape[3] * x.shape[4] * pitch_of(x)
        _hip.check(_hip.load_library().genie_silu_bwd(x.data_ptr(), dy.data_ptr(), dx.data_ptr(), numel, _hip.stream_ptr()), 'genie_silu_bwd')
        return dx


def silu(x: Tensor) -> Tensor:
    return _SiluFn.apply(to_cl(x))


class _GeluFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x: Tensor):
        y = empty_like_cl(x)
        numel = x.shape[0] * x.shape[2] * x.shape[3] * x.shape[4] * pitch_of(x)
        _hip.check(_hip.load_library().genie_gelu_fwd(x.data_ptr(), y.data_ptr(), numel, _hip.stream_ptr()), 'genie_gelu_fwd')
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, dy: Tensor):
        (x,) = ctx.saved_tensors
        dy = to_cl(dy)
        dx = empty_like_cl(x)
        numel = x.shape[0] * x.shape[2] * x.shape[3] * x.shape[4] * pitch_of(x)
        _hip.check(_hip.load_library().genie_gelu_bwd(x.data_ptr(), dy.data_ptr(), dx.data_ptr(), numel, _hip.stream_ptr()), 'genie_gelu_bwd')
        return dx


def gelu(x: Tensor) -> Tensor:
    """nn.GELU() (exact erf form) on a CL tensor."""
    return _GeluFn.apply(to_cl(x))


class _LeakyFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x: Tensor, slope: float):
        y = empty_like_cl(x)
        numel = x.shape[0] * x.shape[2] * x.shape[3] * x.shape[4] * pitch_of(x)
        _hip.check(_hip.load_library().genie_leaky_relu_fwd(x.data_ptr(), y.data_ptr(), numel, slope, _hip.stream_ptr()), 'genie_leaky_relu_fwd')
        ctx.slope = slope
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, dy: Tensor):
        (x,) = ctx.saved_tensors
        dy = to_cl(dy)
        dx = empty_like_cl(x)
        numel = x.shape[0] * x.shape[2] * x.shape[3] * x.shape[4] * pitch_of(x)
        _hip.check(_hip.load_library().genie_leaky_relu_bwd(x.data_ptr(), dy.data_ptr(), dx.data_ptr(), numel, ctx.slope, _hip.stream_ptr()),
                   'genie_leaky_relu_bwd')
        return dx, None


def leaky_relu(x: Tensor, slope: float = 0.01) -> Tensor:
    return _LeakyFn.apply(to_cl(x), float(slope))


# ------------------------------------------------------------------------------------------------
# BlurPooling3d (num_groups = 1): channel sum -> strided Pascal stencil -> broadcast
# ------------------------------------------------------------------------------------------------
class _BlurPoolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x: Tensor, taps: Tensor, stride, pad, out_channels: int):
        from .cl import empty_cl
        n, c, t, h, w = x.shape
        k = tuple(taps.shape)
        to, ho, wo = ((sz + 2 * p - kk) // s + 1 for sz, p, kk, s in zip((t, h, w), pad, k, stride))
        out = empty_cl(n, out_channels, to, ho, wo, x.device)
        ws = workspace(max(n * t * h * w, n * to * ho * wo), x.device, 'blur')
        tp = taps.detach().float().contiguous()
        _hip.check(_hip.load_library().genie_blur_pool3d_fwd(x.data_ptr(), pitch_of(x), _hip.i64(x.shape), tp.data_ptr(), _hip.i32(k), _hip.i32(stride),
                                                             _hip.i32(pad), out.data_ptr(), out_channels, pitch_of(out), ws.data_ptr(),
                                                             _hip.stream_ptr()), 'genie_blur_pool3d_fwd')
        ctx.geom = (tuple(x.shape), k, tuple(stride), tuple(pad), out_channels)
        ctx.save_for_backward(tp)
        return out

    @staticmethod
    def backward(ctx, dy: Tensor):
        from .cl import empty_cl
        (tp,) = ctx.saved_tensors
        shape, k, stride, pad, oc = ctx.geom
        n, c, t, h, w = shape
        dy = to_cl(dy)
        dx = empty_cl(n, c, t, h, w, dy.device)
        ws = workspace(max(n * t * h * w, dy.shape[0] * dy.shape[2] * dy.shape[3] * dy.shape[4]), dy.device, 'blur')
        _hip.check(_hip.load_library().genie_blur_pool3d_bwd(dy.data_ptr(), oc, pitch_of(dy), _hip.i64(shape), tp.data_ptr(), _hip.i32(k), _hip.i32(stride),
                                                             _hip.i32(pad), dx.data_ptr(), pitch_of(dx), ws.data_ptr(), _hip.stream_ptr()),
                   'genie_blur_pool3d_bwd')
        return dx, None, None, None, None


def blur_pool3d(x: Tensor, taps: Tensor, stride, pad, out_channels: int) -> Tensor:
    """x: (N, C, T, H, W); taps: (kt, kh, kw) blur kernel; every one of the `out_channels` outputs = blur(sum_c x)."""
    return _BlurPoolFn.apply(to_cl(x), taps, tuple(int(s) for s in stride), tuple(int(p) for p in pad), int(out_channels))


# ------------------------------------------------------------------------------------------------
# MSE against an arbitrary-strided target
# ------------------------------------------------------------------------------------------------
class _MseFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rec: Tensor, target: Tensor):
        lib = _hip.load_library()
        loss = torch.empty((), dtype=torch.float32, device=rec.device)
        ws = workspace(1024, rec.device, 'mse')
        dt = _hip.GENIE_F32 if target.dtype == torch.float32 else _hip.GENIE_BF16
        _hip.check(lib.genie_mse_fwd(rec.data_ptr(), pitch_of(rec), target.data_ptr(), dt, _hip.i64(rec.shape), _hip.i64(target.stride()),
                                     ws.data_ptr(), loss.data_ptr(), _hip.stream_ptr()), 'genie_mse_fwd')
        ctx.save_for_backward(rec, target)
        return loss

    @staticmethod
    def backward(ctx, dloss: Tensor):
        rec, target = ctx.saved_tensors
        lib = _hip.load_library()
        drec = empty_like_cl(rec)
        g = dloss.float().contiguous()
        dt = _hip.GENIE_F32 if target.dtype == torch.float32 else _hip.GENIE_BF16
        _hip.check(lib.genie_mse_bwd(rec.data_ptr(), pitch_of(rec), target.data_ptr(), dt, _hip.i64(rec.shape), _hip.i64(target.stride()),
                                     g.data_ptr(), drec.data_ptr(), _hip.stream_ptr()), 'genie_mse_bwd')
        return drec, None


def mse_loss(rec: Tensor, target: Tensor) -> Tensor:
    """mean((rec - target)^2) as an fp32 scalar; `target` is read in place (fp32/bf16, any strides)."""
    if tuple(rec.shape) != tuple(target.shape):
        raise ValueError(f'mse_loss: shape mismatch {tuple(rec.shape)} vs {tuple(target.shape)}')
    if target.dtype not in (torch.float32, torch.bfloat16):
        target = target.float()
    _hip.require_gpu(target, 'mse_loss')
    return _MseFn.apply(to_cl(rec), target.detach())


# ------------------------------------------------------------------------------------------------
# Lookup-free quantisation on (ntok, pitch) rows
# ------------------------------------------------------------------------------------------------
class _LfqFn(torch.autograd.Function):
    """z2d: (ntok, pitch) bf16/fp32 rows (a view of the latent); returns (quant rows, idx, loss4)."""

    @staticmethod
    def forward(ctx, z2d: Tensor, width: int, ncb: int, d: int, training: bool, beta: float, commit_w: float, ent_w: float, div_w: float):
        lib = _hip.load_library()
        ntok, pitch = z2d.shape[0], z2d.stride(0)
        dt = _hip.GENIE_F32 if z2d.dtype == torch.float32 else _hip.GENIE_BF16
        quant = torch.zeros((ntok, pitch), dtype=z2d.dtype, device=z2d.device)[:, :z2d.shape[1]] if pitch != width else torch.empty_like(z2d)
        idx = torch.empty((ntok, ncb), dtype=torch.int64, device=z2d.device)
        _hip.check(lib.genie_lfq_quantize(z2d.data_ptr(), dt, ntok, ncb, d, pitch, quant.data_ptr(), idx.data_ptr(), _hip.stream_ptr()), 'genie_lfq_quantize')
        loss4 = None
        if training:
            loss4 = torch.empty(4, dtype=torch.float32, device=z2d.device)
            dzl = torch.empty((ntok, ncb * d), dtype=torch.float32, device=z2d.device)
            ws = workspace(lib.genie_lfq_loss_ws_floats(ntok, ncb, d), z2d.device, 'lfq')
            if _LFQ_DEFER:
                # inside `deferred_lfq_loss()`: the loss kernels (VALU-bound: 2^18 codes per token enumerated on chip, 5 ms at 8192 tokens)
                # go to the side stream and run under whatever the caller launches next (the decoder); the caller joins before it
                # touches the loss (VideoTokenizer.forward does)
                side = wgrad_stream(z2d.device)
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    _hip.check(lib.genie_lfq_loss(z2d.data_ptr(), dt, ntok, ncb, d, pitch, beta, commit_w, ent_w, div_w, ws.data_ptr(), loss4.data_ptr(),
                                                  dzl.data_ptr(), _hip.stream_ptr()), 'genie_lfq_loss')
                for t_ in (z2d, loss4, dzl):
                    t_.record_stream(side)
                _wgrad_pending.add(side.device.index)
            else:
                _hip.check(lib.genie_lfq_loss(z2d.data_ptr(), dt, ntok, ncb, d, pitch, beta, commit_w, ent_w, div_w, ws.data_ptr(), loss4.data_ptr(),
                                              dzl.data_ptr(), _hip.stream_ptr()), 'genie_lfq_loss')
            ctx.save_for_backward(dzl)
        ctx.meta = (dt, ntok, width, pitch, z2d.dtype, z2d.shape[1])
        ctx.training = bool(training)
        ctx.mark_non_differentiable(idx)
        return quant, idx, loss4

    @staticmethod
    def backward(ctx, dquant, _didx, dloss4):
        dt, ntok, width, pitch, dtype, ncol = ctx.meta
        lib = _hip.load_library()
        dzl = ctx.saved_tensors[0] if ctx.saved_tensors else None
        out = torch.empty((ntok, pitch), dtype=dtype, device=dquant.device if dquant is not None else dloss4.device)
        dq = None
        # the straight-through term exists only in training mode: in eval the reference returns `quant` itself
        # (quantization.py:104-113, no `inp + (quant - inp).detach()`), so nothing flows back to the encoder
        if dquant is not None and ctx.training:
            dq = dquant
            if dq.dtype != dtype or dq.stride(0) != pitch or dq.stride(1) != 1:
                tmp = torch.zeros((ntok, pitch), dtype=dtype, device=dq.device)
                tmp[:, :ncol] = dq
                dq = tmp
        gl = None
        if dloss4 is not None and dzl is not None:
            gl = dloss4[0:1].float().contiguous()      # only the total carries gradient
        _hip.check(lib.genie_lfq_bwd(_hip.ptr(dq), _hip.ptr(dzl) if gl is not None else None, _hip.ptr(gl), out.data_ptr(), dt, ntok, width, pitch,
                                     _hip.stream_ptr()), 'genie_lfq_bwd')
        return out[:, :ncol], None, None, None, None, None, None, None, None


def lfq_rows(z2d: Tensor, ncb: int, d: int, training: bool, beta: float, commit_w: float, ent_w: float, div_w: float):
    return _LfqFn.apply(z2d, ncb * d, ncb, d, training, beta, commit_w, ent_w, div_w)


# ------------------------------------------------------------------------------------------------
# Masked token cross-entropy over bf16 logits (DynamicsModel.compute_loss)
# ------------------------------------------------------------------------------------------------
class _MaskedCEFn(torch.autograd.Function):
    """logits: (..., V) bf16 whose rows are evenly pitched (the (B, T, H, W, V) permutation of a CL tensor, or any dense tensor);
    target / mask: (...).  Any V >= 1: rows are addressed with the channel pitch (a multiple of 8), pad columns are masked."""

    @staticmethod
    def forward(ctx, logits: Tensor, target: Tensor, mask: Optional[Tensor]):
        lib = _hip.load_library()
        v = logits.shape[-1]
        rows = logits.numel() // v
        vp = (v + 7) & ~7
        lg = None
        if logits.dtype == torch.bfloat16 and (v == 1 or logits.stride(-1) == 1):
            # rows evenly pitched? (contiguous over the leading dims with a last-dim pitch >= vp)
            pitch = logits.stride(-2) if logits.dim() >= 2 else vp
            expect, ok = pitch, pitch >= vp and pitch % 8 == 0 and logits.storage_offset() % 8 == 0
            for sz, st in zip(reversed(logits.shape[:-1]), reversed(logits.stride()[:-1])):
                ok = ok and (sz == 1 or st == expect)
                expect *= sz
            if ok:
                lg = logits.as_strided((rows, v), (pitch, 1))
        if lg is None:                                     # dense copy into a padded buffer (pad columns are never read as logits)
            buf = torch.zeros((rows, vp), dtype=torch.bfloat16, device=logits.device)
            buf[:, :v] = logits.reshape(rows, v)
            lg = buf[:, :v]
        tgt = target.reshape(rows).to(torch.int64).contiguous()
        mk = None if mask is None else mask.reshape(rows).to(torch.uint8).contiguous()
        lse = torch.empty(rows, dtype=torch.float32, device=logits.device)
        acc = torch.zeros(1, dtype=torch.float32, device=logits.device)
        _hip.check(lib.genie_masked_ce_fwd(lg.data_ptr(), lg.stride(0), rows, v, tgt.data_ptr(), _hip.ptr(mk), lse.data_ptr(), acc.data_ptr(),
                                           _hip.stream_ptr()), 'genie_masked_ce_fwd')
        count = (mk.sum(dtype=torch.float32) if mk is not None else torch.full((), float(rows), device=logits.device)).reshape(1)
        ctx.save_for_backward(lg, tgt, mk, lse, count)
        ctx.shape = tuple(logits.shape)
        return (acc / count).reshape(())

    @staticmethod
    def backward(ctx, g: Tensor):
        lg, tgt, mk, lse, count = ctx.saved_tensors
        rows, v = lg.shape
        vp = (v + 7) & ~7
        scale = (g.float().reshape(1) / count).contiguous()
        dl = torch.empty((rows, vp), dtype=torch.bfloat16, device=lg.device)          # the kernel zeroes the pad columns
        _hip.check(_hip.load_library().genie_masked_ce_bwd(lg.data_ptr(), lg.stride(0), rows, v, tgt.data_ptr(), _hip.ptr(mk), lse.data_ptr(),
                                                           scale.data_ptr(), dl.data_ptr(), vp, _hip.stream_ptr()), 'genie_masked_ce_bwd')
        return dl[:, :v].view(*ctx.shape[:-1], v), None, None


def masked_cross_entropy(logits: Tensor, target: Tensor, mask: Optional[Tensor] = None) -> Tensor:
    """mean over the rows where `mask` is set of -log softmax(logits)[target]  (= F.cross_entropy(logits[mask], target[mask]))."""
    _hip.require_gpu(logits, 'masked_cross_entropy')
    return _MaskedCEFn.apply(logits, target, mask)


# ------------------------------------------------------------------------------------------------
# Fused vocabulary head + masked cross-entropy (DynamicsModel.compute_loss): the logits never reach HBM
# ------------------------------------------------------------------------------------------------
FUSED_LINEAR_CE = __import__('os').environ.get('GENIE_FUSED_LINEAR_CE', '1') != '0'


def linear_ce_supported(h: Tensor, weight: Tensor) -> bool:
    """May ``linear_cross_entropy`` take (h: (M, D) bf16 rows, weight: (V, D))?  (D in {64, 128, 256, 512}; sizes below 2^31 bytes)
    Not in deterministic mode (conv.set_deterministic / GENIE_DETERMINISTIC=1): the fused kernels sum the loss, the split-vocabulary dW
    partials and the one-hot rows with fp32 atomics; the gather-GEMM + masked_ce path (single-owner weight gradient) is the reproducible one."""
    if not FUSED_LINEAR_CE or _conv.DETERMINISTIC or h.dim() != 2 or h.dtype != torch.bfloat16 or h.stride(1) != 1 or h.shape[0] < 1:
        return False
    return bool(_hip.load_library().genie_linear_ce_supported(h.shape[0], h.shape[1], weight.shape[0], h.stride(0), weight.shape[1]))


class _LinearCEFn(torch.autograd.Function):
    """loss = mean over the rows with valid != 0 of -log softmax(h W^T + b)[target]  -- reference genie/dynamics.py:62 (`self.head`) + :89-97
    (F.cross_entropy over the gathered rows) as ONE operator (genie_linear_ce_fwd / _bwd, csrc/linear_ce.hip).
    h: (M, D) bf16; weight: (V, D) fp32 parameter; wpack: its bf16 forward pack (V, D); bias: (V,) fp32 or None."""

    @staticmethod
    def forward(ctx, h: Tensor, weight: Tensor, bias: Optional[Tensor], wpack: Tensor, target: Tensor, valid: Optional[Tensor], grad_mode: bool = True):
        lib = _hip.load_library()
        m, d = h.shape
        v = weight.shape[0]
        wp = wpack.reshape(v, -1)
        assert wp.dtype == torch.bfloat16 and wp.is_contiguous() and wp.shape[1] == d
        tgt = target.reshape(m).to(torch.int64).contiguous()
        vk = None if valid is None else valid.reshape(m).to(torch.uint8).contiguous()
        # (needs_input_grad mirrors requires_grad, not the grad mode -- and inside forward() the grad mode is always off: the caller's mode
        # comes in as an argument; under no_grad the lse-only sweep is enough)
        need = grad_mode and any(ctx.needs_input_grad[:3])
        dev = h.device
        ws_n = lib.genie_linear_ce_ws_floats(m, d, v, int(need))
        ws = workspace(ws_n, dev, 'lce')
        _LinearCEFn.last_ws_floats = ws_n                  # (what the tests read: the no-grad sweep asks for the small workspace)
        lse = torch.empty(m, dtype=torch.float32, device=dev)
        row_e = torch.empty((m + 63) // 64 * 64, dtype=torch.float32, device=dev)
        acc = torch.zeros(1, dtype=torch.float32, device=dev)
        dh = torch.empty((m, d), dtype=torch.float32, device=dev) if need else None
        b = None if bias is None else bias.detach()
        prof = _conv.PROFILER if _conv.PROFILER is not None and not _conv.PROFILER.only_triple else None
        t0 = prof.begin() if prof is not None else None
        _hip.check(lib.genie_linear_ce_fwd(h.data_ptr(), h.stride(0), m, d, wp.data_ptr(), wp.stride(0), v, _hip.ptr(b), tgt.data_ptr(), _hip.ptr(vk),
                                           ws.data_ptr(), ws.numel(), lse.data_ptr(), row_e.data_ptr(), acc.data_ptr(), _hip.ptr(dh), _hip.stream_ptr()),
                   'genie_linear_ce_fwd')
        if prof is not None:
            # algorithmic = the reference's products this launch replaces (logits, and d loss / d h when a gradient is wanted): 2 M V D each;
            # executed = the same here (the forward sweep computes S and, with a gradient, softmax(S) W)
            fl = 2.0 * m * v * d * (2 if need else 1)
            prof.end('linear_ce[mfma]', f'linear_ce fwd{"+dh" if need else ""} rows={m} D={d} V={v}', fl, t0, flops_exec=fl)
        count = (vk.sum(dtype=torch.float32) if vk is not None else torch.full((), float(m), device=dev)).reshape(1)
        ctx.save_for_backward(h, wp, b, tgt, row_e, dh, count)
        ctx.weight, ctx.bias = weight, bias
        ctx.lse = lse
        return (acc / count).reshape(())

    @staticmethod
    def backward(ctx, g: Tensor):
        h, wp, b, tgt, row_e, dh, count = ctx.saved_tensors
        weight, bias = ctx.weight, ctx.bias
        lib = _hip.load_library()
        m, d = h.shape
        v = wp.shape[0]
        scale = (g.float().reshape(1) / count).contiguous()
        need_h, need_w = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        need_b = bias is not None and ctx.needs_input_grad[2]
        dhb = torch.empty((m, d), dtype=torch.bfloat16, device=h.device) if need_h else None
        dw = db = gw = gb = None
        if need_w or need_b:
            # straight into the parameters' gradient buffers where both are arena / leaf parameters that WANT a gradient; a frozen weight
            # next to a trainable bias (or the reverse) gets a scratch tensor that is dropped -- never a .grad it did not ask for
            direct_w = need_w and weight.is_leaf and _direct(weight)
            direct_b = need_b and bias.is_leaf and _direct(bias)
            if direct_w:
                gw = _grad_buffer(weight)
            else:
                gw = torch.zeros((v, d), dtype=torch.float32, device=h.device)
                dw = gw if need_w else None
            if need_b:
                if direct_b:
                    gb = _grad_buffer(bias)
                else:
                    gb = db = torch.zeros(v, dtype=torch.float32, device=h.device)
            assert gw.is_contiguous() and gw.dtype == torch.float32
        prof = _conv.PROFILER if _conv.PROFILER is not None and not _conv.PROFILER.only_triple else None
        t0 = prof.begin() if prof is not None else None
        _hip.check(lib.genie_linear_ce_bwd(h.data_ptr(), h.stride(0), m, d, wp.data_ptr(), wp.stride(0), v, _hip.ptr(b), tgt.data_ptr(), row_e.data_ptr(),
                                           scale.data_ptr(), _hip.ptr(dh) if need_h else None, _hip.ptr(dhb), d, _hip.ptr(gw), _hip.ptr(gb),
                                           _hip.stream_ptr()), 'genie_linear_ce_bwd')
        if prof is not None and gw is not None:
            # algorithmic: d loss / d W = one product of the reference (2 M V D); executed: the scores are recomputed (4 M V D)
            prof.end('linear_ce[mfma]', f'linear_ce bwd dW rows={m} D={d} V={v}', 2.0 * m * v * d, t0, flops_exec=4.0 * m * v * d)
        return dhb, dw, db, None, None, None, None


def linear_cross_entropy(h: Tensor, weight: Tensor, bias: Optional[Tensor], wpack: Tensor, target: Tensor, valid: Optional[Tensor] = None) -> Tensor:
    """mean_{rows with valid} CE(h W^T + b, target) without materialising the (M, V) logits (see ``_LinearCEFn``)."""
    _hip.require_gpu(h, 'linear_cross_entropy')
    return _LinearCEFn.apply(h, weight, bias, wpack, target, valid, torch.is_grad_enabled())


# ------------------------------------------------------------------------------------------------
# Skinny linears (min(in, out) <= 32): AdaGN's condition projections, the LFQ projections, cond K / V, LatentAction.to_act
# ------------------------------------------------------------------------------------------------
LINEAR_SMALL = __import__('os').environ.get('GENIE_LINEAR_SMALL', '1') != '0'      # A/B: 0 = every linear through F.linear (rounds 1-5)


def linear_small_supported(in_features: int, out_features: int) -> bool:
    return LINEAR_SMALL and min(int(in_features), int(out_features)) <= 32


class _LinearSmallFn(torch.autograd.Function):
    """y = x W^T + b in fp32 arithmetic (genie_linear_small_fwd; csrc/linear_small.hip) for the weights of the hot path that have a side of
    <= 32 features.  x: (..., K) fp32 or bf16 rows; weight (N, K) fp32; y: (..., N) in `out_dtype`.  Backward: dx = the same kernel over dy with
    the weight's strides exchanged; dW / db by genie_linear_small_wgrad (partials summed in a fixed order: no atomics), straight into the
    parameters' gradient buffers where they are arena / leaf parameters."""

    @staticmethod
    def forward(ctx, x: Tensor, weight: Tensor, bias: Optional[Tensor], out_dtype: torch.dtype):
        lib = _hip.load_library()
        n, k = weight.shape
        lead = x.shape[:-1]
        rows = x.reshape(-1, k)
        if rows.dtype not in (torch.float32, torch.bfloat16):
            rows = rows.float()
        if rows.stride(-1) != 1 or (rows.shape[0] > 1 and rows.stride(0) < k):
            rows = rows.contiguous()
        w = weight.detach()
        assert w.dtype == torch.float32 and w.dim() == 2
        m = rows.shape[0]
        y = torch.empty((m, n), dtype=out_dtype, device=x.device)
        dt = lambda t: _hip.GENIE_F32 if t.dtype == torch.float32 else _hip.GENIE_BF16
        ws_n = lib.genie_linear_small_ws_floats(m, k, n)
        ws = workspace(ws_n, x.device, 'lin') if ws_n else None
        _hip.check(lib.genie_linear_small_fwd(rows.data_ptr(), dt(rows), rows.stride(0) if m > 1 else k, m, k, w.data_ptr(), w.stride(0), w.stride(1),
                                              _hip.ptr(None if bias is None else bias.detach()), y.data_ptr(), dt(y), n, n, _hip.ptr(ws), ws_n, _hip.stream_ptr()),
                   'genie_linear_small_fwd')
        ctx.save_for_backward(rows)
        ctx.weight, ctx.bias, ctx.x_shape, ctx.x_dtype = weight, bias, tuple(x.shape), x.dtype
        return y.reshape(*lead, n)

    @staticmethod
    def backward(ctx, dy: Tensor):
        (rows,) = ctx.saved_tensors
        weight, bias = ctx.weight, ctx.bias
        lib = _hip.load_library()
        n, k = weight.shape
        m = rows.shape[0]
        g = dy.reshape(-1, n)
        if g.dtype not in (torch.float32, torch.bfloat16):
            g = g.float()
        if g.stride(-1) != 1 or (m > 1 and g.stride(0) < n):
            g = g.contiguous()
        dt = lambda t: _hip.GENIE_F32 if t.dtype == torch.float32 else _hip.GENIE_BF16
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            w = weight.detach()
            dxr = torch.empty((m, k), dtype=torch.float32 if ctx.x_dtype == torch.float32 else torch.bfloat16, device=dy.device)
            ws_n = lib.genie_linear_small_ws_floats(m, n, k)
            ws = workspace(ws_n, dy.device, 'lin') if ws_n else None
            # dx[m][k] = sum_n dy[m][n] W[n][k]: "weight" W'[k][n] = W[n][k], i.e. the strides exchanged
            _hip.check(lib.genie_linear_small_fwd(g.data_ptr(), dt(g), g.stride(0) if m > 1 else n, m, n, w.data_ptr(), w.stride(1), w.stride(0), None,
                                                  dxr.data_ptr(), dt(dxr), k, k, _hip.ptr(ws), ws_n, _hip.stream_ptr()), 'genie_linear_small_fwd (dx)')
            dx = dxr.reshape(ctx.x_shape).to(ctx.x_dtype)
        need_w, need_b = ctx.needs_input_grad[1], bias is not None and ctx.needs_input_grad[2]
        if need_w or need_b:
            direct_w = need_w and weight.is_leaf and _direct(weight)
            direct_b = need_b and bias.is_leaf and _direct(bias)
            gw = _grad_buffer(weight) if direct_w else torch.zeros((n, k), dtype=torch.float32, device=dy.device)
            gb = (_grad_buffer(bias) if direct_b else torch.zeros(n, dtype=torch.float32, device=dy.device)) if need_b else None
            ws_n = lib.genie_linear_small_wgrad_ws_floats(m, n, k)
            ws = workspace(ws_n, dy.device, 'linw')
            _hip.check(lib.genie_linear_small_wgrad(g.data_ptr(), dt(g), g.stride(0) if m > 1 else n, rows.data_ptr(), dt(rows), rows.stride(0) if m > 1 else k,
                                                    m, n, k, gw.data_ptr(), gw.stride(0), gw.stride(1), _hip.ptr(gb), ws.data_ptr(), ws.numel(), _hip.stream_ptr()),
                       'genie_linear_small_wgrad')
            dw = None if (direct_w or not need_w) else gw
            db = None if (direct_b or not need_b) else gb
        return dx, dw, db, None


def linear(x: Tensor, weight: Tensor, bias: Optional[Tensor] = None, out_dtype: Optional[torch.dtype] = None) -> Tensor:
    """``F.linear`` for the hot path's skinny weights (one side <= 32 features, fp32 parameters) on the HIP kernels of csrc/linear_small.hip;
    anything else -- square projections of non-default blueprints (``to_q`` / ``to_out`` with d_inp != n_head * d_head) -- stays a library GEMM."""
    n, k = weight.shape
    if x.is_cuda and weight.dtype == torch.float32 and linear_small_supported(k, n) and (bias is None or bias.dtype == torch.float32):
        return _LinearSmallFn.apply(x, weight, bias, out_dtype if out_dtype is not None else (x.dtype if x.dtype in (torch.float32, torch.bfloat16) else torch.float32))
    y = torch.nn.functional.linear(x.to(weight.dtype), weight, bias)
    return y if out_dtype is None else y.to(out_dtype)


# ------------------------------------------------------------------------------------------------
# Embedding lookup whose gradient goes straight into the parameter's gradient buffer
# ------------------------------------------------------------------------------------------------
class _EmbeddingFn(torch.autograd.Function):
    """``nn.Embedding`` lookup (genie_embedding_fwd); backward scatters the incoming rows straight into the weight's gradient buffer
    (genie_embedding_bwd: one fp32 atomic per distinct index per 64-row chunk and column) instead of building a dense (V, D) gradient that
    AccumulateGrad then adds to ``.grad`` -- at V = 2^18, D = 512 that dense detour is 537 MB written and 1.6 GB of read-modify-write per
    step for at most B*T*H*W touched rows.  (Rounds 3-5 used torch's gather / index_add_ here: 0.6 ms per step at 32768 rows, three quarters of
    them the ONE fill token of the MaskGIT loss.)"""

    @staticmethod
    def forward(ctx, idx: Tensor, weight: Tensor):
        _hip.require_gpu(weight, 'embedding')
        w = weight if weight.dtype == torch.float32 and weight.is_contiguous() else weight.float().contiguous()
        ix = idx.reshape(-1).to(torch.int64).contiguous()
        v, d = w.shape
        out = torch.empty((*idx.shape, d), dtype=torch.float32, device=w.device)
        _hip.check(_hip.load_library().genie_embedding_fwd(ix.data_ptr(), w.data_ptr(), out.data_ptr(), ix.numel(), d, v, _hip.stream_ptr()), 'genie_embedding_fwd')
        ctx.save_for_backward(ix)
        ctx.weight = weight
        return out if weight.dtype == torch.float32 else out.to(weight.dtype)

    @staticmethod
    def backward(ctx, dy: Tensor):
        (ix,) = ctx.saved_tensors
        w = ctx.weight
        if not ctx.needs_input_grad[1]:
            return None, None
        v, d = w.shape
        rows = dy.reshape(-1, d)
        if rows.dtype not in (torch.float32, torch.bfloat16):
            rows = rows.float()
        rows = rows.contiguous()
        direct = _direct(w) and w.is_leaf and w.dtype == torch.float32
        g = _grad_buffer(w) if direct else torch.zeros((v, d), dtype=torch.float32, device=w.device)
        _hip.check(_hip.load_library().genie_embedding_bwd(ix.data_ptr(), rows.data_ptr(), _hip.GENIE_F32 if rows.dtype == torch.float32 else _hip.GENIE_BF16,
                                                           g.data_ptr(), ix.numel(), d, v, _hip.stream_ptr()), 'genie_embedding_bwd')
        return None, (None if direct else g.to(w.dtype))


def embedding(idx: Tensor, weight: Tensor) -> Tensor:
    """``nn.Embedding`` lookup (reference dynamics.py:31-38) with a sparse backward (see ``_EmbeddingFn``)."""
    return _EmbeddingFn.apply(idx, weight)


# ------------------------------------------------------------------------------------------------
# MaskGIT sampling step (DynamicsModel.generate)
# ------------------------------------------------------------------------------------------------
def maskgit_sample(logits: Tensor, u: Tensor, temp: float = 1.):
    """logits: (B, n, V) or (B, h, w, V) bf16/fp32 view whose rows (last dim) are contiguous and whose positions are evenly pitched
    inside a sample; u: (B * n,) uniforms in [0, 1).  Returns (pred int64 (B, n), conf fp32 (B, n)): the inverse-CDF draw from
    softmax(logits / temp) and its probability (reference dynamics.py:138-143 with injected noise)."""
    _hip.require_gpu(logits, 'maskgit_sample')
    if logits.dtype not in (torch.bfloat16, torch.float32):
        logits = logits.float()            # FIRST: the conversion re-packs a strided view, pitches are read from the result
    if logits.dim() == 4:
        b, h, w, v = logits.shape
        if h > 1 and w > 1 and logits.stride(1) != w * logits.stride(2):
            logits = logits.contiguous()
        n, pitch = h * w, (logits.stride(2) if w > 1 else logits.stride(1))
    else:
        b, n, v = logits.shape
        pitch = logits.stride(1)
    if (v > 1 and logits.stride(-1) != 1) or (n > 1 and pitch < v):
        logits = logits.reshape(b, n, v).contiguous()
        pitch = v
    if n == 1:
        pitch = v
    sample_stride = logits.stride(0) if b > 1 else n * pitch
    u = u.reshape(-1).to(device=logits.device, dtype=torch.float32).contiguous()
    if u.numel() != b * n:
        raise ValueError(f'maskgit_sample: {u.numel()} uniforms for {b * n} rows')
    pred = torch.empty((b, n), dtype=torch.int64, device=logits.device)
    conf = torch.empty((b, n), dtype=torch.float32, device=logits.device)
    dt = _hip.GENIE_F32 if logits.dtype == torch.float32 else _hip.GENIE_BF16
    _hip.check(_hip.load_library().genie_maskgit_sample(logits.data_ptr(), dt, b * n, n, sample_stride, pitch, v, u.data_ptr(), float(temp),
                                                        pred.data_ptr(), conf.data_ptr(), _hip.stream_ptr()), 'genie_maskgit_sample')
    return pred, conf


def maskgit_paint(conf: Tensor, pred: Tensor, k: int, code: Tensor, mask: Tensor) -> None:
    """In place: per sample, the k most confident still-masked positions take their sampled token and leave the mask
    (reference dynamics.py:146-158).  conf fp32 / pred int64 / code int64 / mask uint8, all (B, n) contiguous."""
    _hip.require_gpu(conf, 'maskgit_paint')
    b, n = conf.shape
    for t, dt in ((conf, torch.float32), (pred, torch.int64), (code, torch.int64), (mask, torch.uint8)):
        if t.dtype != dt or not t.is_contiguous() or tuple(t.shape) != (b, n):
            raise ValueError('maskgit_paint: conf fp32 / pred int64 / code int64 / mask uint8, contiguous (B, n)')
    _hip.check(_hip.load_library().genie_maskgit_paint(conf.data_ptr(), pred.data_ptr(), b, n, int(k), code.data_ptr(), mask.data_ptr(),
                                                       _hip.stream_ptr()), 'genie_maskgit_paint')


# ------------------------------------------------------------------------------------------------
# VideoResidualBlock as ONE autograd node (reference video.py:588-648, the default configuration: GroupNorm + swish, no
# downsampling):   out = conv_b(act(GN2(conv_a(act(GN1(x)))))) + conv_r(x)
# Same kernels as the module-by-module composition; what the fusion buys is the backward of the fan-out at x: the 1x1x1
# residual conv's backward-data pass takes the main branch's input gradient as its epilogue addend, so autograd never runs a
# separate bf16 add over the activation (43 of them per step in the MAGVIT2 tokenizer).
# ------------------------------------------------------------------------------------------------
def _gn_fwd_raw(x: Tensor, gamma: Tensor, beta: Tensor, groups: int, eps: float, act: int):
    n, c, t, h, w = x.shape
    lib = _hip.load_library()
    y = empty_like_cl(x)
    mean = torch.empty(n * groups, dtype=torch.float32, device=x.device)
    rstd = torch.empty_like(mean)
    ws = workspace(lib.genie_groupnorm_ws_floats(n, c, groups), x.device, 'gn')
    P = _hip.ptr
    _hip.check(lib.genie_groupnorm_fwd(P(x), P(y), n, t * h * w, c, pitch_of(x), groups, P(gamma), P(beta), None, None, eps, act, P(mean), P(rstd),
                                       P(ws), _hip.stream_ptr()), 'genie_groupnorm_fwd')
    return y, mean, rstd


GN_FUSE_COUNT = {'fwd': 0, 'bwd': 0}     # how often a GroupNorm pass was served from a conv epilogue (tests / diagnostics)


def _gn_fwd_from_sums(x: Tensor, gamma: Tensor, beta: Tensor, eps: float, act: int, sums: Tensor):
    """One-group GroupNorm whose statistics the producing conv accumulated in its epilogue (conv_forward(gn_sums=...)): finalize + apply."""
    GN_FUSE_COUNT['fwd'] += 1
    n, c, t, h, w = x.shape
    lib = _hip.load_library()
    y = empty_like_cl(x)
    mean = torch.empty(n, dtype=torch.float32, device=x.device)
    rstd = torch.empty_like(mean)
    P = _hip.ptr
    _hip.check(lib.genie_groupnorm_fwd_from_sums(P(x), P(y), n, t * h * w, c, pitch_of(x), P(gamma), P(beta), eps, act, P(mean), P(rstd), P(sums),
                                                 _hip.stream_ptr()), 'genie_groupnorm_fwd_from_sums')
    return y, mean, rstd


def _gn_bwd_from_part(x: Tensor, dy: Tensor, gamma: Tensor, beta: Tensor, mean: Tensor, rstd: Tensor, act: int, part: Tensor, nblk: int) -> Tensor:
    """GroupNorm backward whose reduce pass ran in the epilogue of the conv that produced `dy` (conv_dgrad(gnb=...)): finalize + apply."""
    GN_FUSE_COUNT['bwd'] += 1
    n, c, t, h, w = x.shape
    lib = _hip.load_library()
    dx = empty_like_cl(x)
    ws = workspace(lib.genie_groupnorm_bwd_from_part_ws_floats(n), x.device, 'gnp')
    P = _hip.ptr
    _hip.check(lib.genie_groupnorm_bwd_from_part(P(x), P(dy), P(dx), n, t * h * w, c, pitch_of(x), P(gamma), P(beta), act, P(mean), P(rstd),
                                                 P(_grad_buffer(gamma)), P(_grad_buffer(beta)), P(part), nblk, P(ws), _hip.stream_ptr()),
               'genie_groupnorm_bwd_from_part')
    return dx


def _gn_bwd_raw(x: Tensor, dy: Tensor, gamma: Tensor, beta: Tensor, mean: Tensor, rstd: Tensor, groups: int, act: int) -> Tensor:
    """dx; the parameter gradients accumulate into gamma.grad / beta.grad."""
    n, c, t, h, w = x.shape
    lib = _hip.load_library()
    dx = empty_like_cl(x)
    ws = workspace(lib.genie_groupnorm_ws_floats(n, c, groups), x.device, 'gn')
    P = _hip.ptr
    _hip.check(lib.genie_groupnorm_bwd(P(x), P(dy), P(dx), n, t * h * w, c, pitch_of(x), groups, P(gamma), P(beta), None, None, act, P(mean), P(rstd),
                                       P(_grad_buffer(gamma)), P(_grad_buffer(beta)), None, None, P(ws), _hip.stream_ptr()), 'genie_groupnorm_bwd')
    return dx


_RESBLOCK_OUT_SUMS = {}


class _ResBlockFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, g1w, g1b, wa, ba, g2w, g2b, wb, bb, wr, br, ops, groups: int, eps1: float, eps2: float, x_sums=None):
        op_a, op_b, op_r = ops
        _conv_gate()
        # GroupNorm statistics ride on the producing conv's epilogue where that kernel can do it (one group): conv_a's output feeds the
        # second norm of this block, conv_b's output (the block output) the first norm of the NEXT block (`x_sums`, handed over on the tensor)
        if groups == 1 and x_sums is not None:
            xn, m1, r1 = _gn_fwd_from_sums(x, g1w, g1b, eps1, 1, x_sums)
        else:
            xn, m1, r1 = _gn_fwd_raw(x, g1w, g1b, groups, eps1, 1)
        s1 = [] if groups == 1 else None
        h1 = conv_forward(xn, op_a.pack_fwd(wa), ba, op_a.spec, gn_sums=s1)
        if s1:
            hn, m2, r2 = _gn_fwd_from_sums(h1, g2w, g2b, eps2, 1, s1[0])
        else:
            hn, m2, r2 = _gn_fwd_raw(h1, g2w, g2b, groups, eps2, 1)
        r = conv_forward(x, op_r.pack_fwd(wr), br, op_r.spec)
        s2 = [] if groups == 1 else None
        out = conv_forward(hn, op_b.pack_fwd(wb), bb, op_b.spec, resid=r, gn_sums=s2)
        _RESBLOCK_OUT_SUMS['last'] = s2[0] if s2 else None
        ctx.ops, ctx.groups = ops, groups
        ctx.size = tuple(x.shape[2:])
        ctx.save_for_backward(x, xn, h1, hn, m1, r1, m2, r2, g1w, g1b, wa, ba, g2w, g2b, wb, bb, wr, br)
        return out

    @staticmethod
    def backward(ctx, dy: Tensor):
        x, xn, h1, hn, m1, r1, m2, r2, g1w, g1b, wa, ba, g2w, g2b, wb, bb, wr, br = ctx.saved_tensors
        op_a, op_b, op_r = ctx.ops
        dy = to_cl(dy)
        gb = lambda b: _grad_buffer(b) if b is not None else None
        _conv_gate()
        # the reduce pass of each GroupNorm backward rides on the epilogue of the backward-data conv that produces its output gradient
        f2 = _conv.GnBwdFuse(h1, g2w, g2b, m2, r2, 1) if ctx.groups == 1 else None
        d_hn = conv_dgrad(dy, op_b.pack_bwd(wb), op_b.spec, ctx.size, gnb=f2)
        wg = _wgrad if getattr(wb, '_genie_arena', False) else conv_wgrad
        wg(hn, dy, op_b.spec, _grad_buffer(wb), gb(bb))
        if f2 is not None and f2.fused:
            d_h1 = _gn_bwd_from_part(h1, d_hn, g2w, g2b, m2, r2, 1, f2.part, f2.nblk)
        else:
            d_h1 = _gn_bwd_raw(h1, d_hn, g2w, g2b, m2, r2, ctx.groups, 1)
        del d_hn, f2
        _conv_gate()                                           # wgrad of conv_b ran under the GroupNorm backward above
        f1 = _conv.GnBwdFuse(x, g1w, g1b, m1, r1, 1) if ctx.groups == 1 and ctx.needs_input_grad[0] else None
        d_xn = conv_dgrad(d_h1, op_a.pack_bwd(wa), op_a.spec, ctx.size, gnb=f1)
        wg(xn, d_h1, op_a.spec, _grad_buffer(wa), gb(ba))
        del d_h1
        wg(x, dy, op_r.spec, _grad_buffer(wr), gb(br))
        dx = None
        if ctx.needs_input_grad[0]:
            if f1 is not None and f1.fused:
                d_xm = _gn_bwd_from_part(x, d_xn, g1w, g1b, m1, r1, 1, f1.part, f1.nblk)
            else:
                d_xm = _gn_bwd_raw(x, d_xn, g1w, g1b, m1, r1, ctx.groups, 1)
            dx = conv_dgrad(dy, op_r.pack_bwd(wr), op_r.spec, ctx.size, resid=d_xm)      # dgrad of the 1x1x1 conv + main-branch gradient
        return (dx,) + (None,) * 15


def residual_block(x: Tensor, norm1, conv_a, norm2, conv_b, conv_r) -> Optional[Tensor]:
    """Fused VideoResidualBlock when every piece is in the standard form (fp32 leaf parameters, direct gradient accumulation);
    returns None when the caller must fall back to the module-by-module composition."""
    params = [norm1.weight, norm1.bias, conv_a.weight, conv_a.bias, norm2.weight, norm2.bias, conv_b.weight, conv_b.bias, conv_r.weight, conv_r.bias]
    if not FUSED_RESBLOCK or not torch.is_grad_enabled():
        return None
    for p in params:
        if p is not None and not (p.is_leaf and p.requires_grad and p.dtype == torch.float32 and _direct(p)):
            return None
    if any(p is None for p in (norm1.weight, norm1.bias, norm2.weight, norm2.bias)) or norm1.num_groups != norm2.num_groups:
        return None
    if not (norm1.weight.is_contiguous() and norm2.weight.is_contiguous()):
        return None
    x = to_cl(x)
    fn = _ResBlockFn
    out = fn.apply(x, norm1.weight, norm1.bias, conv_a.weight, conv_a.bias, norm2.weight, norm2.bias, conv_b.weight, conv_b.bias,
                   conv_r.weight, conv_r.bias, (conv_a.op, conv_b.op, conv_r.op), norm1.num_groups, norm1.eps, norm2.eps,
                   getattr(x, '_genie_gn_sums', None))
    sums = _RESBLOCK_OUT_SUMS.pop('last', None)
    if sums is not None:
        out._genie_gn_sums = sums          # one-group statistics of `out`, for the GroupNorm that opens the next residual block
    return out
