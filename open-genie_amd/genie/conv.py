"""Host side of the Conv3d family: geometry -> tap tables -> ``genie_conv_igemm`` / ``genie_conv_wgrad``.

One gather-GEMM kernel serves forward, backward-data (stride 1; strided, one launch per input parity
class; and through the depth-to-space-time shuffle) and the 1x1x1 convolutions -- only the tap table,
the weight pack and the destination mapping differ (DESIGN.md section "Conv3d").

Reference semantics:
  CausalConv3d      genie/module/video.py:154-192   time pad (kt-1)*dil + (1 - stride), all in front
  nn.Conv3d(pad=p)  genie/module/video.py:580-620   symmetric zero padding
  depth-to-space    genie/module/video.py:403-408   'b (c p q r) t h w -> b c (t p) (h q) (w r)'
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass
from functools import lru_cache
from typing import Optional, Tuple

import torch
from torch import Tensor

from . import _hip
from .cl import cpitch, empty_cl, is_cl, pitch_of

Triple = Tuple[int, int, int]


class LaunchProfiler:
    """Optional per-launch HIP-event timing of the conv kernels (bench.py's roofline leg).  Events are recorded
    on the stream the kernels are launched on; nothing is synchronised until ``summary()``."""

    def __init__(self, only_triple: bool = False, narrow: bool = True) -> None:
        self.records = []          # (variant, label, flops, start_event, end_event)
        # only_triple: time only the forward / backward-data launches that carry a kw-triple schedule (the dominant kernel).
        # Two events around each of the ~360 conv launches of a step cost ~4 % of the step; around these ~100, under 1 %.
        self.only_triple = only_triple
        # narrow: the stem / head convs' HBM-bound launches too (five per step: their in-step rate is the one that counts -- a stand-alone
        # microbench of the head forward read 0.54 or 0.59 of 8 TB/s depending on what had run on the chip before it)
        self.narrow = narrow

    def times_narrow(self) -> bool:
        return self.narrow or not self.only_triple

    def begin(self):
        ev = torch.cuda.Event(enable_timing=True)
        ev.record(torch.cuda.current_stream())
        return ev

    def end(self, variant: str, label: str, flops: float, start, bytes_: float = 0.0, flops_exec: Optional[float] = None) -> None:
        """`flops`: the ALGORITHMIC (dense) count of the launch, 2 M Cout Cin taps -- SURVEY.md 8(d)'s formula, zero padding included.
        `flops_exec`: what the kernel actually issues -- the kw-triple kernels skip the (frame, dt) pairs whose source frame is time
        padding (tri_trim_range): 2 of 48 at 16 frames, 3 of 48 for a causal conv (None = the same as `flops`).
        `bytes_`: algorithmic HBM bytes of the launch (traffic-bound kernels: attention on short sequences); 0 = priced by `flops` alone."""
        ev = torch.cuda.Event(enable_timing=True)
        ev.record(torch.cuda.current_stream())
        self.records.append((variant, label, flops, start, ev, bytes_, flops if flops_exec is None else flops_exec))

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for variant, label, flops, s, e, nbytes, fexec in self.records:
            ms = s.elapsed_time(e)
            v = out.setdefault(variant, {'launches': 0, 'ms': 0.0, 'flops': 0.0, 'bytes': 0.0, 'flops_exec': 0.0, 'by_label': {}})
            v['launches'] += 1; v['ms'] += ms; v['flops'] += flops; v['bytes'] += nbytes; v['flops_exec'] += fexec
            b = v['by_label'].setdefault(label, {'launches': 0, 'ms': 0.0, 'flops': 0.0, 'bytes': 0.0, 'flops_exec': 0.0})
            b['launches'] += 1; b['ms'] += ms; b['flops'] += flops; b['bytes'] += nbytes; b['flops_exec'] += fexec
        return out


PROFILER: Optional[LaunchProfiler] = None


VARIANT_NAMES = {0: 'igemm_kernel<128,generic>', 1: 'igemm_kernel<128,smallc>', 2: 'igemm_kernel<32,generic>', 3: 'igemm_kernel<32,smallc>',
                 4: 'igemm3_kernel<128>', 5: 'igemm3_kernel<256>', 6: 'gemm_pw_kernel', 7: 'igemm3_kernel<256,splitk>', 8: 'wgrad_kernel<128,128>', 9: 'wgrad_kernel<128,32>',
                 10: 'wgrad_kernel<32,128>', 11: 'wgrad3_kernel', 12: 'igemm3_kernel<256x256>', 13: 'wgrad_pw_kernel<256x256>', 14: 'wgrad3l_kernel', 15: 'igemm3_kernel<256,k32>'}


def _variant(kind: str, spec: 'ConvSpec', ncols: int, small_c: bool) -> str:
    """Name of the kernel the last launch of this thread used (reported by the library)."""
    return VARIANT_NAMES.get(_hip.load_library().genie_last_conv_variant(), 'unknown')


@dataclass(frozen=True)
class ConvSpec:
    cin: int
    cout: int                 # number of weight rows (for an upsample conv: Cf * P * Q * R)
    kernel: Triple
    stride: Triple = (1, 1, 1)
    dilation: Triple = (1, 1, 1)
    pad_front: Triple = (0, 0, 0)
    pad_back: Triple = (0, 0, 0)
    shuffle: Optional[Triple] = None   # (P, Q, R) depth-to-space-time factors applied to the output

    @property
    def ntaps(self) -> int:
        return self.kernel[0] * self.kernel[1] * self.kernel[2]

    @property
    def cinp(self) -> int:
        return cpitch(self.cin)

    @property
    def coutp(self) -> int:
        return cpitch(self.cout)

    def out_size(self, size: Triple) -> Triple:
        return tuple((size[i] + self.pad_front[i] + self.pad_back[i] - self.dilation[i] * (self.kernel[i] - 1) - 1)
                     // self.stride[i] + 1 for i in range(3))

    @property
    def cfinal(self) -> int:
        if self.shuffle is None:
            return self.cout
        p, q, r = self.shuffle
        return self.cout // (p * q * r)


def causal_spec(cin: int, cout: int, kernel: Triple, stride: Triple = (1, 1, 1), dilation: Triple = (1, 1, 1),
                space_pad=(None, None), shuffle=None) -> ConvSpec:
    """video.py:154-164."""
    kt, kh, kw = kernel
    tp = (kt - 1) * dilation[0] + (1 - stride[0])
    hp = space_pad[0] if space_pad[0] is not None else (kh - 1) // 2
    wp = space_pad[1] if space_pad[1] is not None else (kw - 1) // 2
    # tp < 0 (kt = 1 with a time stride of 2): the reference's F.pad with a negative amount CROPS the first -tp frames (video.py:154-164);
    # the module slices them off (CausalConv3d.forward) and the conv itself runs without time padding
    return ConvSpec(cin, cout, tuple(kernel), tuple(stride), tuple(dilation), (max(tp, 0), hp, wp), (0, hp, wp), shuffle)


def causal_time_crop(kernel: Triple, stride: Triple = (1, 1, 1), dilation: Triple = (1, 1, 1)) -> int:
    """Frames the reference's negative causal padding removes from the FRONT of the clip: max(0, -((kt - 1) dil_t + 1 - stride_t))."""
    return max(0, -((kernel[0] - 1) * dilation[0] + (1 - stride[0])))


def same_spec(cin: int, cout: int, kernel: Triple) -> ConvSpec:
    """nn.Conv3d(kernel, padding=(k-1)//2), stride 1."""
    pad = tuple((k - 1) // 2 for k in kernel)
    return ConvSpec(cin, cout, tuple(kernel), (1, 1, 1), (1, 1, 1), pad, pad, None)


# ------------------------------------------------------------------------------------------------
# tap tables (cached on the device)
# ------------------------------------------------------------------------------------------------
_tap_cache = {}


def _upload_taps(key, taps):
    dev = torch.cuda.current_device()
    hit = _tap_cache.get((dev, key))
    if hit is not None:
        return hit
    arr = torch.tensor([[t[0], t[1], t[2], t[3], t[4], t[5], 0, 0] for t in taps], dtype=torch.int32).reshape(-1, 8)
    dev_arr = arr.cuda()
    nk = sum((t[5] + 63) // 64 for t in taps)
    _tap_cache[(dev, key)] = (dev_arr, len(taps), nk)
    return _tap_cache[(dev, key)]


# kw-triple schedule (conv_igemm3.hip): GENIE_TRI = 0 choose the row tile (default), 128 / 256 force it, -1 never use it;
# GENIE_TRI_FLAGS bit 0 = drain every barrier (debug / A-B timing)
TRI_BM = int(os.environ.get('GENIE_TRI', '0'))
TRI_FLAGS = int(os.environ.get('GENIE_TRI_FLAGS', '0'))
TRI_WGRAD = int(os.environ.get('GENIE_TRI_WGRAD', '1'))          # conv_wgrad3.hip: 0 never, 1 when it pays, 2 whenever eligible
_tri_cache = {}
FORCE_SPLIT_K = 0                   # experiments only: split-K factor handed to genie_conv_wgrad (0 = library chooses)
# GENIE_DETERMINISTIC=1 (or conv.set_deterministic(True)): weight gradients with ONE K split per output tile -- a single owner adds each
# element in a fixed order, so they are bit-reproducible run to run (the default split-K partial sums meet in fp32 atomics whose
# order the hardware picks: reproducible to ~1e-6 relative, tests/test_gpu_properties.py::test_gradient_determinism).  Slower (the
# low-resolution layers lose their parallelism); the stem / head weight gradients go through the generic kernel in this mode.
# Bias gradients in this mode: a fixed-order column sum (_bias_grad_fixed_order) instead of the kernels' epilogue atomics.
DETERMINISTIC = os.environ.get('GENIE_DETERMINISTIC', '0') not in ('0', '')


def set_deterministic(on: bool) -> bool:
    global DETERMINISTIC
    old, DETERMINISTIC = DETERMINISTIC, bool(on)
    return old


TRI_DH_INNER = os.environ.get('GENIE_TRI_DH_INNER', '1') not in ('0', '')   # step-table order inside a dt: (channel block, dh) instead of (dh, channel block)
TRI_TRIM = os.environ.get('GENIE_TRI_TRIM', '1') not in ('0', '')       # A/B switch: 0 = padding frames are staged and multiplied like any other


def tri_rows(taps, hs: int, ws: int, cs: int):
    """Pure host logic of the kw-triple schedule: group a tap list [(dt, dh, dw, wofs, c0, nch)] into one row per
    (dt, dh, 64-channel block): [a_delta, dt, dh, wofs(dw=-1), wofs(dw=0), wofs(dw=+1), 0, 0] (struct GenieTriStep), or None when
    the taps do not come as complete dw = -1, 0, +1 triples over whole 64-channel blocks."""
    groups = {}
    for (dt, dh, dw, wofs, c0, nch) in taps:
        groups.setdefault((dt, dh, c0, nch), {})[dw] = wofs
    if not groups or len(taps) != 3 * len(groups):
        return None
    rows, chan = [], []
    for (dt, dh, c0, nch), by_dw in groups.items():
        if sorted(by_dw) != [-1, 0, 1] or nch % 64 != 0 or c0 % 8 != 0:
            return None
        for cb in range(nch // 64):
            rows.append([((dt * hs + dh) * ws) * cs + c0 + cb * 64, dt, dh, by_dw[-1] + cb * 64, by_dw[0] + cb * 64, by_dw[1] + cb * 64, 0, 0])
            chan.append(c0 + cb * 64)
    # rows sorted by dt, every dt owning the same number of consecutive rows: a row tile inside frame t can then skip the rows whose
    # frame t + dt is padding (GenieTriStep.rows_per_dt / dt_min; 2 of 48 (frame, dt) pairs of a 16-frame 'same' conv, 3 of a causal one).
    # Inside a dt: channel block OUTER, dh INNER (TRI_DH_INNER, default) -- the three images of a (dt, channel block) are the same image rows
    # shifted by one (7 of 8 rows shared), so the second and third come out of the XCD's L2 while it still holds them (40 KB per block in
    # between); in the (dh, channel block) order of rounds 2-4 four channel-block images per resident block (5 MB per XCD > its 4-MB L2) sat
    # between two reads of a row and every one of them went back to the fabric: FETCH_SIZE 5.3 x the input of a 256 -> 256 @16x32x32 launch.
    order = sorted(range(len(rows)), key=(lambda i: (rows[i][1], chan[i], rows[i][2])) if TRI_DH_INNER else (lambda i: rows[i][1]))
    rows = [rows[i] for i in order]
    dts = [r[1] for r in rows]
    lo, n = dts[0], dts.count(dts[0])
    if TRI_TRIM and all(dts[i] == lo + i // n for i in range(len(rows))):
        for r in rows:
            r[6], r[7] = n, lo
    return rows


def tri_schedule(key, taps, hs: int, ws: int, cs: int):
    """Device copy of ``tri_rows`` (cached per device / conv / geometry): (int32 [n, 8] tensor, n) or None."""
    dev = torch.cuda.current_device()
    ck = (dev, key, hs, ws, cs)
    if ck in _tri_cache:
        return _tri_cache[ck]
    rows = tri_rows(taps, hs, ws, cs)
    out = None if rows is None else (torch.tensor(rows, dtype=torch.int32).cuda(), len(rows))
    _tri_cache[ck] = out
    _tri_meta[ck] = None if rows is None else (rows[0][6], rows[0][7], len(rows))      # (rows per dt, smallest dt, rows): what tri_trim_range reads
    return out


_tri_meta = {}


def tri_executed_fraction(meta, frames: int, hw: int, rows_total: int, bm: int) -> float:
    """Share of a kw-triple launch's step-table rows that its row tiles actually execute -- the host restatement of tri_trim_range
    (csrc/conv_igemm3.hip): a tile of `bm` output rows that lies inside ONE frame t skips the table rows whose source frame t + dt is
    time padding.  meta = (rows per dt, smallest dt, table rows) of the schedule (rows per dt 0: nothing is skipped)."""
    if meta is None or meta[0] <= 0 or frames <= 0:
        return 1.0
    rpd, dmin, nrows = meta
    ndt = nrows // rpd
    done = total = 0
    per = frames * hw                                    # the pattern repeats per sample when a sample is whole tiles
    span = per if per % bm == 0 else rows_total
    for m0 in range(0, span, bm):
        last = min(m0 + bm - 1, rows_total - 1)
        f0, f1 = m0 // hw, last // hw
        lo, hi = max(0, -(f0 % frames) - dmin), min(ndt, frames - (f0 % frames) - dmin)
        done += (hi - lo) if (f0 == f1 and hi > lo) else ndt
        total += ndt
    return done / total if total else 1.0


def _tri_bm_of_last_launch() -> int:
    """Row-tile height of the kernel genie_conv_igemm just launched (VARIANT_NAMES: 4 = the 128-row kw-triple kernel, the others 256)."""
    return 128 if _hip.load_library().genie_last_conv_variant() == 4 else 256


def _fwd_tap_list(spec: ConvSpec):
    taps = []
    kt, kh, kw = spec.kernel
    j = 0
    for a in range(kt):
        for b in range(kh):
            for c in range(kw):
                taps.append((a * spec.dilation[0] - spec.pad_front[0], b * spec.dilation[1] - spec.pad_front[1],
                             c * spec.dilation[2] - spec.pad_front[2], j * spec.cinp, 0, spec.cinp))
                j += 1
    return taps


def fwd_taps(spec: ConvSpec):
    taps = []
    kt, kh, kw = spec.kernel
    j = 0
    for a in range(kt):
        for b in range(kh):
            for c in range(kw):
                taps.append((a * spec.dilation[0] - spec.pad_front[0], b * spec.dilation[1] - spec.pad_front[1],
                             c * spec.dilation[2] - spec.pad_front[2], j * spec.cinp, 0, spec.cinp))
                j += 1
    return _upload_taps(('fwd', spec), taps)


def dgrad_taps(spec: ConvSpec, parity: Triple, hi_pitch: int, want_list: bool = False):
    """Taps of the backward-data gather for the input positions i = a * stride + parity.

    Plain conv: source (dy) coordinate = a + (parity + pad - k*dil) / stride for the k that divide.
    Shuffled conv (stride 1): dy lives at high resolution; low-res coordinate o and sub-pixel (p,q,r)
    sit at o * (P,Q,R) + (p,q,r), K segment = the Cf channels of that sub-pixel."""
    taps = []
    kt, kh, kw = spec.kernel
    P, Q, R = spec.shuffle if spec.shuffle is not None else (1, 1, 1)
    cf = spec.cfinal
    j = 0
    for a in range(kt):
        for b in range(kh):
            for c in range(kw):
                num = tuple(parity[i] + spec.pad_front[i] - (a, b, c)[i] * spec.dilation[i] for i in range(3))
                if all(num[i] % spec.stride[i] == 0 for i in range(3)):
                    d = tuple(num[i] // spec.stride[i] for i in range(3))
                    if spec.shuffle is None:
                        taps.append((d[0], d[1], d[2], j * spec.coutp, 0, spec.coutp))
                    else:
                        for p in range(P):
                            for q in range(Q):
                                for r in range(R):
                                    sub = (p * Q + q) * R + r
                                    taps.append((d[0] * P + p, d[1] * Q + q, d[2] * R + r, j * spec.coutp + sub * cf, 0, cf))
                j += 1
    if want_list:
        return taps
    return _upload_taps(('dgrad', spec, parity, hi_pitch), taps) if taps else (None, 0, 0)


# ------------------------------------------------------------------------------------------------
# weight packs
# ------------------------------------------------------------------------------------------------
def _weight_view(weight: Tensor):
    """(cout, cin, kt, kh, kw) fp32 -> strides (cout, tap, cin) with the taps flattened."""
    co, ci, kt, kh, kw = weight.shape
    s = weight.stride()
    if kt * kh * kw > 1 and not (s[3] == kw * s[4] and s[2] == kh * s[3]):
        weight = weight.contiguous()
        s = weight.stride()
    return weight, s[0], s[4], s[1]


def pack_weight_fwd(weight: Tensor, spec: ConvSpec) -> Tensor:
    """bf16 [cout][tap][cinp] (K = input channels contiguous)."""
    w, s_co, s_tap, s_ci = _weight_view(weight.detach())
    out = torch.empty((spec.cout, spec.ntaps, spec.cinp), dtype=torch.bfloat16, device=weight.device)
    lib = _hip.load_library()
    _hip.check(lib.genie_pack_weight(w.data_ptr(), out.data_ptr(), spec.cout, spec.ntaps, spec.cin, s_co, s_tap, s_ci,
                                     0, 1, _hip.stream_ptr()), 'genie_pack_weight')
    return out


def pack_weight_bwd(weight: Tensor, spec: ConvSpec) -> Tensor:
    """bf16 [cin][tap][coutp] (K = output channels contiguous; sub-pixel-major order for shuffled convs)."""
    w, s_co, s_tap, s_ci = _weight_view(weight.detach())
    out = torch.empty((spec.cin, spec.ntaps, spec.coutp), dtype=torch.bfloat16, device=weight.device)
    if spec.shuffle is not None:
        p, q, r = spec.shuffle
        perm_c, perm_f = spec.cfinal, p * q * r
    else:
        perm_c, perm_f = 0, 1
    lib = _hip.load_library()
    _hip.check(lib.genie_pack_weight(w.data_ptr(), out.data_ptr(), spec.cin, spec.ntaps, spec.cout, s_ci, s_tap, s_co,
                                     perm_c, perm_f, _hip.stream_ptr()), 'genie_pack_weight')
    return out


# ------------------------------------------------------------------------------------------------
# HBM-bound narrow convolutions (conv_narrow.hip): <= 4 input channels -> 128 output channels, 3x3x3, stride 1 -- the tokenizer's
# stem (video.py:154-192 via MAGVIT2_ENC_DESC[0]) and the backward-data pass of its head conv (128 -> 3)
# ------------------------------------------------------------------------------------------------
UPCONV_DGRAD_UNSHUFFLE = os.environ.get('GENIE_UPCONV_UNSHUFFLE', '1') != '0'
NARROW_WGRAD = os.environ.get('GENIE_NARROW_WGRAD', '1') != '0'
NARROW_CONV = os.environ.get('GENIE_NARROW_CONV', '1') != '0'
_NARROW_W = (32, 64, 128)


def _narrow_geometry_ok(spec: ConvSpec) -> bool:
    return (spec.kernel == (3, 3, 3) and spec.stride == (1, 1, 1) and spec.dilation == (1, 1, 1) and spec.shuffle is None
            and spec.pad_front[1:] == (1, 1) and spec.pad_back[1:] == (1, 1) and spec.pad_front[0] + spec.pad_back[0] == 2)


def _narrow_wide(c: int) -> bool:
    """The wide side of a narrow conv: 128 channels (the tokenizer's stem / head) or a small multiple of them (LatentAction's proj_in / proj_out: 256),
    taken one 128-channel slab per launch."""
    return c in (128, 256, 384, 512)


def narrow_fwd_ok(spec: ConvSpec, x: Tensor) -> bool:
    """Forward of a (<= 4) -> 128 k channel conv on the narrow-input kernel?"""
    return NARROW_CONV and spec.cin <= 4 and _narrow_wide(spec.cout) and _narrow_geometry_ok(spec) and x.shape[4] in _NARROW_W and pitch_of(x) % 4 == 0


def narrow_dgrad_ok(spec: ConvSpec, dy: Tensor) -> bool:
    """Backward-data of a 128 k -> (<= 4) channel conv (= a (<= 4) -> 128 k conv of dy with flipped taps) on the narrow-input kernel?"""
    return NARROW_CONV and _narrow_wide(spec.cin) and spec.cout <= 4 and _narrow_geometry_ok(spec) and dy.shape[4] in _NARROW_W and pitch_of(dy) % 4 == 0


def _narrow_pack(w_rows: Tensor, bias: Optional[Tensor]) -> Tensor:
    """w_rows: fp32 (128, 27, c <= 4) in the kernel's tap order -> bf16 [128][112]: k = tap * 4 + c, k = 108 / 109 = bias as a bf16
    hi / lo pair (the kernel multiplies both by 1.0), k = 110, 111 zero."""
    rows, ntap, c = w_rows.shape
    pack = torch.zeros((rows, 28, 4), dtype=torch.float32, device=w_rows.device)
    pack[:, :27, :c] = w_rows
    if bias is not None:
        hi = bias.detach().float().to(torch.bfloat16).float()
        pack[:, 27, 0] = hi
        pack[:, 27, 1] = bias.detach().float() - hi
    return pack.reshape(rows, 112).to(torch.bfloat16).contiguous()


def pack_narrow_fwd(weight: Tensor, bias: Optional[Tensor]) -> Tensor:
    """weight (128, cin <= 4, 3, 3, 3) -> narrow pack, taps in (kt, kh, kw) order."""
    w = weight.detach().float().permute(0, 2, 3, 4, 1).reshape(weight.shape[0], 27, weight.shape[1])
    return _narrow_pack(w, bias)


def pack_narrow_bwd(weight: Tensor) -> Tensor:
    """weight (cout <= 4, 128, 3, 3, 3) of the FORWARD conv -> pack of its backward-data pass: rows = input channels, taps flipped."""
    w = weight.detach().float().flip(2, 3, 4).permute(1, 2, 3, 4, 0).reshape(weight.shape[1], 27, weight.shape[0])
    return _narrow_pack(w, None)


def narrow_wgrad_ok(spec: ConvSpec, x: Tensor, dy: Tensor) -> bool:
    """Weight gradient of the stem ((<= 4) -> 128) or head (128 -> (<= 4)) conv on the one-pass narrow kernel?"""
    if not (NARROW_CONV and NARROW_WGRAD and not DETERMINISTIC and _narrow_geometry_ok(spec) and x.shape[4] in _NARROW_W and (x.shape[4] != 32 or x.shape[3] % 2 == 0)):
        return False
    if spec.cin <= 4 and _narrow_wide(spec.cout):
        return pitch_of(dy) == spec.cout and pitch_of(x) % 4 == 0
    return _narrow_wide(spec.cin) and spec.cout <= 4 and pitch_of(x) == spec.cin and pitch_of(dy) % 4 == 0


def conv_narrow_wgrad(x: Tensor, dy: Tensor, spec: ConvSpec, dweight: Tensor, dbias: Optional[Tensor], label: str = '') -> None:
    """dW (+ dbias) of a narrow conv accumulated from ONE pass over the 128-channel tensor (genie_conv_narrow_wgrad_acc): the kernel forms
    G[ch][tap * 4 + c] = sum_p big[p][ch] * small[p + tap][c] per workgroup and adds it straight into the parameter gradients.  Stem: big = dy,
    small = x, G[co][tap, ci] is dW[co][ci][tap], a ones column gives db.  Head: big = x, small = dy, G[ci][tap', co] is dW[co][ci][26 - tap'], db is
    the plain sum of dy (taken from the centre tap while the im2col tile is built)."""
    n, _, t, h, w = x.shape
    stem = spec.cin <= 4
    big, small = (dy, x) if stem else (x, dy)
    t_lo = -spec.pad_front[0] if stem else spec.pad_front[0] - 2
    cs = spec.cin if stem else spec.cout
    wcl = 0 if dweight.is_contiguous() else 1 if dweight.permute(0, 2, 3, 4, 1).is_contiguous() else -1        # channels_last_3d is what the modules keep
    if wcl < 0 or not (dbias is None or dbias.is_contiguous()):                              # any other strided view: through temporaries
        dw_c = torch.zeros(dweight.shape, dtype=torch.float32, device=dweight.device)
        db_c = None if dbias is None else torch.zeros(dbias.shape, dtype=torch.float32, device=dbias.device)
        conv_narrow_wgrad(x, dy, spec, dw_c, db_c, label)
        dweight += dw_c
        if dbias is not None:
            dbias += db_c
        return
    t0 = PROFILER.begin() if PROFILER is not None and PROFILER.times_narrow() else None
    # straight into dW / db (the kernel's epilogue knows both parameter layouts): no G tile to zero, no scatter, no torch reduction for the head's bias
    wide = spec.cout if stem else spec.cin
    if wide == 128:
        _hip.check(_hip.load_library().genie_conv_narrow_wgrad_acc(big.data_ptr(), small.data_ptr(), pitch_of(small), dweight.data_ptr(), _hip.ptr(dbias),
                                                                   n, t, h, w, int(t_lo), int(stem), int(cs), wcl, _hip.stream_ptr()), 'genie_conv_narrow_wgrad_acc')
    else:                                                # one launch per 128-channel slab of the wide tensor, all of them into the same dW / db
        for w0 in range(0, wide, 128):
            _hip.check(_hip.load_library().genie_conv_narrow_wgrad_wide(big.data_ptr(), pitch_of(big), w0, wide, small.data_ptr(), pitch_of(small), dweight.data_ptr(),
                                                                        _hip.ptr(dbias), n, t, h, w, int(t_lo), int(stem), int(cs), wcl, _hip.stream_ptr()),
                       'genie_conv_narrow_wgrad_wide')
    if t0 is not None:
        PROFILER.end('conv_narrow_wgrad_kernel', label, 2.0 * n * t * h * w * wide * min(spec.cin, spec.cout) * 27, t0)


def narrow_out_ok(spec: ConvSpec, x: Tensor) -> bool:
    """Forward of a 128 -> (<= 3) channel conv on the narrow-output kernel (the tokenizer's head conv)?"""
    return (NARROW_CONV and spec.cin == 128 and spec.cout <= 3 and _narrow_geometry_ok(spec) and x.shape[4] % 32 == 0 and pitch_of(x) == 128
            and 0 <= spec.pad_front[0] <= 2 and x.shape[3] * x.shape[4] * 256 < 2 ** 32)


def pack_narrow_out(weight: Tensor) -> Tensor:
    """weight (cout <= 3, 128, 3, 3, 3) -> bf16 [16][1152]: row = 4 * dt + co, k = (dh * 3 + dw) * 128 + ci (every other row zero)."""
    co = weight.shape[0]
    w = weight.detach().float().permute(2, 0, 3, 4, 1)                      # (dt, co, dh, dw, ci)
    pack = torch.zeros((4, 4, 1152), dtype=torch.float32, device=weight.device)
    pack[:3, :co] = w.reshape(3, co, 1152)
    return pack.reshape(16, 1152).to(torch.bfloat16).contiguous()


def conv_narrow_out(x: Tensor, pack: Tensor, bias: Optional[Tensor], cout: int, t_lo: int, label: str = '') -> Tensor:
    """x: CL (N, 128, T, H, W); returns CL (N, cout <= 3, T, H, W)."""
    n, c, t, h, w = x.shape
    out = empty_cl(n, cout, t, h, w, x.device, zero_pad=False)           # the kernel writes whole 8-channel pixels (pad channels zero)
    b32 = None if bias is None else bias.detach().float().contiguous()
    t0 = PROFILER.begin() if PROFILER is not None and PROFILER.times_narrow() else None
    _hip.check(_hip.load_library().genie_conv_narrow_out(x.data_ptr(), pack.data_ptr(), _hip.ptr(b32), out.data_ptr(), n, t, h, w, cout, int(t_lo),
                                                         _hip.stream_ptr()), 'genie_conv_narrow_out')
    if t0 is not None:
        PROFILER.end('conv_narrow_out_kernel', label, 2.0 * n * t * h * w * 128 * cout * 27, t0)
    return out


def conv_narrow_in(x: Tensor, pack: Tensor, t_lo: int, label: str = '') -> Tensor:
    """x: CL (N, c <= 4, T, H, W); pack: (128 k, 112) rows of `_narrow_pack`; returns CL (N, 128 k, T, H, W) = sum over the 27 taps (dt in t_lo .. t_lo + 2,
    dh, dw in -1 .. 1) -- one launch per 128 output channels (the kernel keeps 128 weight rows in registers and writes them at the output's pitch)."""
    n, c, t, h, w = x.shape
    cout = pack.shape[0]
    out = empty_cl(n, cout, t, h, w, x.device)
    t0 = PROFILER.begin() if PROFILER is not None and PROFILER.times_narrow() else None
    for c0 in range(0, cout, 128):
        _hip.check(_hip.load_library().genie_conv_narrow_in(x.data_ptr(), pitch_of(x), pack[c0:c0 + 128].data_ptr(), out.data_ptr() + 2 * c0, pitch_of(out), n, t, h, w,
                                                            int(t_lo), _hip.stream_ptr()), 'genie_conv_narrow_in')
    if t0 is not None:
        PROFILER.end('conv_narrow_in_kernel', label, 2.0 * n * t * h * w * cout * c * 27, t0)
    return out


# ------------------------------------------------------------------------------------------------
# launches
# ------------------------------------------------------------------------------------------------
_splitk_cache = {}
SPLITK_WS_FLOATS = 16 << 20          # 64 MiB of fp32 partial tiles per device


def _is_pointwise(spec: 'ConvSpec') -> bool:
    """1x1x1, stride 1, no padding, no shuffle: the conv (and its input gradient) is a plain GEMM over the pixel rows."""
    return (spec.kernel == (1, 1, 1) and spec.stride == (1, 1, 1) and spec.shuffle is None and spec.pad_front == (0, 0, 0)
            and spec.pad_back == (0, 0, 0))


def _splitk_ws(device) -> Tensor:
    key = (device.index if device.index is not None else torch.cuda.current_device(), torch.cuda.current_stream(device).cuda_stream)   # one per stream
    buf = _splitk_cache.get(key)
    if buf is None:
        buf = _splitk_cache[key] = torch.empty(SPLITK_WS_FLOATS, dtype=torch.float32, device=device)
    return buf


def _check_cl(x: Tensor, c: int, what: str):
    if not is_cl(x):
        raise ValueError(f'{what}: expected a CL (bf16 channels-last) tensor')
    if x.shape[1] != c:
        raise ValueError(f'{what}: expected {c} channels, got {x.shape[1]}')


class GnBwdFuse:
    """Request to do the reduce pass of a GroupNorm backward inside the epilogue of the backward-data conv that produces the
    GroupNorm's output gradient (GenieConvDesc.gnb_*): `x` is the GroupNorm input, one group, `act` 0 / 1 (SiLU) / 2 (LeakyReLU).
    After the call `part` (fp32 (N, nblk, Cp, 2)) holds the per-tile partial sums if `fused`."""
    __slots__ = ('x', 'gamma', 'beta', 'mean', 'rstd', 'act', 'part', 'nblk', 'fused')

    def __init__(self, x: Tensor, gamma: Tensor, beta: Tensor, mean: Tensor, rstd: Tensor, act: int) -> None:
        self.x, self.gamma, self.beta, self.mean, self.rstd, self.act = x, gamma, beta, mean, rstd, act
        self.part, self.nblk, self.fused = None, 0, False


# GroupNorm work in conv epilogues -- 0: off (default), 1: forward statistics, 2: + the backward reduce pass.  Built, parity-tested and
# measured at 32 clips: 285.8 / 285.9 / 285.1 ms per step for 0 / 1 / 2 -- what the stand-alone passes cost (stats 3.4 ms, reduce 8.8 ms
# per step) comes back as epilogue time of the MFMA kernels (+9 ms), whose matrix pipes idle while it runs.
GN_FUSE = int(os.environ.get('GENIE_GN_FUSE', '0'))
WGRAD_PW = 1     # GenieWgradDesc.pointwise for 1x1x1 convolutions / Linear layers: 1 = the library decides, 2 = force conv_wgrad_pw.hip (tests)


def _gn_rows_ok(t: int, h: int, w: int) -> bool:
    return GN_FUSE and (t * h * w) % 256 == 0


def conv_forward(x: Tensor, wpack: Tensor, bias: Optional[Tensor], spec: ConvSpec, resid: Optional[Tensor] = None,
                 act: int = 0, gn_sums: Optional[list] = None) -> Tensor:
    """`gn_sums`: pass an empty list to ask for the one-group GroupNorm statistics of the OUTPUT from the epilogue; on return it holds
    the fp64 (N, 2) tensor of (sum, sum of squares) if the kernel that ran could do it, and stays empty otherwise."""
    _check_cl(x, spec.cin, 'conv_forward')
    n, _, t, h, w = x.shape
    to, ho, wo = spec.out_size((t, h, w))
    if min(to, ho, wo) <= 0:
        raise ValueError(f'conv_forward: input {(t, h, w)} too small for kernel {spec.kernel}')
    P, Q, R = spec.shuffle if spec.shuffle is not None else (1, 1, 1)
    cf = spec.cfinal
    out = empty_cl(n, cf, to * P, ho * Q, wo * R, x.device)
    if resid is not None:
        _check_cl(resid, cf, 'conv_forward(resid)')
        assert resid.shape == out.shape and pitch_of(resid) == pitch_of(out)
    taps, ntaps, nk = fwd_taps(spec)
    d = _hip.GenieConvDesc()
    d.src, d.wgt, d.dst = x.data_ptr(), wpack.data_ptr(), out.data_ptr()
    d.resid = _hip.ptr(resid)
    d.bias = _hip.ptr(bias)
    d.taps, d.ntaps, d.nk = taps.data_ptr(), ntaps, nk
    d.small_c = 1 if (spec.cinp in (8, 16, 32) and ntaps <= 32 and pitch_of(x) == spec.cinp) else 0
    d.N, d.Ts, d.Hs, d.Ws, d.Cs = n, t, h, w, pitch_of(x)
    d.To, d.Ho, d.Wo = to, ho, wo
    d.st, d.sh, d.sw = spec.stride
    d.Ncols, d.w_row_stride = spec.cout, spec.ntaps * spec.cinp
    d.Td, d.Hd, d.Wd, d.Cd = to * P, ho * Q, wo * R, pitch_of(out)
    d.dmt, d.dmh, d.dmw = P, Q, R
    d.dot = d.doh = d.dow = 0
    if spec.shuffle is not None:
        d.perm_c, d.perm_f = cf, P * Q * R
        d.shuf_c, d.shuf_q, d.shuf_r = cf, Q, R
    else:
        d.perm_c, d.perm_f = 0, 1
        d.shuf_c, d.shuf_q, d.shuf_r = spec.cout, 1, 1
    d.act = act
    ws = _splitk_ws(x.device)
    d.splitk_ws, d.splitk_ws_bytes = ws.data_ptr(), ws.numel() * 4
    d.tri_bm, d.tri_flags = TRI_BM, TRI_FLAGS
    d.pointwise = int(_is_pointwise(spec) and spec.cinp % 64 == 0)
    if TRI_BM >= 0 and spec.stride == (1, 1, 1) and (to, ho, wo) == (t, h, w) and not d.small_c and pitch_of(x) % 64 == 0:
        sched = tri_schedule(('fwd', spec), _fwd_tap_list(spec), h, w, pitch_of(x))
        if sched is not None:
            d.tri_steps, d.n_tri_steps = sched[0].data_ptr(), sched[1]
    sums = None
    if gn_sums is not None and spec.shuffle is None and d.n_tri_steps > 0 and _gn_rows_ok(to, ho, wo):
        sums = torch.zeros((n, 2), dtype=torch.float64, device=x.device)
        d.gn_sums = sums.data_ptr()
    t0 = PROFILER.begin() if PROFILER is not None and (d.n_tri_steps > 0 or not PROFILER.only_triple) else None
    lib = _hip.load_library()
    _hip.check(lib.genie_conv_igemm(C.byref(d), _hip.stream_ptr()), 'genie_conv_igemm(fwd)')
    if sums is not None and (lib.genie_last_conv_gn_fused() & 1):
        gn_sums.append(sums)
    if t0 is not None:
        flops = 2.0 * n * to * ho * wo * spec.cout * spec.cin * spec.ntaps
        fexec = flops
        if d.n_tri_steps > 0:
            meta = _tri_meta.get((torch.cuda.current_device(), ('fwd', spec), h, w, pitch_of(x)))
            fexec = flops * tri_executed_fraction(meta, to, ho * wo, n * to * ho * wo, _tri_bm_of_last_launch())
        PROFILER.end(_variant('fwd', spec, spec.cout if spec.shuffle is not None else spec.cout, bool(d.small_c)),
                     f'fwd {spec.cin}->{spec.cout} k{spec.kernel} s{spec.stride} @{(t, h, w)}', flops, t0, flops_exec=fexec)
    return out


def _unshuffle(dy: Tensor, spec: ConvSpec, order: str) -> Tensor:
    """Inverse of the depth-to-space-time rearrange on a gradient: (N, cf, T P, H Q, W R) -> CL (N, cf P Q R, T, H, W) with the
    channels in natural '(c p q r)' order or in the sub-pixel-major '(p q r c)' order of the transposed weight pack.  Only the
    slow path of upsample convs whose final channel count is not a multiple of 8 (e.g. REPR_TOK_DEC's 512 -> 3 x 16) uses it."""
    from .cl import to_cl
    P, Q, R = spec.shuffle
    n, cf, tp, hq, wr = dy.shape
    t, h, w = tp // P, hq // Q, wr // R
    if order == 'pqrc' and cf % 8 == 0 and is_cl(dy):
        out = empty_cl(n, cf * P * Q * R, t, h, w, dy.device)                 # pitch == channels: cf * P * Q * R is a multiple of 8
        _hip.check(_hip.load_library().genie_unshuffle_cl(dy.data_ptr(), pitch_of(dy), out.data_ptr(), n, t, h, w, cf, P, Q, R, _hip.stream_ptr()),
                   'genie_unshuffle_cl')
        return out
    v = dy.reshape(n, cf, t, P, h, Q, w, R)
    v = v.permute(0, 1, 3, 5, 7, 2, 4, 6) if order == 'cpqr' else v.permute(0, 3, 5, 7, 1, 2, 4, 6)
    return to_cl(v.reshape(n, cf * P * Q * R, t, h, w).contiguous())


def _plain(spec: ConvSpec) -> ConvSpec:
    return ConvSpec(spec.cin, spec.cout, spec.kernel, spec.stride, spec.dilation, spec.pad_front, spec.pad_back, None)


def conv_dgrad(dy: Tensor, wpack_bwd: Tensor, spec: ConvSpec, in_size: Triple, resid: Optional[Tensor] = None,
               gnb: Optional[GnBwdFuse] = None, dy_unshuffled: Optional[Tensor] = None) -> Tensor:
    """Gradient w.r.t. the conv input.  dy is the CL gradient of the (shuffled) output.  `gnb`: see GnBwdFuse.  `dy_unshuffled`:
    ``unshuffle_dy(dy, spec)`` when the caller already made it (it shares the tensor with the weight gradient)."""
    if dy_unshuffled is not None:
        return conv_dgrad(dy_unshuffled, wpack_bwd, _plain(spec), in_size, resid)
    _check_cl(dy, spec.cfinal, 'conv_dgrad')
    if spec.shuffle is not None and (spec.cfinal % 8 != 0 or spec.ntaps * spec.shuffle[0] * spec.shuffle[1] * spec.shuffle[2] > 256
                                     or (UPCONV_DGRAD_UNSHUFFLE and spec.stride == (1, 1, 1) and spec.kernel[2] == 3
                                         and spec.cout % 64 == 0 and spec.cin >= 128)):
        # the gather through the shuffle wants whole 16-B channel chunks per sub-pixel: un-shuffle the gradient instead; the
        # transposed pack of a shuffled conv is sub-pixel-major, so the plain conv over '(p q r c)' channels is the same GEMM.
        # Also taken for the big upsample convs: one extra pass over dy buys the kw-triple kernels (1.1 - 1.3 PFLOP/s) instead of the
        # generic gather through the shuffle (0.54 - 0.94); and whenever the gather would need more than the 256 taps genie_conv_igemm's table
        # holds (taps x sub-pixels: a 3x3x3 upsample conv with space_factor 4 has 432 -- found by tests/test_gpu_random_geometry.py)
        return conv_dgrad(_unshuffle(dy, spec, 'pqrc'), wpack_bwd, _plain(spec), in_size, resid)
    n = dy.shape[0]
    t, h, w = in_size
    P, Q, R = spec.shuffle if spec.shuffle is not None else (1, 1, 1)
    dx = empty_cl(n, spec.cin, t, h, w, dy.device)
    lib = _hip.load_library()
    st = spec.stride
    first = True
    tri_ok = (TRI_BM >= 0 and st == (1, 1, 1) and spec.shuffle is None and tuple(dy.shape[2:]) == (t, h, w) and pitch_of(dy) % 64 == 0
              and spec.kernel[2] == 3)
    t0 = PROFILER.begin() if PROFILER is not None and (tri_ok or not PROFILER.only_triple) else None
    exec_frac = 1.0
    for rt in range(st[0]):
        for rh in range(st[1]):
            for rw in range(st[2]):
                ao, bo, co = -(-(t - rt) // st[0]), -(-(h - rh) // st[1]), -(-(w - rw) // st[2])
                if min(ao, bo, co) <= 0:
                    continue
                taps, ntaps, nk = dgrad_taps(spec, (rt, rh, rw), pitch_of(dy))
                if ntaps == 0:
                    dx[:, :, rt::st[0], rh::st[1], rw::st[2]] = 0 if resid is None else resid[:, :, rt::st[0], rh::st[1], rw::st[2]]
                    continue
                d = _hip.GenieConvDesc()
                d.src, d.wgt, d.dst = dy.data_ptr(), wpack_bwd.data_ptr(), dx.data_ptr()
                d.resid = _hip.ptr(resid)
                d.bias = None
                d.taps, d.ntaps, d.nk = taps.data_ptr(), ntaps, nk
                # narrow gradients (head conv: 3 -> pitch 8 channels): pack the taps back to back inside 64-wide K chunks instead of
                # giving each tap its own, mostly empty, chunk.  Needs the complete tap list in pack order: stride 1, no shuffle.
                d.small_c = 1 if (st == (1, 1, 1) and spec.shuffle is None and spec.coutp in (8, 16, 32) and ntaps <= 32
                                  and pitch_of(dy) == spec.coutp) else 0
                d.N, d.Ts, d.Hs, d.Ws, d.Cs = n, dy.shape[2], dy.shape[3], dy.shape[4], pitch_of(dy)
                d.To, d.Ho, d.Wo = ao, bo, co
                d.st, d.sh, d.sw = P, Q, R
                d.Ncols, d.w_row_stride = spec.cin, spec.ntaps * spec.coutp
                d.perm_c, d.perm_f = 0, 1
                d.Td, d.Hd, d.Wd, d.Cd = t, h, w, pitch_of(dx)
                d.dmt, d.dmh, d.dmw = st
                d.dot, d.doh, d.dow = rt, rh, rw
                d.shuf_c, d.shuf_q, d.shuf_r = spec.cin, 1, 1
                d.act = 0
                ws = _splitk_ws(dy.device)
                d.splitk_ws, d.splitk_ws_bytes = ws.data_ptr(), ws.numel() * 4
                d.tri_bm, d.tri_flags = TRI_BM, TRI_FLAGS
                d.pointwise = int(_is_pointwise(spec) and spec.coutp % 64 == 0)
                if (TRI_BM >= 0 and st == (1, 1, 1) and spec.shuffle is None and tuple(dy.shape[2:]) == (t, h, w)
                        and pitch_of(dy) % 64 == 0):
                    sched = tri_schedule(('dgrad', spec), dgrad_taps(spec, (0, 0, 0), pitch_of(dy), want_list=True), h, w, pitch_of(dy))
                    if sched is not None:
                        d.tri_steps, d.n_tri_steps = sched[0].data_ptr(), sched[1]
                want_gnb = (gnb is not None and int(GN_FUSE) >= 2 and st == (1, 1, 1) and d.n_tri_steps > 0 and _gn_rows_ok(t, h, w) and resid is None
                            and pitch_of(gnb.x) == pitch_of(dx) and tuple(gnb.x.shape) == tuple(dx.shape))
                if want_gnb:
                    gnb.nblk = t * h * w // 256
                    gnb.part = torch.empty((n, gnb.nblk, pitch_of(dx), 2), dtype=torch.float32, device=dy.device)
                    d.gnb_x, d.gnb_gamma, d.gnb_beta = gnb.x.data_ptr(), _hip.ptr(gnb.gamma), _hip.ptr(gnb.beta)
                    d.gnb_mean, d.gnb_rstd, d.gnb_part = gnb.mean.data_ptr(), gnb.rstd.data_ptr(), gnb.part.data_ptr()
                    d.gnb_act, d.gnb_nblk = gnb.act, gnb.nblk
                _hip.check(lib.genie_conv_igemm(C.byref(d), _hip.stream_ptr()), 'genie_conv_igemm(dgrad)')
                if want_gnb:
                    gnb.fused = bool(lib.genie_last_conv_gn_fused() & 2)
                if t0 is not None and d.n_tri_steps > 0:
                    exec_frac = tri_executed_fraction(_tri_meta.get((torch.cuda.current_device(), ('dgrad', spec), h, w, pitch_of(dy))), t, h * w, n * t * h * w,
                                                      _tri_bm_of_last_launch())
                first = False
    if t0 is not None:
        to, ho, wo = spec.out_size((t, h, w))
        flops = 2.0 * n * to * ho * wo * spec.cout * spec.cin * spec.ntaps
        PROFILER.end(_variant('dgrad', spec, spec.cin, False), f'dgrad {spec.cin}->{spec.cout} k{spec.kernel} s{spec.stride} @{(t, h, w)}', flops, t0,
                     flops_exec=flops * exec_frac)
    return dx


def wgrad_unshuffled_ok(spec: ConvSpec, x: Tensor) -> bool:
    """May the weight gradient of this upsample conv take the UN-SHUFFLED output gradient (the tensor `unshuffle_dy` makes for the
    backward-data pass) instead of gathering dy through the shuffle?  The preconditions of the lean kw-triple kernel (conv_wgrad3.hip)."""
    t, h, w = x.shape[2:]
    return bool(UPCONV_DGRAD_UNSHUFFLE and TRI_WGRAD and spec.shuffle is not None and spec.cfinal % 8 == 0 and spec.stride == (1, 1, 1) and spec.kernel[2] == 3
                and spec.dilation[2] == 1 and spec.pad_front[2] == 1 and spec.pad_back[2] == 1 and spec.out_size((t, h, w)) == (t, h, w)
                and w in (8, 16, 32, 64) and (h * w) % 64 == 0 and 64 // w <= h and spec.cin >= 64 and spec.cout >= 64
                and os.environ.get('GENIE_W3_LEAN', '1') != '0')


def unshuffle_dy(dy: Tensor, spec: ConvSpec) -> Tensor:
    """The output gradient of an upsample conv brought back onto the conv's own row grid, sub-pixel-major channels '(p q r c)': what both
    backward passes of the conv consume (conv_dgrad(..., dy_unshuffled=...) / conv_wgrad(..., dy_unshuffled=True))."""
    return _unshuffle(dy, spec, 'pqrc')


# 128-pixel-wide layers (BASELINE configs[4]: the LatentAction ST blocks' 3x3x3 feed-forward convs at 128 x 128): the lean kw-triple weight-gradient
# kernel stages whole 64-pixel chunks with zero columns left and right, i.e. W <= 64.  A 128-wide image is TWO 64-column windows of the same memory
# (GenieWgradDesc.row_px / px0: a chunk is then half a memory row) plus the two products the windows' zero edges leave out -- at the seam, dy column 63
# meets x column 64 through the kw = +1 tap and dy column 64 meets x column 63 through kw = -1: two (kt, kh, 1) weight gradients over one-pixel-wide
# copies of those columns (1 / 64 of the work, generic kernel).  GENIE_WGRAD_WINDOWS=0: the whole layer on the generic kernel (rounds 1-5; A/B, tests).
WGRAD_WINDOWS = os.environ.get('GENIE_WGRAD_WINDOWS', '1') != '0'


def wgrad_windows_ok(spec: ConvSpec, x: Tensor, dy: Tensor, dy_unshuffled: bool = False) -> bool:
    n, _, t, h, w = x.shape
    return bool(WGRAD_WINDOWS and TRI_WGRAD and w == 128 and not dy_unshuffled and spec.shuffle is None and spec.stride == (1, 1, 1) and spec.kernel[2] == 3
                and spec.dilation[2] == 1 and spec.pad_front[2] == 1 and spec.pad_back[2] == 1 and tuple(dy.shape[2:]) == (t, h, w)
                and spec.cin >= 64 and spec.cout >= 64
                and (n * t * h + h + 2) * w * max(pitch_of(x), pitch_of(dy)) * 2 < 0x7f000000)        # a block's buffer range (the kernel's own 2-GiB rule)


def _conv_wgrad_windows(x: Tensor, dy: Tensor, spec: ConvSpec, dweight: Tensor, dbias: Optional[Tensor]) -> None:
    from .cl import to_cl
    n, _, t, h, w = x.shape
    s = dweight.stride()
    kt, kh, kw = spec.kernel
    if not (s[3] == kw * s[4] and s[2] == kh * s[3]):
        raise ValueError('conv_wgrad: weight-gradient taps must be flattenable (contiguous or channels_last_3d)')
    taps, ntaps, _ = fwd_taps(spec)
    t0 = PROFILER.begin() if PROFILER is not None and not PROFILER.only_triple else None
    for px0 in (0, 64):
        d = _hip.GenieWgradDesc()
        d.src, d.dy, d.dw, d.dbias, d.taps, d.ntaps = x.data_ptr(), dy.data_ptr(), dweight.data_ptr(), _hip.ptr(dbias), taps.data_ptr(), ntaps
        d.N, d.Ts, d.Hs, d.Ws, d.Cs, d.Cin = n, t, h, 64, pitch_of(x), spec.cin
        d.To, d.Ho, d.Wo = t, h, 64
        d.st = d.sh = d.sw = 1
        d.Td, d.Hd, d.Wd, d.Cd, d.Cout = t, h, 64, pitch_of(dy), spec.cout
        d.dmt = d.dmh = d.dmw = 1
        d.dot = d.doh = d.dow = 0
        d.shuf_c, d.shuf_q, d.shuf_r = spec.cout, 1, 1
        d.s_cout, d.s_tap, d.s_cin = s[0], s[4], s[1]
        d.split_k = 1 if DETERMINISTIC else FORCE_SPLIT_K
        d.tri_mode, d.pointwise, d.dy_unshuffled = 2, 0, 0
        d.row_px, d.px0 = w, px0
        _hip.check(_hip.load_library().genie_conv_wgrad(C.byref(d), _hip.stream_ptr()), 'genie_conv_wgrad (W-window)')
    if t0 is not None:
        PROFILER.end(_variant('wgrad', spec, 0, False), f'wgrad {spec.cin}->{spec.cout} k{spec.kernel} s{spec.stride} @{(t, h, w)} (2 windows)',
                     2.0 * n * t * h * w * spec.cout * spec.cin * spec.ntaps, t0)
    seam = ConvSpec(spec.cin, spec.cout, (kt, kh, 1), (1, 1, 1), (spec.dilation[0], spec.dilation[1], 1), (spec.pad_front[0], spec.pad_front[1], 0),
                    (spec.pad_back[0], spec.pad_back[1], 0), None)
    for xc, dc, k in ((64, 63, 2), (63, 64, 0)):
        dwv = dweight[:, :, :, :, k:k + 1]
        dwv = dwv.as_strided(dwv.shape, (s[0], s[1], s[2], s[3], s[3]))      # (the size-1 axis takes the stride that makes the taps flattenable)
        conv_wgrad(to_cl(x[:, :, :, :, xc:xc + 1]), to_cl(dy[:, :, :, :, dc:dc + 1]), seam, dwv, None)       # one-pixel-wide dense copies


def _bias_grad_fixed_order(dy: Tensor, spec: ConvSpec, dy_unshuffled: bool) -> Tensor:
    """fp32 [cout] column sums of `dy` in the natural order of the bias, reduced in a fixed tree (torch's sum uses no atomics).  For an
    upsample conv the bias sits before the depth-to-space-time rearrange: channel ``c * PQR + f`` of the bias collects final channel c at
    sub-pixel f; an unshuffled `dy` holds the same sums in '(p q r c)' order."""
    if spec.shuffle is None:
        return dy[:, :spec.cout].sum((0, 2, 3, 4), dtype=torch.float32)
    P, Q, R = spec.shuffle
    f = P * Q * R
    if dy_unshuffled:
        return dy[:, :spec.cout].sum((0, 2, 3, 4), dtype=torch.float32).view(f, spec.cfinal).t().reshape(-1)
    n, _, tp, hq, wr = dy.shape
    v = dy[:, :spec.cfinal].reshape(n, spec.cfinal, tp // P, P, hq // Q, Q, wr // R, R)
    return v.sum((0, 2, 4, 6), dtype=torch.float32).reshape(-1)


def conv_wgrad(x: Tensor, dy: Tensor, spec: ConvSpec, dweight: Tensor, dbias: Optional[Tensor], dy_unshuffled: bool = False) -> None:
    """Accumulate dW (fp32, any strides, shape (cout, cin, kt, kh, kw)) and dbias (fp32 [cout]).  `dy_unshuffled`: `dy` is
    ``unshuffle_dy(dy, spec)`` of an upsample conv (only where ``wgrad_unshuffled_ok``)."""
    _check_cl(x, spec.cin, 'conv_wgrad(x)')
    _check_cl(dy, spec.cout if dy_unshuffled else spec.cfinal, 'conv_wgrad(dy)')
    if spec.shuffle is not None and spec.cfinal % 8 != 0:
        return conv_wgrad(x, _unshuffle(dy, spec, 'cpqr'), _plain(spec), dweight, dbias)
    n, _, t, h, w = x.shape
    to, ho, wo = spec.out_size((t, h, w))
    P, Q, R = spec.shuffle if spec.shuffle is not None and not dy_unshuffled else (1, 1, 1)
    assert tuple(dy.shape[2:]) == (to * P, ho * Q, wo * R)
    assert dweight.dtype == torch.float32 and tuple(dweight.shape) == (spec.cout, spec.cin, *spec.kernel)
    if DETERMINISTIC and dbias is not None:
        # The kernels' epilogues add per-wave column sums of dy into dbias with atomics; several waves of a workgroup (and every workgroup
        # of a column) meet on one address, so the ORDER of those adds -- and the last ulp of the sum -- changes from run to run even with
        # one K split.  Deterministic mode sums the bias gradient in a fixed tree instead and hands the kernel no bias pointer.
        dbias.add_(_bias_grad_fixed_order(dy, spec, dy_unshuffled))
        dbias = None
    if narrow_wgrad_ok(spec, x, dy):
        return conv_narrow_wgrad(x, dy, spec, dweight, dbias, f'wgrad {spec.cin}->{spec.cout} k3 @{(t, h, w)}')
    if wgrad_windows_ok(spec, x, dy, dy_unshuffled):
        return _conv_wgrad_windows(x, dy, spec, dweight, dbias)
    s = dweight.stride()
    kt, kh, kw = spec.kernel
    if kt * kh * kw > 1 and not (s[3] == kw * s[4] and s[2] == kh * s[3]):
        raise ValueError('conv_wgrad: weight-gradient taps must be flattenable (contiguous or channels_last_3d)')
    taps, ntaps, _ = fwd_taps(spec)
    d = _hip.GenieWgradDesc()
    d.src, d.dy, d.dw, d.dbias, d.taps, d.ntaps = x.data_ptr(), dy.data_ptr(), dweight.data_ptr(), _hip.ptr(dbias), taps.data_ptr(), ntaps
    d.N, d.Ts, d.Hs, d.Ws, d.Cs, d.Cin = n, t, h, w, pitch_of(x), spec.cin
    d.To, d.Ho, d.Wo = to, ho, wo
    d.st, d.sh, d.sw = spec.stride
    d.Td, d.Hd, d.Wd, d.Cd, d.Cout = dy.shape[2], dy.shape[3], dy.shape[4], pitch_of(dy), spec.cout
    d.dmt, d.dmh, d.dmw = P, Q, R
    d.dot = d.doh = d.dow = 0
    if spec.shuffle is not None:
        d.shuf_c, d.shuf_q, d.shuf_r = spec.cfinal, spec.shuffle[1], spec.shuffle[2]
    else:
        d.shuf_c, d.shuf_q, d.shuf_r = spec.cout, 1, 1
    d.dy_unshuffled = 1 if dy_unshuffled else 0
    d.s_cout, d.s_tap, d.s_cin = s[0], s[4], s[1]
    d.split_k = 1 if DETERMINISTIC else FORCE_SPLIT_K
    d.tri_mode = (2 if dy_unshuffled else TRI_WGRAD) if (TRI_WGRAD and spec.stride == (1, 1, 1) and spec.kernel[2] == 3 and spec.dilation[2] == 1
                       and spec.pad_front[2] == 1 and spec.pad_back[2] == 1 and (to, ho, wo) == (t, h, w)) else 0
    d.pointwise = WGRAD_PW if _is_pointwise(spec) else 0
    t0 = PROFILER.begin() if PROFILER is not None and not PROFILER.only_triple else None
    _hip.check(_hip.load_library().genie_conv_wgrad(C.byref(d), _hip.stream_ptr()), 'genie_conv_wgrad')
    if t0 is not None:
        flops = 2.0 * n * to * ho * wo * spec.cout * spec.cin * spec.ntaps
        PROFILER.end(_variant('wgrad', spec, 0, False), f'wgrad {spec.cin}->{spec.cout} k{spec.kernel} s{spec.stride} @{(t, h, w)}', flops, t0)
