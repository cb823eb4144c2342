"""ctypes binding of ``libgenie_hip.so`` (C ABI declared in ``include/genie_hip.h``).

There is NO fallback: if the library is missing or a call fails this module raises.  A GPU box that
silently computed on some other path would void every parity claim ("boundary").
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('GENIE_HIP_LIB', os.path.join(os.path.dirname(_HERE), 'lib', 'libgenie_hip.so'))

GENIE_F32, GENIE_BF16 = 0, 1
ABI_VERSION = 13


class GenieTap(C.Structure):
    _fields_ = [('dt', C.c_int32), ('dh', C.c_int32), ('dw', C.c_int32), ('wofs', C.c_int32),
                ('c0', C.c_int32), ('nch', C.c_int32), ('pad0', C.c_int32), ('pad1', C.c_int32)]


class GenieConvDesc(C.Structure):
    _fields_ = [('src', C.c_void_p), ('wgt', C.c_void_p), ('dst', C.c_void_p), ('resid', C.c_void_p),
                ('bias', C.c_void_p), ('taps', C.c_void_p),
                ('ntaps', C.c_int32), ('nk', C.c_int32), ('small_c', C.c_int32),
                ('N', C.c_int32), ('Ts', C.c_int32), ('Hs', C.c_int32), ('Ws', C.c_int32), ('Cs', C.c_int32),
                ('To', C.c_int32), ('Ho', C.c_int32), ('Wo', C.c_int32),
                ('st', C.c_int32), ('sh', C.c_int32), ('sw', C.c_int32),
                ('Ncols', C.c_int32), ('w_row_stride', C.c_int32), ('perm_c', C.c_int32), ('perm_f', C.c_int32),
                ('Td', C.c_int32), ('Hd', C.c_int32), ('Wd', C.c_int32), ('Cd', C.c_int32),
                ('dmt', C.c_int32), ('dmh', C.c_int32), ('dmw', C.c_int32),
                ('dot', C.c_int32), ('doh', C.c_int32), ('dow', C.c_int32),
                ('shuf_c', C.c_int32), ('shuf_q', C.c_int32), ('shuf_r', C.c_int32), ('act', C.c_int32),
                ('splitk_ws', C.c_void_p), ('splitk_ws_bytes', C.c_int64),
                ('tri_steps', C.c_void_p), ('n_tri_steps', C.c_int32), ('tri_bm', C.c_int32), ('tri_flags', C.c_int32),
                ('pointwise', C.c_int32),
                ('gn_sums', C.c_void_p), ('gnb_x', C.c_void_p), ('gnb_gamma', C.c_void_p), ('gnb_beta', C.c_void_p), ('gnb_mean', C.c_void_p),
                ('gnb_rstd', C.c_void_p), ('gnb_part', C.c_void_p), ('gnb_act', C.c_int32), ('gnb_nblk', C.c_int32)]


class GenieWgradDesc(C.Structure):
    _fields_ = [('src', C.c_void_p), ('dy', C.c_void_p), ('dw', C.c_void_p), ('dbias', C.c_void_p), ('taps', C.c_void_p),
                ('ntaps', C.c_int32),
                ('N', C.c_int32), ('Ts', C.c_int32), ('Hs', C.c_int32), ('Ws', C.c_int32), ('Cs', C.c_int32), ('Cin', C.c_int32),
                ('To', C.c_int32), ('Ho', C.c_int32), ('Wo', C.c_int32), ('st', C.c_int32), ('sh', C.c_int32), ('sw', C.c_int32),
                ('Td', C.c_int32), ('Hd', C.c_int32), ('Wd', C.c_int32), ('Cd', C.c_int32), ('Cout', C.c_int32),
                ('dmt', C.c_int32), ('dmh', C.c_int32), ('dmw', C.c_int32),
                ('dot', C.c_int32), ('doh', C.c_int32), ('dow', C.c_int32),
                ('shuf_c', C.c_int32), ('shuf_q', C.c_int32), ('shuf_r', C.c_int32),
                ('s_cout', C.c_int64), ('s_tap', C.c_int64), ('s_cin', C.c_int64), ('split_k', C.c_int32), ('tri_mode', C.c_int32),
                ('pointwise', C.c_int32), ('dy_unshuffled', C.c_int32), ('row_px', C.c_int32), ('px0', C.c_int32)]


class GeniePackJob(C.Structure):
    _fields_ = [('src_off', C.c_int64), ('dst_off', C.c_int64), ('R', C.c_int32), ('J', C.c_int32), ('K', C.c_int32),
                ('perm_c', C.c_int32), ('perm_f', C.c_int32), ('tiles_r', C.c_int32), ('tiles_k', C.c_int32), ('first_block', C.c_int32)]


_lib: Optional[C.CDLL] = None

# name -> (restype, argtypes); the single source for the symbol-export test as well
_P, _I, _L, _F = C.c_void_p, C.c_int, C.c_int64, C.c_float
_PL = C.POINTER(C.c_int64)
SIGNATURES = {
    'genie_abi_version': (C.c_int, []),
    'genie_last_error': (C.c_char_p, []),
    'genie_to_channels_last': (C.c_int, [_P, _I, _PL, _PL, _P, _I, _P]),
    'genie_unshuffle_cl': (C.c_int, [_P, _I, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    'genie_from_channels_last': (C.c_int, [_P, _I, _PL, _P, _I, _PL, _P]),
    'genie_conv_igemm': (C.c_int, [C.POINTER(GenieConvDesc), _P]),
    'genie_conv_narrow_in': (C.c_int, [_P, _I, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    'genie_conv_narrow_out': (C.c_int, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    'genie_conv_narrow_wgrad': (C.c_int, [_P, _P, _I, _P, _I, _I, _I, _I, _I, _I, _P]),
    'genie_conv_narrow_wgrad_acc': (C.c_int, [_P, _P, _I, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    'genie_conv_narrow_wgrad_wide': (C.c_int, [_P, _I, _I, _I, _P, _I, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    'genie_conv_wgrad': (C.c_int, [C.POINTER(GenieWgradDesc), _P]),
    'genie_last_conv_variant': (C.c_int, []),
    'genie_pack_weight': (C.c_int, [_P, _P, _I, _I, _I, _L, _L, _L, _I, _I, _P]),
    'genie_cast_f32_to_bf16': (C.c_int, [_P, _P, _L, _P]),
    'genie_groupnorm_ws_floats': (C.c_int64, [_I, _I, _I]),
    'genie_groupnorm_fwd': (C.c_int, [_P, _P, _I, _L, _I, _I, _I, _P, _P, _P, _P, _F, _I, _P, _P, _P, _P]),
    'genie_groupnorm_bwd': (C.c_int, [_P, _P, _P, _I, _L, _I, _I, _I, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _P, _P, _P]),
    'genie_groupnorm_fwd_from_sums': (C.c_int, [_P, _P, _I, _L, _I, _I, _P, _P, _F, _I, _P, _P, _P, _P]),
    'genie_groupnorm_bwd_from_part_ws_floats': (C.c_int64, [_I]),
    'genie_groupnorm_bwd_from_part': (C.c_int, [_P, _P, _P, _I, _L, _I, _I, _P, _P, _I, _P, _P, _P, _P, _P, _I, _P, _P]),
    'genie_last_conv_gn_fused': (C.c_int, []),
    'genie_blur_pool3d_fwd': (C.c_int, [_P, _I, _PL, _P, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), _P, _I, _I, _P, _P]),
    'genie_blur_pool3d_bwd': (C.c_int, [_P, _I, _I, _PL, _P, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), _P, _I, _P, _P]),
    'genie_silu_fwd': (C.c_int, [_P, _P, _L, _P]),
    'genie_silu_bwd': (C.c_int, [_P, _P, _P, _L, _P]),
    'genie_gelu_fwd': (C.c_int, [_P, _P, _L, _P]),
    'genie_gelu_bwd': (C.c_int, [_P, _P, _P, _L, _P]),
    'genie_leaky_relu_fwd': (C.c_int, [_P, _P, _L, _F, _P]),
    'genie_leaky_relu_bwd': (C.c_int, [_P, _P, _P, _L, _F, _P]),
    'genie_add': (C.c_int, [_P, _P, _P, _L, _P]),
    'genie_lfq_quantize': (C.c_int, [_P, _I, _L, _I, _I, _L, _P, _P, _P]),
    'genie_lfq_loss_ws_floats': (C.c_int64, [_L, _I, _I]),
    'genie_lfq_loss': (C.c_int, [_P, _I, _L, _I, _I, _L, _F, _F, _F, _F, _P, _P, _P, _P]),
    'genie_lfq_bwd': (C.c_int, [_P, _P, _P, _P, _I, _L, _I, _L, _P]),
    'genie_masked_ce_fwd': (C.c_int, [_P, _L, _L, _I, _P, _P, _P, _P, _P]),
    'genie_masked_ce_bwd': (C.c_int, [_P, _L, _L, _I, _P, _P, _P, _P, _P, _L, _P]),
    'genie_linear_ce_supported': (C.c_int, [_L, _I, _L, _L, _L]),
    'genie_linear_ce_ws_floats': (C.c_int64, [_L, _I, _L, _I]),
    'genie_linear_ce_fwd': (C.c_int, [_P, _L, _L, _I, _P, _L, _L, _P, _P, _P, _P, _L, _P, _P, _P, _P, _P]),
    'genie_linear_ce_bwd': (C.c_int, [_P, _L, _L, _I, _P, _L, _L, _P, _P, _P, _P, _P, _P, _L, _P, _P, _P]),
    'genie_u8_frames_to_cl': (C.c_int, [_P, _L, _I, _P, _I, _P]),
    'genie_embedding_fwd': (C.c_int, [_P, _P, _P, _L, _I, _L, _P]),
    'genie_linear_small_ws_floats': (C.c_int64, [_L, _I, _I]),
    'genie_linear_small_fwd': (C.c_int, [_P, _I, _L, _L, _I, _P, _L, _L, _P, _P, _I, _L, _I, _P, _L, _P]),
    'genie_linear_small_wgrad_ws_floats': (C.c_int64, [_L, _I, _I]),
    'genie_linear_small_wgrad': (C.c_int, [_P, _I, _L, _P, _I, _L, _L, _I, _I, _P, _L, _L, _P, _P, _L, _P]),
    'genie_embedding_bwd': (C.c_int, [_P, _P, _I, _P, _L, _I, _L, _P]),
    'genie_guard_alloc': (C.c_int, [_L, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]),
    'genie_guard_free': (C.c_int, [_P]),
    'genie_maskgit_sample': (C.c_int, [_P, _I, _L, _L, _L, _L, _L, _P, _F, _P, _P, _P]),
    'genie_maskgit_paint': (C.c_int, [_P, _P, _L, _L, _L, _P, _P, _P]),
    'genie_mse_fwd': (C.c_int, [_P, _I, _P, _I, _PL, _PL, _P, _P, _P]),
    'genie_mse_bwd': (C.c_int, [_P, _I, _P, _I, _PL, _PL, _P, _P, _P]),
    'genie_adamw_step': (C.c_int, [_P, _P, _P, _P, _L, _F, _F, _F, _F, _F, _I, _F, _I, _P]),
    'genie_adamw_step_mirror': (C.c_int, [_P, _P, _P, _P, _P, _L, _F, _F, _F, _F, _F, _I, _F, _I, _P]),
    'genie_gn_fused_error': (C.c_int, []),
    'genie_adamw_step_graph': (C.c_int, [_P, _P, _P, _P, _P, _L, _P, _F, _F, _F, _F, _I, _P]),
    'genie_pack_transpose_batched': (C.c_int, [_P, _I, _I, _P, _P, _P]),
    'genie_rotary_layernorm_fwd': (C.c_int, [_P, _P, _L, _I, _L, _P, _L, _I, _P, _P, _F, _P, _P]),
    'genie_rotary_layernorm_bwd': (C.c_int, [_P, _P, _P, _P, _L, _I, _L, _P, _L, _I, _P, _P, _P, _P, _P]),
    'genie_attention_fwd': (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _PL, _PL, _PL, _F, _I, _I, _P]),
    'genie_attention_bwd_cond': (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _PL, _PL, _PL, _F, _I, _I, _I, _L, _P]),
    'genie_attention_bwd': (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _PL, _PL, _PL, _PL, _F, _I, _I, _L, _P]),
    # genie_attention_fwd / _bwd + (float dropout_p, uint64 seed) before the stream
    'genie_attention_fwd_dropout': (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _PL, _PL, _PL, _F, _I, _I, _F, C.c_uint64, _P]),
    'genie_attention_bwd_dropout': (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _PL, _PL, _PL, _PL, _F, _I, _I, _L, _F, C.c_uint64, _P]),
    'genie_attention_dropout_mask': (C.c_int, [_P, _I, _I, _I, _I, _F, C.c_uint64, _P]),
    'genie_attention_lean_mode': (C.c_int, [_I]),
    'genie_attention_lean_occupancy': (C.c_int, [_I]),
    'genie_probe_ds_read_tr16': (C.c_int, [_P, _P, _P, _P]),
}


def load_library() -> C.CDLL:
    """Load libgenie_hip.so; raises (never falls back) when it is missing or stale."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f'genie: HIP library not found at {LIB_PATH}. Build it with `make -C open-genie_amd` '
            f'(or `python -c "import __graft_entry__ as g; g.build()"`). There is no CPU/PyTorch fallback.')
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError -> missing export: fail loudly
        fn.restype, fn.argtypes = res, args
    if lib.genie_abi_version() != ABI_VERSION:
        raise RuntimeError(f'genie: {LIB_PATH} has ABI {lib.genie_abi_version()}, expected {ABI_VERSION}; rebuild')
    if os.environ.get('GENIE_ROCTX', '0') not in ('0', ''):
        lib = _RoctxProxy(lib)
    _lib = lib
    return lib


class _RoctxProxy:
    """GENIE_ROCTX=1: every C-ABI call that enqueues work is bracketed by a roctx range named after the entry point (SURVEY.md section 5,
    "tracing"): `rocprofv3 --marker-trace --kernel-trace` then shows which operator of the reference's module tree a kernel belongs to.
    Off by default -- two extra foreign calls per launch.  Queries (version, error string, last variant, switches) pass straight through."""

    _PLAIN = ('genie_abi_version', 'genie_last_error', 'genie_last_conv_variant', 'genie_last_conv_gn_fused', 'genie_attention_lean_mode',
              'genie_attention_lean_occupancy')

    def __init__(self, lib: C.CDLL) -> None:
        self._lib = lib
        rt = None
        for name in ('libroctx64.so', 'librocprofiler-sdk-roctx.so', '/opt/rocm/lib/libroctx64.so'):
            try:
                rt = C.CDLL(name)
                break
            except OSError:
                continue
        if rt is None:
            raise RuntimeError('GENIE_ROCTX=1 but no roctx library (libroctx64.so / librocprofiler-sdk-roctx.so) could be loaded')
        rt.roctxRangePushA.argtypes, rt.roctxRangePushA.restype = [C.c_char_p], C.c_int
        rt.roctxRangePop.argtypes, rt.roctxRangePop.restype = [], C.c_int
        self._push, self._pop = rt.roctxRangePushA, rt.roctxRangePop
        self._wrapped = {}

    def __getattr__(self, name: str):
        fn = getattr(self._lib, name)
        if name in self._PLAIN or not name.startswith('genie_'):
            return fn
        w = self._wrapped.get(name)
        if w is None:
            label, push, pop = name.encode(), self._push, self._pop

            def w(*args, _fn=fn):
                push(label)
                try:
                    return _fn(*args)
                finally:
                    pop()
            self._wrapped[name] = w
        return w


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load_library().genie_last_error().decode('utf-8', 'replace')
        raise RuntimeError(f'{what} failed (code {rc}): {msg}')


def stream_ptr() -> int:
    return torch.cuda.current_stream().cuda_stream


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def i32(vals) -> C.Array:
    return (C.c_int * len(vals))(*[int(v) for v in vals])


def i64(vals) -> C.Array:
    return (C.c_int64 * len(vals))(*[int(v) for v in vals])


def require_gpu(t: torch.Tensor, what: str) -> None:
    if not t.is_cuda:
        raise RuntimeError(f'{what}: tensor is on {t.device}; the genie hot path runs on the MI355X only '
                           f'(there is no CPU fallback -- use oracle/ for CPU reference results in tests)')
