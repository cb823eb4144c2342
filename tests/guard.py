"""Guard-page allocation harness for the kernel-level GPU tests ("memory safety is tested by luck").

Inside ``AllocationGuard()`` every CUDA tensor the Python side creates -- the tests' own inputs (``.cuda()``, ``.to('cuda')``,
``torch.randn(..., device='cuda')``, ``torch.tensor``, ...) and what the genie wrappers allocate for outputs, workspaces and gradients
(``torch.empty`` / ``zeros`` / ``empty_like`` / ``new_empty`` ...) -- lives in its own ``genie_guard_alloc`` mapping (csrc/guard.hip): the
tensor's last addressable byte is the last byte of the mapping (up to 15 bytes of 16-byte alignment slack), with an UNMAPPED page in front
of and behind it.  A kernel that reads or writes one element outside an operand takes a GPU page fault ("Memory access fault by GPU
node ...", the process aborts) on EVERY run -- instead of only when the caching allocator happened to leave a hole behind the tensor, which
is how round 4's chan_sum_kernel bug stayed hidden in one file order and crashed in another.

Not covered: tensors that torch's C++ side allocates for the results of its own operators (``a + b``, ``x[mask]``, ``torch.cat`` ...).

Use:  python scripts/guard_run.py tests/test_gpu_kernels.py tests/test_gpu_attention.py ...   (pytest in-process under the guard)"""
import ctypes as C
import math

import torch


class _Block:
    """One guard-page mapping, exposed to torch through __cuda_array_interface__ (torch keeps this object alive as long as the tensor)."""
    pending = []                                          # handles of dead blocks: unmapped only after a device synchronise

    def __init__(self, nbytes: int) -> None:
        from genie import _hip
        lib = _hip.load_library()
        p, h = C.c_void_p(), C.c_void_p()
        _hip.check(lib.genie_guard_alloc(int(nbytes), C.byref(p), C.byref(h)), 'genie_guard_alloc')
        self.ptr, self.handle, self.nbytes = p.value, h.value, max(int(nbytes), 1)
        self.__cuda_array_interface__ = {'shape': (self.nbytes,), 'typestr': '|u1', 'data': (self.ptr, False), 'version': 2}

    def __del__(self):
        _Block.pending.append(self.handle)


# Dead blocks are NOT unmapped while tests run unless the budget below is exceeded: un-mapping and re-reserving virtual ranges in the middle of a
# run made kernels read stale pages now and then (the first drain after 512 dead blocks was followed by a wrong conv result in two of two
# runs; without drains: clean) -- a mapping, once made, stays until the process ends or GENIE_GUARD_MAX_GB (default 160) is exceeded.
_LIVE_BYTES = [0]


def _drain(force: bool = False) -> None:
    import os
    budget = float(os.environ.get('GENIE_GUARD_MAX_GB', '160')) * 2 ** 30
    if _Block.pending and (force or STATS['bytes'] - _LIVE_BYTES[0] > budget):
        _LIVE_BYTES[0] = STATS['bytes']
        from genie import _hip
        lib = _hip.load_library()
        torch.cuda.synchronize()                          # kernels still reading a dead tensor must finish before its pages go away
        hs, _Block.pending = _Block.pending, []
        for h in hs:
            lib.genie_guard_free(h)


_ORIG = {}
_ACTIVE = False
STATS = {'blocks': 0, 'bytes': 0}


def guard_storage(nbytes: int) -> torch.Tensor:
    """uint8 tensor of `nbytes` bytes in its own guard-page mapping."""
    global _ACTIVE
    _drain()
    was, _ACTIVE = _ACTIVE, False                         # torch.as_tensor itself must not be re-homed
    try:
        t = _ORIG.get('as_tensor', torch.as_tensor)(_Block(nbytes), device='cuda')
    finally:
        _ACTIVE = was
    STATS['blocks'] += 1
    STATS['bytes'] += int(nbytes)
    return t


def _span(t: torch.Tensor) -> int:
    """Elements from the first to the last addressable element of t (1 + sum (size - 1) * |stride|); 0 for an empty tensor."""
    if t.numel() == 0:
        return 0
    return 1 + sum((s - 1) * abs(st) for s, st in zip(t.shape, t.stride()))


def rehome(t: torch.Tensor) -> torch.Tensor:
    """A copy of the CUDA tensor `t` with the same shape / strides / dtype whose last addressable element ends its guard mapping.
    Tensors that take part in autograd, are not on the GPU, or have an exotic dtype are returned as they are."""
    if not isinstance(t, torch.Tensor) or not t.is_cuda or t.requires_grad or t.grad_fn is not None or t.is_sparse or t.dtype.is_complex:
        return t
    if any(st < 0 for st in t.stride()):
        return t
    span = _span(t)
    es = t.element_size()
    if span == 0:
        return t
    raw = guard_storage(span * es)                        # starts 16-byte aligned; ends at most 15 bytes before the unmapped page
    out = raw.view(t.dtype).as_strided(t.shape, t.stride())
    global _ACTIVE
    was, _ACTIVE = _ACTIVE, False
    try:
        out.copy_(t)
    finally:
        _ACTIVE = was
    return out


_FACTORIES = ['empty', 'zeros', 'ones', 'full', 'rand', 'randn', 'randint', 'arange', 'tensor', 'as_tensor', 'empty_like', 'zeros_like', 'ones_like',
              'full_like', 'rand_like', 'randn_like', 'empty_strided', 'cat', 'stack']
_METHODS = ['cuda', 'to', 'clone', 'contiguous', 'new_empty', 'new_zeros', 'new_full', 'new_ones', 'float', 'bfloat16', 'long', 'int', 'half']


def _wrap(fn):
    def wrapped(*a, **k):
        out = fn(*a, **k)
        if _ACTIVE and isinstance(out, torch.Tensor) and out.is_cuda and not (a and out is a[0]):
            return rehome(out)
        return out
    wrapped.__wrapped__ = fn
    return wrapped


class AllocationGuard:
    def __enter__(self):
        global _ACTIVE
        for n in _FACTORIES:
            _ORIG[n] = getattr(torch, n)
            setattr(torch, n, _wrap(_ORIG[n]))
        for n in _METHODS:
            _ORIG['Tensor.' + n] = getattr(torch.Tensor, n)
            setattr(torch.Tensor, n, _wrap(_ORIG['Tensor.' + n]))
        _ACTIVE = True
        return self

    def __exit__(self, *exc):
        global _ACTIVE
        _ACTIVE = False
        for n in _FACTORIES:
            setattr(torch, n, _ORIG[n])
        for n in _METHODS:
            setattr(torch.Tensor, n, _ORIG['Tensor.' + n])
        _drain(force=True)
        return False
