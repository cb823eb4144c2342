"""Training runtime for the hot path: one flat parameter arena, fused AdamW, clip-sharded data parallel.

The reference trains through Lightning (``Trainer.fit`` -> DDP wrapper -> ``training_step`` -> AdamW,
reference genie/tokenizer.py:391-442, config/tokenize.yaml:49-53,75-77).  Lightning is optional here; this
module is the MI355X-first equivalent of that loop:

* ``ParamArena`` re-homes every trainable parameter (and its ``.grad``) as a view into ONE fp32 buffer each.
  The wgrad kernels accumulate straight into the gradient arena, the optimiser is a single fused kernel
  over the arena (``genie_adamw_step``: update + "consume and clear" of the gradients), and data-parallel
  reduction is a handful of large RCCL all-reduces over contiguous arena ranges.
* ``DataParallel`` = one process per GPU, clips sharded by rank, gradient all-reduce (mean) over xGMI in
  a few big buckets issued on a side stream as soon as backward has passed the bucket's first layer, so
  the reduction of the decoder's gradients overlaps the encoder's backward.  The logged scalars of a step
  are reduced in ONE small all-reduce (the reference's ``log_dict(sync_dist=True)`` sends six).
"""
from __future__ import annotations

import contextlib
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist
import torch.nn as nn
from torch import Tensor

from . import _hip

_ALIGN = 64   # elements (256 B)


def _is_dense(shape, stride) -> bool:
    """True if (shape, stride) enumerates each of prod(shape) storage slots exactly once."""
    dims = sorted(((st, sz) for sz, st in zip(shape, stride) if sz > 1), key=lambda t: t[0])
    expect = 1
    for st, sz in dims:
        if st != expect:
            return False
        expect *= sz
    return True


def execution_order(module: nn.Module, order: Optional[Sequence[nn.Module]] = None) -> List[Tuple[str, nn.Parameter]]:
    """Trainable (name, parameter) pairs of `module` in FORWARD-EXECUTION order.

    ``named_parameters()`` is registration order, which is not execution order: VideoTokenizer registers ``quant`` after
    ``dec_layers`` although it runs between encoder and decoder, DynamicsModel registers ``head`` / ``tok_emb`` / ``act_emb``
    after ``dec_layers`` although the embeddings run first.  The data-parallel buckets (DataParallel.install_overlap_hooks) are
    contiguous arena ranges that are reduced as soon as backward has passed them, which is only correct if "later in the arena"
    means "later in forward".  `order` (default: ``module.forward_order()`` when the model defines it) lists sub-modules in the
    order they execute; parameters outside every listed sub-module come FIRST (they land in the bucket that is reduced last, by
    ``DataParallel.finish()``, when every gradient is known to be complete)."""
    if order is None and hasattr(module, 'forward_order'):
        order = module.forward_order()
    named = [(n, p) for n, p in module.named_parameters() if p.requires_grad]
    if not order:
        return named
    rank = {}
    for i, sub in enumerate(order):
        for p in sub.parameters():
            rank.setdefault(id(p), i)
    return sorted(named, key=lambda np_: rank.get(id(np_[1]), -1))        # stable: registration order inside one sub-module


class ParamArena:
    def __init__(self, module: nn.Module, device=None, order: Optional[Sequence[nn.Module]] = None) -> None:
        named = execution_order(module, order)
        if not named:
            raise ValueError('ParamArena: module has no trainable parameters')
        device = device if device is not None else named[0][1].device
        offs, total = [], 0
        for _, p in named:
            offs.append(total)
            total += (p.numel() + _ALIGN - 1) // _ALIGN * _ALIGN
        self.numel = total
        self.params = torch.zeros(total, dtype=torch.float32, device=device)
        self.grads = torch.zeros(total, dtype=torch.float32, device=device)
        self.exp_avg = torch.zeros(total, dtype=torch.float32, device=device)
        self.exp_avg_sq = torch.zeros(total, dtype=torch.float32, device=device)
        self.slots: Dict[str, Tuple[int, int]] = {}
        self.order_names = [n for n, _ in named]      # layout order
        self.order_is_execution = bool(order) or hasattr(module, 'forward_order')
        self._plist: List[nn.Parameter] = []
        with torch.no_grad():
            for (name, p), off in zip(named, offs):
                n = p.numel()
                if p.dtype != torch.float32:
                    raise TypeError(f'ParamArena: parameter {name} is {p.dtype}; master parameters are fp32')
                shape, stride = tuple(p.shape), tuple(p.stride())
                if not _is_dense(shape, stride):
                    stride = tuple(torch.empty(shape).stride())
                view = self.params[off:off + n].as_strided(shape, stride)
                view.copy_(p.detach().to(device))
                p.data = view
                p.grad = self.grads[off:off + n].as_strided(shape, stride)
                p._genie_arena = True                 # functional._direct: the wgrad kernels accumulate straight into this view
                self.slots[name] = (off, n)
                self._plist.append(p)
        self.step_count = 0
        self.mirror: Optional[Tensor] = None          # bf16 image of `params`, kept current by the optimiser kernel
        self._packs = None
        self._opt_state: Optional[Tensor] = None      # device-side {step, lr, weight decay, coefficients} of the capture-safe AdamW

    # ------------------------------------------------------------------------------------------------------------------
    # bf16 weight packs without per-step repacking
    # ------------------------------------------------------------------------------------------------------------------
    def attach_weight_packs(self, root: nn.Module) -> int:
        """Keep the conv kernels' bf16 weight packs current as a side effect of the optimiser step instead of re-packing
        ~260 tensors per step: the optimiser kernel also writes a bf16 mirror of the arena -- for a channels_last_3d Conv3d
        weight with in_channels % 8 == 0 that mirror IS the forward pack ([cout][tap][cin]) -- and ONE batched transpose
        kernel rebuilds every backward-data pack ([cin][tap][coutp]).  Returns the number of convolutions managed; the
        rest (stem, 18-channel latent convs, non-arena weights) keep packing themselves on demand."""
        import ctypes as C

        from .module.video import Conv3d
        lib = _hip.load_library()
        if self.mirror is None:
            self.mirror = torch.empty(self.numel, dtype=torch.bfloat16, device=self.params.device)
            _hip.check(lib.genie_cast_f32_to_bf16(self.params.data_ptr(), self.mirror.data_ptr(), self.numel, _hip.stream_ptr()),
                       'genie_cast_f32_to_bf16')
        by_id = {id(p): name for name, p in root.named_parameters()}
        managed, jobs, dst_total, blocks = [], [], 0, 0
        cands = []                                     # (weight parameter, ConvOp, spec, (cout, cin, kt, kh, kw))
        for m in root.modules():
            if isinstance(m, Conv3d):
                cands.append((m.weight, m.op, m.spec, tuple(m.weight.shape)))
            elif isinstance(getattr(m, 'head', None), nn.Linear) and hasattr(m, '_head_op'):
                # DynamicsModel's vocabulary head: Linear(D -> V) run as the 1x1x1 case of the gather-GEMM
                cands.append((m.head.weight, m._head_op, m._head_op.spec, (*m.head.weight.shape, 1, 1, 1)))
        for w, op, spec, (cout, cin, kt, kh, kw) in cands:
            name = by_id.get(id(w))
            if name is None or name not in self.slots:
                continue
            nt = kt * kh * kw
            dense = tuple(w.stride()) == ((nt * cin, 1, kh * kw * cin, kw * cin, cin) if w.dim() == 5 else (cin, 1))
            if cin % 8 != 0 or not dense:
                continue
            off, n = self.slots[name]
            coutp = (cout + 7) & ~7
            perm_c, perm_f = (spec.cfinal, spec.shuffle[0] * spec.shuffle[1] * spec.shuffle[2]) if spec.shuffle is not None else (0, 1)
            tiles_r, tiles_k = (cout + 63) // 64, (cin + 63) // 64
            jobs.append((off, dst_total, cout, nt, cin, perm_c, perm_f, tiles_r, tiles_k, blocks))
            fwd = self.mirror[off:off + n].view(cout, nt, cin)
            managed.append((op, w, fwd, dst_total, (cin, nt, coutp)))
            dst_total += (cin * nt * coutp + 63) // 64 * 64
            blocks += tiles_r * nt * tiles_k
        if not managed:
            return 0
        arr = (_hip.GeniePackJob * len(jobs))(*[_hip.GeniePackJob(*j) for j in jobs])
        host = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).clone()
        bwd = torch.empty(dst_total, dtype=torch.bfloat16, device=self.params.device)
        self._packs = {'jobs': host.to(self.params.device), 'njobs': len(jobs), 'blocks': blocks, 'bwd': bwd,
                       'managed': [(op, w, fwd, bwd[o:o + shp[0] * shp[1] * shp[2]].view(*shp)) for op, w, fwd, o, shp in managed]}
        self._refresh_packs()
        return len(managed)

    def _refresh_packs(self) -> None:
        pk = self._packs
        lib = _hip.load_library()
        _hip.check(lib.genie_pack_transpose_batched(pk['jobs'].data_ptr(), pk['njobs'], pk['blocks'], self.mirror.data_ptr(),
                                                    pk['bwd'].data_ptr(), _hip.stream_ptr()), 'genie_pack_transpose_batched')
        for op, w, fwd, bwdv in pk['managed']:
            key = (w._version, w.data_ptr())
            op._fwd = (key, fwd)
            op._bwd = (key, bwdv)

    def offset_of(self, module: nn.Module, root: nn.Module) -> Optional[int]:
        """Lowest arena offset of the trainable parameters of `module` (a sub-module of `root`)."""
        ids = {id(p) for p in module.parameters() if p.requires_grad}
        offs = [self.slots[name][0] for name, p in root.named_parameters() if id(p) in ids and name in self.slots]
        return min(offs) if offs else None

    def zero_grad(self) -> None:
        from . import functional as GF
        GF.join_wgrad()
        self.grads.zero_()

    def set_graph_hyperparameters(self, lr: float, weight_decay: float) -> None:
        """Device-side learning rate / weight decay / step count of ``adamw_step(graph_safe=True)``: what a scheduler writes between
        replays of a captured step (plain tensor writes -- no re-capture)."""
        if self._opt_state is None:
            self._opt_state = torch.zeros(8, dtype=torch.float32, device=self.params.device)
        self._opt_state[:1].view(torch.int32).fill_(int(self.step_count))
        self._opt_state[1:3].copy_(torch.tensor([float(lr), float(weight_decay)], dtype=torch.float32))

    def mark_updated(self) -> None:
        """Host bookkeeping after the arena was updated by kernels Python did not launch (a graph replay): bump the parameters'
        version counters (on-demand weight packs rebuild on their next eager use) and re-key the packs the optimiser kernels keep
        current themselves."""
        torch.autograd.graph.increment_version(self._plist)
        if self._packs is not None:
            for op, w, fwd, bwdv in self._packs['managed']:
                key = (w._version, w.data_ptr())
                op._fwd = (key, fwd)
                op._bwd = (key, bwdv)

    def adamw_step(self, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 1e-2,
                   grad_scale: float = 1.0, zero_grad: bool = True, graph_safe: bool = False) -> None:
        """torch.optim.AdamW semantics (the reference's optimiser, tokenizer.py:437-442) in ONE kernel over the arena.
        ``graph_safe``: step count, learning rate and weight decay live in device memory (``set_graph_hyperparameters``; `lr` /
        `weight_decay` here are ignored) so that the launch can be captured in a hipGraph and replayed (genie/graph.py)."""
        self.step_count += 1
        lib = _hip.load_library()
        from . import functional as GF
        GF.join_wgrad()                                   # weight-gradient kernels issued on the side stream (functional.ASYNC_WGRAD)
        if graph_safe:
            if self._opt_state is None:
                raise RuntimeError('adamw_step(graph_safe=True): call set_graph_hyperparameters(lr, weight_decay) first')
            _hip.check(lib.genie_adamw_step_graph(self.params.data_ptr(), self.grads.data_ptr(), self.exp_avg.data_ptr(),
                                                  self.exp_avg_sq.data_ptr(), _hip.ptr(self.mirror), self.numel, self._opt_state.data_ptr(),
                                                  betas[0], betas[1], eps, grad_scale, 1 if zero_grad else 0, _hip.stream_ptr()),
                       'genie_adamw_step_graph')
        elif self.mirror is not None:
            _hip.check(lib.genie_adamw_step_mirror(self.params.data_ptr(), self.grads.data_ptr(), self.exp_avg.data_ptr(),
                                                   self.exp_avg_sq.data_ptr(), self.mirror.data_ptr(), self.numel, lr, betas[0], betas[1],
                                                   eps, weight_decay, self.step_count, grad_scale, 1 if zero_grad else 0,
                                                   _hip.stream_ptr()), 'genie_adamw_step_mirror')
        else:
            _hip.check(lib.genie_adamw_step(self.params.data_ptr(), self.grads.data_ptr(), self.exp_avg.data_ptr(),
                                            self.exp_avg_sq.data_ptr(), self.numel, lr, betas[0], betas[1], eps, weight_decay,
                                            self.step_count, grad_scale, 1 if zero_grad else 0, _hip.stream_ptr()), 'genie_adamw_step')
        # the kernel wrote through the arena, not through the Parameter objects: tell autograd, so that the
        # cached bf16 weight packs (functional.ConvOp, keyed on the version counter) are rebuilt ...
        torch.autograd.graph.increment_version(self._plist)
        # ... except the ones this arena keeps current itself (attach_weight_packs)
        if self._packs is not None:
            self._refresh_packs()


class _HostEvent:
    """time.perf_counter() with torch.cuda.Event's elapsed_time() interface (milliseconds): DataParallel's trace on host tensors."""

    def __init__(self) -> None:
        import time
        self.t = time.perf_counter()

    def elapsed_time(self, other: '_HostEvent') -> float:
        return (other.t - self.t) * 1e3


class DataParallel:
    """Clip-sharded data parallelism over the gradient arena.  One process per GPU; ``torch.distributed`` is
    initialised by the caller (``nccl`` = RCCL on the GPUs, ``gloo`` in the CPU tests of the bookkeeping).

    ``compress='bf16'``: a bucket travels as bf16 (arena -> bf16 staging buffer -> all-reduce -> fp32 arena), halving the bytes on
    the xGMI links (0.75 GB instead of 1.5 GB per step for the MAGVIT2 tokenizer) at the price of a bf16 sum over the ranks -- what
    torch DDP's ``bf16_compress_hook`` does; off by default (the fp32 reduction is exact).  ``loopback=True`` issues the
    collectives even when the group has a single rank, so that the side-stream path can be exercised and timed on one GPU."""

    def __init__(self, grads: Tensor, boundaries: Sequence[int] = (), group=None, compress: Optional[str] = None, loopback: bool = False,
                 algorithm: str = 'allreduce') -> None:
        if compress not in (None, 'none', 'bf16'):
            raise ValueError(f"DataParallel: compress must be None or 'bf16', got {compress!r}")
        if algorithm not in ('allreduce', 'rs_ag'):
            raise ValueError(f"DataParallel: algorithm must be 'allreduce' or 'rs_ag', got {algorithm!r}")
        # 'allreduce': one all_reduce per bucket (RCCL picks ring / tree).  'rs_ag': the same sum as an explicit reduce-scatter over all ranks
        # followed by an all-gather (every rank sums 1 / N of the bucket, each of the 7 xGMI links of a GPU carries one
        # peer's shard in each phase -- no ring order, 2 (N - 1) / N of the bytes per GPU as in the ring); the mean's 1 / N scaling runs on the
        # rank's OWN shard between the two phases, i.e. on 1 / N of the elements.  Same result up to fp32 summation order.
        self.algorithm = algorithm
        self._rs_out = None                              # rs_ag: this rank's reduced shard (staging, grown on demand)
        self.grads = grads
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        self.active = self.world > 1 or (loopback and dist.is_available() and dist.is_initialized())
        self.compress = compress if compress != 'none' else None
        self._stage = None                               # bf16 staging buffer, allocated on first use
        cuts = sorted(set([0, grads.numel()] + [int(b) for b in boundaries if 0 < b < grads.numel()]))
        self.buckets = list(zip(cuts[:-1], cuts[1:]))
        self.comm_stream = torch.cuda.Stream(device=grads.device) if grads.is_cuda else None
        self._done = [False] * len(self.buckets)
        self.bytes_reduced = 0                           # payload bytes handed to all_reduce so far
        self.fired: List[int] = []                       # bucket indices in the order their reduction was issued (this step)
        self._hooks: list = []
        # optional timing of the overlap (bench.py --gpus N): per step, HIP events at every bucket's issue point (compute stream), around
        # its all-reduce (comm stream) and around finish()'s wait -- read back by comm_report() after a synchronise
        self.trace = False
        self._events: list = []                          # per step: {'issue': {i: ev}, 'comm': {i: (ev0, ev1)}, 'wait': (ev0, ev1)}
        self._cur: Optional[dict] = None

    def _sum_over_ranks(self, buf: Tensor, scale: float) -> None:
        """buf <- scale * sum over the ranks of buf, in place, by the configured algorithm."""
        w = self.world
        if self.algorithm == 'rs_ag' and buf.numel() >= w:
            main = buf.numel() - buf.numel() % w
            shard = main // w
            if self._rs_out is None or self._rs_out.numel() < shard or self._rs_out.dtype != buf.dtype:
                self._rs_out = torch.empty(shard, dtype=buf.dtype, device=buf.device)
            out = self._rs_out[:shard]
            dist.reduce_scatter_tensor(out, buf[:main], op=dist.ReduceOp.SUM, group=self.group)
            if scale != 1.0:
                out.mul_(scale)
            dist.all_gather_into_tensor(buf[:main], out, group=self.group)
            if main < buf.numel():                       # the few elements that do not divide by the world size
                tail = buf[main:]
                dist.all_reduce(tail, op=dist.ReduceOp.SUM, group=self.group)
                if scale != 1.0:
                    tail.mul_(scale)
            return
        dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group)
        if scale != 1.0:
            buf.mul_(scale)

    def _all_reduce_mean(self, chunk: Tensor, lo: int) -> None:
        scale = 1.0 / self.world if self.world > 1 else 1.0
        if self.compress == 'bf16':
            if self._stage is None:
                self._stage = torch.empty(self.grads.numel(), dtype=torch.bfloat16, device=self.grads.device)
            st = self._stage[lo:lo + chunk.numel()]
            st.copy_(chunk)
            self._sum_over_ranks(st, 1.0)                # (the mean's scaling in fp32, after the expansion: bf16 would round it twice)
            chunk.copy_(st)
            if scale != 1.0:
                chunk.mul_(scale)
            self.bytes_reduced += chunk.numel() * 2
        else:
            self._sum_over_ranks(chunk, scale)
            self.bytes_reduced += chunk.numel() * 4

    def _reduce(self, lo: int, hi: int) -> None:
        if not self.active or hi <= lo:
            return
        chunk = self.grads[lo:hi]
        if self.comm_stream is not None:
            from . import functional as GF
            tr = self._trace_step() if self.trace else None
            if tr is not None:
                ev = torch.cuda.Event(enable_timing=True); ev.record(torch.cuda.current_stream())
                tr['issue'][lo] = ev
            self.comm_stream.wait_stream(torch.cuda.current_stream())       # behind every kernel enqueued so far ...
            GF.join_wgrad(self.comm_stream)                                 # ... including weight gradients on their side stream
            with torch.cuda.stream(self.comm_stream):
                if tr is not None:
                    e0 = torch.cuda.Event(enable_timing=True); e0.record(self.comm_stream)
                self._all_reduce_mean(chunk, lo)
                if tr is not None:
                    e1 = torch.cuda.Event(enable_timing=True); e1.record(self.comm_stream)
                    tr['comm'][lo] = (e0, e1)
        else:
            # host tensors (the gloo tests): the collective is synchronous; host clocks stand in for the HIP events so that comm_report()
            # has the same schema on every backend
            tr = self._trace_step() if self.trace else None
            t0 = _HostEvent() if tr is not None else None
            if tr is not None:
                tr['issue'][lo] = t0
            self._all_reduce_mean(chunk, lo)
            if tr is not None:
                tr['comm'][lo] = (t0, _HostEvent())

    def _trace_step(self) -> dict:
        if self._cur is None:
            self._cur = {'issue': {}, 'comm': {}, 'wait': None}
        return self._cur

    def bucket_ready(self, index: int) -> None:
        """Gradients of bucket `index` (and of every later bucket) are fully enqueued."""
        for i in range(len(self.buckets) - 1, index - 1, -1):
            if not self._done[i]:
                self._reduce(*self.buckets[i])
                self._done[i] = True
                self.fired.append(i)

    def finish(self) -> None:
        """Reduce whatever is left and make the compute stream wait for the reductions (call before the optimiser)."""
        self.bucket_ready(0)
        if self.comm_stream is not None and self.active:
            if self.trace:
                tr = self._trace_step()
                w0 = torch.cuda.Event(enable_timing=True); w0.record(torch.cuda.current_stream())      # = end of backward on the compute stream
            torch.cuda.current_stream().wait_stream(self.comm_stream)
            if self.trace:
                w1 = torch.cuda.Event(enable_timing=True); w1.record(torch.cuda.current_stream())
                tr['wait'] = (w0, w1)
                self._events.append(tr)
                self._cur = None
        elif self.trace and self.active:                 # host path: nothing is left to wait for at this point
            tr = self._trace_step()
            w = _HostEvent()
            tr['wait'] = (w, w)
            self._events.append(tr)
            self._cur = None
        self._done = [False] * len(self.buckets)
        self.last_fired, self.fired = self.fired, []
        if hasattr(self, 'armed'):
            self.last_armed, self.armed = self.armed, []

    def comm_report(self) -> dict:
        """Mean over the traced steps (call after torch.cuda.synchronize()): per bucket its payload, how long before the END of
        backward its all-reduce was issued and how long the collective took; `exposed_ms` = what the compute stream spent waiting for
        the comm stream in finish() -- the communication that backward did NOT hide.  Makes a scaling run say why it scales."""
        if not self._events:
            return {}
        n = len(self._events)
        per = {}
        exposed = 0.0
        for tr in self._events:
            w0, w1 = tr['wait']
            exposed += w0.elapsed_time(w1)
            for lo, ev in tr['issue'].items():
                d = per.setdefault(lo, {'issued_before_backward_end_ms': 0.0, 'allreduce_ms': 0.0})
                d['issued_before_backward_end_ms'] += ev.elapsed_time(w0)
                e0, e1 = tr['comm'][lo]
                d['allreduce_ms'] += e0.elapsed_time(e1)
        el = 2 if self.compress == 'bf16' else 4
        buckets = []
        for (lo, hi) in self.buckets:
            d = per.get(lo, {})
            buckets.append({'elements': hi - lo, 'payload_MB': round((hi - lo) * el / 1e6, 1),
                            'issued_before_backward_end_ms': round(d.get('issued_before_backward_end_ms', 0.0) / n, 3),
                            'allreduce_ms': round(d.get('allreduce_ms', 0.0) / n, 3)})
        tot = sum(b['allreduce_ms'] for b in buckets)
        return {'algorithm': self.algorithm, 'payload': 'bf16' if self.compress == 'bf16' else 'fp32', 'world': self.world,
                'steps_traced': n, 'exposed_ms_per_step': round(exposed / n, 3), 'allreduce_ms_per_step': round(tot, 3),
                'hidden_fraction': round(1.0 - (exposed / n) / tot, 4) if tot > 0 else None,
                'bus_GBps': round(sum(b['payload_MB'] for b in buckets) * 1e-3 * 2 * (self.world - 1) / max(self.world, 1) / (tot * 1e-3), 1) if tot > 0 else None,
                'buckets': buckets}

    def reduce_scalars(self, values: Sequence[Tensor]) -> Tensor:
        """One all-reduce for all logged scalars of a step (mean over ranks)."""
        v = torch.stack([torch.as_tensor(x, dtype=torch.float32, device=self.grads.device).detach().reshape(()) for x in values])
        if self.world > 1:
            dist.all_reduce(v, op=dist.ReduceOp.SUM, group=self.group)
            v /= self.world
        return v

    def install_overlap_hooks(self, arena: ParamArena, root: nn.Module, modules: Sequence[nn.Module]) -> None:
        """Cut the arena at the first parameter of each module in `modules` and start a bucket's reduction when backward delivers
        the gradient of that module's input -- by then the module itself and everything that ran after it in forward has enqueued
        its weight gradients (the HIP backward functions launch their wgrad kernels before returning; torch's own AccumulateGrad
        nodes run with top priority as soon as their producer has).  `modules` must be in forward order AND the arena must be laid
        out in forward order (``ParamArena`` does that through ``execution_order``): both are checked here, because a bucket that
        fires before one of its gradients exists silently de-synchronises the replicas."""
        for h in self._hooks:
            h.remove()
        self._hooks = []
        offs = [arena.offset_of(m, root) for m in modules]
        pairs = [(o, m) for o, m in zip(offs, modules) if o is not None]
        if any(o2 <= o1 for (o1, _), (o2, _) in zip(pairs, pairs[1:])):
            raise ValueError('install_overlap_hooks: modules must be given in forward order and the arena must be laid out in '
                             'execution order (ParamArena(model, order=...) / model.forward_order()); got offsets '
                             f'{[o for o, _ in pairs]}')
        if len(pairs) > 0 and not arena.order_is_execution:
            raise ValueError('install_overlap_hooks: the arena is in registration order (the model defines no forward_order() and no '
                             '`order` was given to ParamArena); early bucket reduction would be unsafe')
        cuts = sorted(set([0, self.grads.numel()] + [o for o, _ in pairs if 0 < o < self.grads.numel()]))
        self.buckets = list(zip(cuts[:-1], cuts[1:]))
        self._done = [False] * len(self.buckets)
        index_of = {lo: i for i, (lo, _) in enumerate(self.buckets)}
        self.armed: List[int] = []                       # buckets whose early-reduction hook was registered in the current step
        for off, mod in pairs:
            if off not in index_of:
                continue
            idx = index_of[off]

            def pre_hook(_m, args, idx=idx):
                x = args[0] if args else None
                if isinstance(x, Tensor) and x.requires_grad and idx not in self.armed:
                    self.armed.append(idx)
                    x.register_hook(lambda g, idx=idx: self.bucket_ready(idx))

            # a stage that is a container (nn.ModuleList iterated by the model's own loop, e.g. VideoTokenizer.enc_layers / dec_layers) is
            # never CALLED, so a pre-hook on it would never fire: hook the module that actually receives the stage's input.
            # A stage whose input carries no gradient (integer tokens) or that is entered through another method than __call__ simply
            # stays un-armed in that step and is reduced by finish() -- late, never early.
            self._hooks.append(self.entry_module(mod).register_forward_pre_hook(pre_hook))

    @staticmethod
    def equal_byte_cuts(arena: ParamArena, root: nn.Module, candidates: Sequence[nn.Module], nbuckets: int) -> List[nn.Module]:
        """Pick from `candidates` (modules in forward order, e.g. the layers of enc_layers + dec_layers) the ones whose first parameter
        lies closest to the k / nbuckets points of the arena: buckets of (nearly) equal BYTES, so that the all-reduces of a step take
        equal times and the last one -- the only one backward cannot hide -- is as small as the others (bench.py used hard-coded layer
        indices in round 2)."""
        offs = [(arena.offset_of(m, root), m) for m in candidates]
        offs = [(o, m) for o, m in offs if o is not None and 0 < o < arena.numel]
        picks, used = [], set()
        if offs and nbuckets > 1:
            # plus one cut right behind the FIRST layers: bucket 0 is the one finish() reduces after backward has ended -- the only
            # all-reduce nothing hides -- so it should hold as little as possible (the stem and whatever is laid out before it), and
            # the first equal-byte bucket then starts while backward still has the full-resolution encoder layers to go
            o, m = min(offs, key=lambda om: om[0])
            used.add(o)
            picks.append((o, m))
        # ... and one more near 1 / (8 nbuckets) of the arena: the bucket behind the small first one holds the EARLIEST layers, whose gradients
        # are the last backward produces -- its hook fires a fraction of a millisecond before backward ends (measured with a single-rank
        # RCCL group on the tokenizer: 185 MB issued 0.34 ms before the end = exposed on a real ring), so it should be small as well; the
        # rest of that range then starts its reduction while backward still has the full-resolution encoder layers to go
        targets = [arena.numel * k / nbuckets for k in range(1, nbuckets)]
        if nbuckets > 1:
            targets.append(arena.numel / (8.0 * nbuckets))
        for target in targets:
            o, m = min(offs, key=lambda om: abs(om[0] - target), default=(None, None))
            if o is not None and o not in used:
                used.add(o)
                picks.append((o, m))
        return [m for _, m in sorted(picks, key=lambda om: om[0])]

    @staticmethod
    def entry_module(mod: nn.Module) -> nn.Module:
        """The sub-module that receives a stage's input: descends through containers that have no forward of their own."""
        while isinstance(mod, (nn.ModuleList, nn.ModuleDict)) and len(mod) > 0:
            mod = mod[0] if isinstance(mod, nn.ModuleList) else next(iter(mod.values()))
        return mod


def sync_replicas(arena: ParamArena, model: nn.Module, group=None, seed: Optional[int] = None) -> None:
    """Make every rank start from rank 0's weights and buffers (what Lightning's DDP strategy -- the reference's
    ``strategy: ddp``, config/tokenize.yaml:77 -- does when it wraps the module), then give each rank its OWN random stream
    (seed + rank) so that per-rank stochastic operations (MaskGIT Bernoulli masks, GAN frame picks, synthetic clips) differ across
    ranks.  Without the broadcast, replica equality would rest on every rank having seeded identically before building the model
    (ADVICE r2).  Call BEFORE ``arena.attach_weight_packs`` (the bf16 mirror is built from the arena)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) < 2:
        return
    dist.broadcast(arena.params, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
    for b in model.buffers():
        if b.numel():
            dist.broadcast(b, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
    torch.autograd.graph.increment_version(arena._plist)
    if seed is not None:
        torch.manual_seed(int(seed) + dist.get_rank(group))


def shard_clips(num_clips: int, rank: int, world: int) -> range:
    """Rank r takes clips r::world -- what Lightning's DistributedSampler does for the reference's DataLoader
    (reference genie/module/data.py:97)."""
    return range(rank, num_clips, world)


# ----------------------------------------------------------------------------------------------------------------------------------
# fit loop: what ``Trainer.fit`` does for the reference's LightningCLI entry points (tokenizer.py:6-19, config/tokenize.yaml:74-92),
# on this runtime: parameter arena + fused AdamW, clip-sharded data parallel, pinned prefetch, one scalar all-reduce per log line
# ----------------------------------------------------------------------------------------------------------------------------------
class Trainer:
    """Minimal stand-in for ``lightning.Trainer`` with the keys of the reference's ``trainer:`` config section that matter here:
    ``max_epochs``, ``max_steps``, ``log_every_n_steps``, ``val_check_interval``, ``limit_val_batches``, ``default_root_dir``;
    ``accelerator`` / ``devices`` / ``strategy`` / ``precision`` / ``callbacks`` / ``logger`` are accepted and recorded (one process
    per GPU is launched by ``torch.distributed.run``; arithmetic is bf16 with fp32 masters; ``save_last`` is always on)."""

    def __init__(self, max_epochs: int = 1, max_steps: int = -1, log_every_n_steps: int = 16, val_check_interval: int | None = None,
                 limit_val_batches: int | None = None, default_root_dir: str = 'runs', grad_compress: Optional[str] = None, grad_algorithm: str = 'allreduce',
                 graph: bool = False, grad_buckets: int = 8, device_state_adamw: bool = False, **ignored) -> None:
        # graph: after two eager steps the training step (forward, backward, AdamW) is captured in a hipGraph and replayed per batch
        # (genie/graph.py; single-GPU runs, batches of the captured shape -- anything else takes the eager path)
        self.graph = bool(graph)
        # device_state_adamw: the eager steps use the capture-safe AdamW (hyper-parameters read from device memory) as well -- the arithmetic
        # of a graph=True run without the graph, for A/B runs that must agree bit for bit (tests/test_gpu_graph.py)
        self.device_state_adamw = bool(device_state_adamw)
        self.grad_buckets = max(1, int(grad_buckets))
        self.max_epochs, self.max_steps = max_epochs, (max_steps if max_steps and max_steps > 0 else None)
        self.log_every_n_steps, self.val_check_interval, self.limit_val_batches = max(1, log_every_n_steps), val_check_interval, limit_val_batches
        self.default_root_dir, self.grad_compress, self.ignored = default_root_dir, grad_compress, dict(ignored)
        self.grad_algorithm = grad_algorithm               # 'allreduce' | 'rs_ag' (DataParallel)
        self.global_step = 0
        self.history: List[dict] = []

    @staticmethod
    def bucket_modules(arena: ParamArena, model, nbuckets: int = 8) -> list:
        """Where fit() cuts the gradient arena for the overlapped all-reduce: `nbuckets` buckets of (nearly) equal BYTES plus a small first
        one, chosen among the LAYERS of the model's stages (``DataParallel.equal_byte_cuts``) -- the same rule ``bench.py --gpus N`` uses,
        so the scaling bench measures the path users run (round 3 cut one bucket per top-level stage here: two or
        three buckets of very unequal size)."""
        layers = []
        for stage in model.forward_order():
            subs = list(stage) if isinstance(stage, nn.ModuleList) else [stage]
            layers += [m for m in subs if any(p.requires_grad for p in m.parameters())]
        return DataParallel.equal_byte_cuts(arena, model, layers, nbuckets)

    @staticmethod
    def _adamw_hparams(model) -> dict:
        opt = model.configure_optimizers()
        if type(opt).__name__ != 'AdamW':
            raise NotImplementedError(f'Trainer: the fused optimiser kernel implements AdamW (the reference default, tokenizer.py:22,437-442); got {type(opt).__name__}')
        d = opt.defaults
        return dict(lr=d['lr'], betas=tuple(d['betas']), eps=d['eps'], weight_decay=d['weight_decay'])

    def _log(self, model, dp: 'DataParallel', tag: str) -> dict:
        # 'train' lines report what the last TRAINING step logged: a validation pass in between overwrites model._last_logged, and a
        # hipGraph replay never re-enters Python to refresh it -- the replayed step's metrics are the captured tensors
        logged = (getattr(self, '_train_logged', None) if tag == 'train' else None) or getattr(model, '_last_logged', {})
        keys = sorted(logged)
        vals = dp.reduce_scalars([logged[k] for k in keys]) if keys else torch.zeros(0)
        rec = {'step': self.global_step, 'split': tag, **{k: round(v, 6) for k, v in zip(keys, vals.tolist())}}
        self.history.append(rec)
        if not dist.is_initialized() or dist.get_rank() == 0:
            import json
            print(json.dumps(rec), flush=True)
        return rec

    def validate(self, model, loader, dp) -> None:
        from .module.data import DevicePrefetcher
        model.eval()
        with torch.no_grad():
            for i, batch in enumerate(DevicePrefetcher(loader)):
                if self.limit_val_batches is not None and i >= self.limit_val_batches:
                    break
                model.validation_step(batch, i)
        model.train()
        self._log(model, dp, 'val')

    def save_last(self, model, arena: Optional[ParamArena] = None, hp: Optional[dict] = None, epoch: int = 0) -> str:
        """``last.ckpt`` with the top-level keys of a Lightning checkpoint (``ModelCheckpoint(save_last=True)`` of the reference's config,
        config/tokenize.yaml:80-86): ``state_dict``, ``global_step``, ``epoch`` and ``optimizer_states`` -- AdamW's step / exp_avg /
        exp_avg_sq per parameter -- which is what makes the run resumable HERE (``fit(ckpt_path=...)``).  Interoperability with the
        reference is limited to ``state_dict`` (``VideoTokenizer.load_state_dict`` / the reference's ``load_from_checkpoint`` with
        ``strict`` keys): the optimiser state is keyed by parameter NAME (so that a resume does not depend on the arena layout) with a
        plain-int ``step``, where torch / Lightning key by integer index, and ``hyper_parameters`` / ``pytorch-lightning_version`` are
        not written -- a Lightning ``Trainer(resume)`` cannot consume the optimiser part.  Tensors go to the host one at a time."""
        import os
        path = os.path.join(self.default_root_dir, 'last.ckpt')
        if not dist.is_initialized() or dist.get_rank() == 0:
            os.makedirs(self.default_root_dir, exist_ok=True)
            ck = {'state_dict': {k: v.detach().cpu() for k, v in model.state_dict().items()}, 'global_step': self.global_step, 'epoch': epoch,
                  'loops': {'batches_done_in_epoch': getattr(self, '_batches_done', 0)}, 'genie_runtime': 'genie-mi355x'}
            if arena is not None:
                state = {}
                for name, (off, n) in arena.slots.items():
                    state[name] = {'step': arena.step_count, 'exp_avg': arena.exp_avg[off:off + n].cpu(), 'exp_avg_sq': arena.exp_avg_sq[off:off + n].cpu()}
                ck['optimizer_states'] = [{'state': state, 'param_groups': [dict(hp or {}, params=list(arena.slots))]}]
            torch.save(ck, path)
        return path

    @staticmethod
    def load_checkpoint(path: str, model, arena: ParamArena) -> dict:
        """Restore parameters (through the arena views), buffers and the AdamW moments saved by ``save_last``; returns the checkpoint."""
        ck = torch.load(path, map_location='cpu')
        model.load_state_dict(ck['state_dict'])                     # copies INTO the arena views (p.data are views)
        torch.autograd.graph.increment_version(arena._plist)
        opt = ck.get('optimizer_states')
        if opt:
            state = opt[0]['state']
            for name, (off, n) in arena.slots.items():
                st = state.get(name)
                if st is None:
                    raise KeyError(f'checkpoint has no optimiser state for {name}')
                arena.exp_avg[off:off + n].copy_(st['exp_avg'].reshape(-1))
                arena.exp_avg_sq[off:off + n].copy_(st['exp_avg_sq'].reshape(-1))
                arena.step_count = int(st['step'])
        return ck

    def fit(self, model, datamodule, ckpt_path: Optional[str] = None, seed: Optional[int] = None) -> 'Trainer':
        """`ckpt_path`: resume from a ``last.ckpt`` (weights, AdamW moments, step counter).  `seed`: after the replicas have been
        synchronised each rank reseeds with seed + rank."""
        from .module.data import DevicePrefetcher
        dev = torch.device('cuda', torch.cuda.current_device())
        model.to(dev).train()
        datamodule.setup('fit')
        hp = self._adamw_hparams(model)
        arena = ParamArena(model)
        start_epoch, skip = 0, 0
        if ckpt_path:
            ck = self.load_checkpoint(ckpt_path, model, arena)
            self.global_step, start_epoch = int(ck.get('global_step', 0)), int(ck.get('epoch', 0))
            skip = int((ck.get('loops') or {}).get('batches_done_in_epoch', 0))
        sync_replicas(arena, model, seed=seed)
        arena.attach_weight_packs(model)
        self.arena = arena
        dp = DataParallel(arena.grads, compress=self.grad_compress, algorithm=self.grad_algorithm)
        if dp.active and hasattr(model, 'forward_order'):
            dp.install_overlap_hooks(arena, model, self.bucket_modules(arena, model, self.grad_buckets))
        done = self.max_steps is not None and self.global_step >= self.max_steps
        epoch = start_epoch
        use_graph = self.graph and not dp.active
        if use_graph and not getattr(model, 'graph_capture_safe', False):
            raise ValueError(f'Trainer(graph=True): the training step of {type(model).__name__} draws per-step randomness on the host (or does not say '
                             'otherwise: `graph_capture_safe`); a replayed hipGraph would repeat it.  Train it eagerly.')
        gstep, eager_steps = None, 0
        side = None
        if use_graph or self.device_state_adamw:
            arena.set_graph_hyperparameters(hp['lr'], hp['weight_decay'])
        if use_graph:
            # everything before the capture runs on a NON-default stream: autograd remembers the stream every parameter's gradient was
            # last accumulated on and synchronises with it during backward -- with the legacy default stream that is illegal inside a capture
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream(dev))
        with (torch.cuda.stream(side) if side is not None else contextlib.nullcontext()):
            epoch = self._epochs(model, datamodule, dp, arena, hp, start_epoch, skip, done, use_graph)
        if side is not None:
            torch.cuda.current_stream(dev).wait_stream(side)
        self._log(model, dp, 'train')
        self.save_last(model, arena, hp, epoch)
        return self

    def _epochs(self, model, datamodule, dp, arena, hp, start_epoch: int, skip: int, done: bool, use_graph: bool) -> int:
        from .module.data import DevicePrefetcher
        gstep, eager_steps, graph_logged = None, 0, None
        epoch = start_epoch
        self._batches_done = skip          # a resume that is already at max_steps never enters the loop: keep the saved position
        for epoch in range(start_epoch, self.max_epochs if not done else start_epoch):
            loader = datamodule.train_dataloader()
            if hasattr(getattr(loader, 'sampler', None), 'set_epoch'):
                loader.sampler.set_epoch(epoch)
            self._batches_done = 0
            for i, batch in enumerate(DevicePrefetcher(loader)):
                if epoch == start_epoch and i < skip:             # resumed mid-epoch: these batches were consumed before the checkpoint
                    self._batches_done = i + 1
                    continue
                if gstep is not None and torch.is_tensor(batch) and batch.shape == gstep.batch.shape and batch.dtype == gstep.batch.dtype:
                    gstep(batch)                                  # one replay: forward, backward, AdamW
                    self._train_logged = graph_logged             # the tensors the captured training_step logged, rewritten by the replay
                elif use_graph and gstep is None and eager_steps >= 2 and torch.is_tensor(batch):
                    from .graph import GraphedTrainStep
                    gstep = GraphedTrainStep(model, arena, batch, loss_fn=lambda m, b: m.training_step(b, 0), lr=hp['lr'], betas=hp['betas'],
                                             eps=hp['eps'], weight_decay=hp['weight_decay'], warmup=0)      # captures AND performs this step
                    graph_logged = self._train_logged = dict(getattr(model, '_last_logged', {}))
                else:
                    loss = model.training_step(batch, i)
                    loss.backward()
                    dp.finish()
                    arena.adamw_step(**hp, graph_safe=use_graph or self.device_state_adamw)
                    eager_steps += 1
                    self._train_logged = dict(getattr(model, '_last_logged', {}))
                self.global_step += 1
                self._batches_done = i + 1
                if self.global_step % self.log_every_n_steps == 0:
                    self._log(model, dp, 'train')
                if self.val_check_interval and self.global_step % self.val_check_interval == 0:
                    self.validate(model, datamodule.val_dataloader(), dp)
                if self.max_steps is not None and self.global_step >= self.max_steps:
                    done = True
                    break
            if done:
                break
        return epoch
