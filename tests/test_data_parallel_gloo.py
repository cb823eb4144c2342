"""The N > 1 path on CPU: world_size-2 gloo runs of (a) the data-parallel bookkeeping (bucket cuts, readiness order, mean reduction,
batched scalar reduction, bf16 compression) and (b) the real thing minus the kernels -- an nn.Module re-homed into a ParamArena laid
out in execution order, overlap hooks installed, forward/backward on different data per rank -- checked against an unbucketed
all-reduce.  The arithmetic kernels are not involved (torch CPU ops stand in for them)."""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn

from util import ROOT


def _setup(rank: int, world: int, port: int):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    for p in (ROOT, os.path.join(ROOT, 'open-genie_amd')):
        if p not in sys.path:
            sys.path.insert(0, p)
    dist.init_process_group('gloo', rank=rank, world_size=world)


def _worker(rank: int, world: int, port: int, q):
    _setup(rank, world, port)
    try:
        from genie.trainer import DataParallel, shard_clips
        grads = torch.arange(1000, dtype=torch.float32) * (rank + 1)
        dp = DataParallel(grads, boundaries=[256, 640])
        assert dp.buckets == [(0, 256), (256, 640), (640, 1000)] and dp.world == world
        dp.bucket_ready(2)                       # backward reaches the last bucket first
        assert torch.equal(grads[640:], torch.arange(640, 1000, dtype=torch.float32) * 1.5)
        assert torch.equal(grads[:640], torch.arange(640, dtype=torch.float32) * (rank + 1))      # not yet reduced
        dp.bucket_ready(1)
        dp.finish()                              # reduces what is left exactly once
        assert dp.last_fired == [2, 1, 0]
        assert torch.equal(grads, torch.arange(1000, dtype=torch.float32) * 1.5)
        dp.finish()                              # a second step starts clean
        assert torch.allclose(grads, torch.arange(1000, dtype=torch.float32) * 1.5)
        s = dp.reduce_scalars([torch.tensor(float(rank)), 2.0 * (rank + 1)])
        assert torch.allclose(s, torch.tensor([0.5, 3.0]))
        clips = list(shard_clips(7, rank, world))
        # bf16 gradient compression: half the bytes, bf16-rounded mean
        g2 = (torch.arange(1000, dtype=torch.float32) * 0.37 + 1.0) * (rank + 1)
        dpc = DataParallel(g2, boundaries=[500], compress='bf16')
        dpc.finish()
        want = ((torch.arange(1000, dtype=torch.float32) * 0.37 + 1.0).to(torch.bfloat16).float() * 1
                + ((torch.arange(1000, dtype=torch.float32) * 0.37 + 1.0) * 2).to(torch.bfloat16).float()).to(torch.bfloat16).float() / 2
        assert torch.allclose(g2, want, rtol=2 ** -7), (g2 - want).abs().max()
        assert dpc.bytes_reduced == 1000 * 2
        # explicit reduce-scatter + all-gather (--allreduce rs_ag) == the all-reduce, bucket sizes that do and do not divide by the world
        # size (odd buckets: the tail goes through a small all-reduce), early + late buckets, fp32 and bf16 payload
        base = torch.arange(1003, dtype=torch.float32) * 0.37 + 1.0
        ga, gr = base * (rank + 1), base * (rank + 1)
        da = DataParallel(ga, boundaries=[257, 641])
        dr = DataParallel(gr, boundaries=[257, 641], algorithm='rs_ag')
        for d_ in (da, dr):
            d_.bucket_ready(2)
            d_.finish()
        assert torch.equal(ga, gr), (ga - gr).abs().max()              # two ranks: the sums are the same two addends in either order
        assert dr.last_fired == [2, 1, 0] and dr.bytes_reduced == da.bytes_reduced
        gb_a, gb_r = base * (rank + 1), base * (rank + 1)
        DataParallel(gb_a, boundaries=[500], compress='bf16').finish()
        DataParallel(gb_r, boundaries=[500], compress='bf16', algorithm='rs_ag').finish()
        assert torch.equal(gb_a, gb_r)
        tiny = torch.ones(1) * (rank + 1)                                # a bucket smaller than the world: plain all-reduce
        DataParallel(tiny, algorithm='rs_ag').finish()
        assert tiny.item() == 1.5
        q.put((rank, clips, dp.bytes_reduced))
    finally:
        dist.destroy_process_group()


class _Block(nn.Module):
    def __init__(self, d):
        super().__init__()
        self.fc = nn.Linear(d, d)

    def forward(self, x):
        return x + torch.tanh(self.fc(x))


class _Toy(nn.Module):
    """Registration order != execution order, as in VideoTokenizer (quant after dec_layers) and DynamicsModel (embeddings after
    dec_layers): `late` is registered first but runs last, `quant` is registered last but runs in the middle."""

    def __init__(self, d=16):
        super().__init__()
        self.late = nn.Linear(d, 1)
        self.enc = nn.ModuleList([_Block(d) for _ in range(3)])
        self.dec = nn.ModuleList([_Block(d) for _ in range(3)])
        self.quant = nn.Linear(d, d)
        self.stray = nn.Parameter(torch.ones(d))          # used everywhere, listed nowhere -> goes first in the arena

    def forward_order(self):
        return [self.enc, self.quant, self.dec, self.late]

    def forward(self, x):
        for m in self.enc:
            x = m(x)
        x = self.quant(x) * self.stray
        for m in self.dec:
            x = m(x)
        return self.late(x * self.stray).pow(2).mean()


def _model_worker(rank: int, world: int, port: int, q, compress):
    _setup(rank, world, port)
    try:
        from genie.trainer import DataParallel, ParamArena
        torch.manual_seed(0)
        model = _Toy()
        ref = _Toy()
        ref.load_state_dict(model.state_dict())
        arena = ParamArena(model)
        names = arena.order_names
        assert names[0] == 'stray' and names.index('quant.weight') < names.index('dec.0.fc.weight') < names.index('late.weight'), names
        assert names.index('enc.2.fc.bias') < names.index('quant.weight')
        dp = DataParallel(arena.grads, compress=compress)
        cuts = [model.enc[1], model.quant, model.dec[1], model.late]
        dp.install_overlap_hooks(arena, model, cuts)
        assert len(dp.buckets) == 5
        fired_during_backward = []
        for step in range(2):
            torch.manual_seed(100 + rank + 10 * step)              # different data on every rank
            x = torch.randn(8, 16)
            loss = model(x)
            loss.backward()
            fired_during_backward.append(list(dp.fired))           # buckets whose reduction was issued by hooks, before finish()
            dp.finish()
            # unbucketed reference: plain autograd on a copy, one all-reduce over every gradient
            ref.zero_grad()
            ref(x).backward()
            for (n, p) in ref.named_parameters():
                g = p.grad.clone()
                dist.all_reduce(g)
                g /= world
                got = dict(model.named_parameters())[n].grad
                tol = dict(rtol=2 ** -6, atol=1e-3) if compress else dict(rtol=1e-6, atol=1e-7)
                assert torch.allclose(got, g, **tol), (n, (got - g).abs().max().item())
            arena.zero_grad()
        # buckets fire in reverse arena order while backward is still running: late -> dec[1:] -> quant+dec[0] -> enc[1:]
        assert fired_during_backward[0] == [4, 3, 2, 1], fired_during_backward
        assert dp.last_fired == [4, 3, 2, 1, 0]
        # an arena in registration order must refuse early reduction
        m2 = _Toy()
        del _Toy.forward_order
        try:
            a2 = ParamArena(m2)
            try:
                DataParallel(a2.grads).install_overlap_hooks(a2, m2, [m2.enc[1]])
                raise AssertionError('registration-order arena accepted overlap hooks')
            except ValueError:
                pass
        finally:
            pass
        q.put((rank, dp.bytes_reduced))
    finally:
        dist.destroy_process_group()


def _sync_worker(rank: int, world: int, port: int, q):
    """(a) replicas built from DIFFERENT seeds are equal after sync_replicas and draw different random numbers afterwards;
    (b) stages given as nn.ModuleList containers (what VideoTokenizer.forward_order() returns: the model's own loop iterates them, they
    are never called) still arm their early-reduction hooks, through their first layer."""
    _setup(rank, world, port)
    try:
        from genie.trainer import DataParallel, ParamArena, sync_replicas
        torch.manual_seed(1234 + rank)                             # a config without seed_everything: every rank initialises differently
        model = _Toy()
        model.register_buffer('table', torch.randn(5))
        arena = ParamArena(model)
        before = arena.params.clone()
        sync_replicas(arena, model, seed=7)
        gathered = [torch.empty_like(arena.params) for _ in range(world)]
        dist.all_gather(gathered, arena.params)
        assert all(torch.equal(g, gathered[0]) for g in gathered)
        if rank > 0:
            assert not torch.equal(before, arena.params)
        bufs = [torch.empty(5) for _ in range(world)]
        dist.all_gather(bufs, model.table)
        assert torch.equal(bufs[0], bufs[1])
        # parameters are views of the arena: the module sees the broadcast weights
        assert torch.equal(model.late.weight.detach().reshape(-1), arena.params[arena.slots['late.weight'][0]:][:16])
        draws = [torch.empty(3) for _ in range(world)]
        dist.all_gather(draws, torch.rand(3))
        assert not torch.equal(draws[0], draws[1])                 # per-rank random streams after the broadcast
        dp = DataParallel(arena.grads)
        dp.install_overlap_hooks(arena, model, [model.quant, model.dec, model.late])     # model.dec is a ModuleList
        assert DataParallel.entry_module(model.dec) is model.dec[0]
        x = torch.randn(4, 16)
        model(x).backward()
        assert dp.fired == [3, 2, 1], dp.fired                     # all three armed and fired during backward
        dp.finish()
        assert dp.last_armed == [1, 2, 3] and dp.last_fired == [3, 2, 1, 0]
        q.put((rank, float(arena.params.sum())))
    finally:
        dist.destroy_process_group()


def _comm_worker(rank: int, world: int, port: int, q, algorithm, compress):
    """DataParallel.trace + comm_report() with two ranks: the N > 1 schema bench.py prints as `comm` -- host clocks
    stand in for the HIP events on gloo."""
    _setup(rank, world, port)
    try:
        from genie.trainer import DataParallel
        grads = torch.arange(1001, dtype=torch.float32) * (rank + 1)
        dp = DataParallel(grads, boundaries=[256, 640], algorithm=algorithm, compress=compress)
        assert dp.active and dp.comm_report() == {}
        dp.trace = True
        for step in range(3):
            grads.copy_(torch.arange(1001, dtype=torch.float32) * (rank + 1))
            dp.bucket_ready(2)
            dp.bucket_ready(1)
            dp.finish()
        dp.trace = False
        dp.finish()                                           # an untraced step does not enter the report
        q.put((rank, dp.comm_report()))
    finally:
        dist.destroy_process_group()


def _run(target, extra=()):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29500 + (os.getpid() * 7 + len(extra) * 131 + hash(str(extra)) % 997) % 2000
    procs = [ctx.Process(target=target, args=(r, 2, port, q, *extra)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    return sorted(q.get(timeout=5) for _ in range(2))


def test_world_size_2_gloo():
    res = _run(_worker)
    assert res[0][1] == [0, 2, 4, 6] and res[1][1] == [1, 3, 5]
    assert res[0][2] == 2 * 1000 * 4


def test_world_size_2_module_arena_hooks():
    res = _run(_model_worker, (None,))
    assert res[0][1] == res[1][1] > 0


def test_world_size_2_module_arena_hooks_bf16_compress():
    _run(_model_worker, ('bf16',))


def test_world_size_2_replica_sync_and_container_stages():
    res = _run(_sync_worker)
    assert res[0][1] == res[1][1]


def test_world_size_2_comm_report_schema():
    """`comm` of the N > 1 bench line: algorithm / payload / world, per-bucket payload + issue lead + duration, exposed time, bus bandwidth
    -- for both algorithms and both payload types, identical structure on both ranks."""
    for algorithm, compress, el in (('allreduce', None, 4), ('rs_ag', None, 4), ('allreduce', 'bf16', 2), ('rs_ag', 'bf16', 2)):
        res = _run(_comm_worker, (algorithm, compress))
        for rank, rep in res:
            assert set(rep) == {'algorithm', 'payload', 'world', 'steps_traced', 'exposed_ms_per_step', 'allreduce_ms_per_step', 'hidden_fraction',
                                'bus_GBps', 'buckets'}, sorted(rep)
            assert rep['algorithm'] == algorithm and rep['payload'] == ('bf16' if compress else 'fp32') and rep['world'] == 2 and rep['steps_traced'] == 3
            assert [b['elements'] for b in rep['buckets']] == [256, 384, 361]
            assert [b['payload_MB'] for b in rep['buckets']] == [round(n * el / 1e6, 1) for n in (256, 384, 361)]
            assert all(b['allreduce_ms'] > 0 and b['issued_before_backward_end_ms'] >= 0 for b in rep['buckets'])
            assert rep['allreduce_ms_per_step'] > 0 and rep['exposed_ms_per_step'] >= 0 and rep['hidden_fraction'] is not None
        import json
        json.dumps(res[0][1])                                 # the report goes into the bench's JSON line as it is


def test_fit_cuts_buckets_like_the_bench():
    """Trainer.fit and bench.py share ONE cut rule (Trainer.bucket_modules -> DataParallel.equal_byte_cuts over the LAYERS of the model's
    stages): buckets of nearly equal bytes plus a small first one, in forward order, all of them installable."""
    import sys
    for p_ in (ROOT, os.path.join(ROOT, 'open-genie_amd')):
        if p_ not in sys.path:
            sys.path.insert(0, p_)
    from genie.trainer import DataParallel, ParamArena, Trainer

    class Wide(_Toy):
        def __init__(self):
            super().__init__(16)
            self.enc = nn.ModuleList([_Block(16) for _ in range(8)])
            self.dec = nn.ModuleList([_Block(16) for _ in range(8)])

    torch.manual_seed(0)
    m = Wide()
    arena = ParamArena(m)
    picks = Trainer.bucket_modules(arena, m, 4)
    layers = list(m.enc) + [m.quant] + list(m.dec) + [m.late]
    assert all(any(p is l for l in layers) for p in picks) and 2 <= len(picks) <= 5        # small first, small second (1 / 32 of the arena), three equal-byte cuts
    offs = [arena.offset_of(p, m) for p in picks]
    assert offs == sorted(offs) and offs[0] == min(o for o in (arena.offset_of(l, m) for l in layers) if o)      # small first bucket: what finish() reduces unhidden
    assert all(min(abs(o - arena.numel * k) for k in (1 / 32, 1 / 4, 2 / 4, 3 / 4)) <= arena.numel / 8 for o in offs[1:])
    dp = DataParallel(arena.grads)
    dp.install_overlap_hooks(arena, m, picks)                                  # forward order + execution-order arena: accepted
    assert len(dp.buckets) == len(picks) + 1
    assert Trainer(grad_buckets=6).grad_buckets == 6
