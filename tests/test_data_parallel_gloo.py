"""The N > 1 path on CPU: world_size-2 gloo run of the data-parallel bookkeeping (bucket cuts, bucket readiness
order, mean reduction, batched scalar reduction).  The arithmetic kernels are not involved."""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from util import ROOT


def _worker(rank: int, world: int, port: int, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    for p in (ROOT, os.path.join(ROOT, 'open-genie_amd')):
        if p not in sys.path:
            sys.path.insert(0, p)
    dist.init_process_group('gloo', rank=rank, world_size=world)
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
        assert torch.equal(grads, torch.arange(1000, dtype=torch.float32) * 1.5)
        dp.finish()                              # a second step starts clean
        assert torch.allclose(grads, torch.arange(1000, dtype=torch.float32) * 1.5)
        s = dp.reduce_scalars([torch.tensor(float(rank)), 2.0 * (rank + 1)])
        assert torch.allclose(s, torch.tensor([0.5, 3.0]))
        clips = list(shard_clips(7, rank, world))
        q.put((rank, clips, dp.bytes_reduced))
    finally:
        dist.destroy_process_group()


def test_world_size_2_gloo():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    res = sorted(q.get(timeout=5) for _ in range(2))
    assert res[0][1] == [0, 2, 4, 6] and res[1][1] == [1, 3, 5]
    assert res[0][2] == 2 * 1000 * 4
