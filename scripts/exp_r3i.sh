#!/bin/bash
# round 3, GPU call 10: where do the DynamicsModel step's 27 ms go?  (10.7 ms of kernels)
set -u
OUT=gpurun_out/r3i; mkdir -p $OUT
timeout 300 python scripts/bench_models.py dyn 2>&1 | cut -c1-300 | tail -2
timeout 300 python - <<'PY' 2>&1 | tail -45
import sys, time, os, cProfile, pstats
sys.path[:0] = ['.', 'open-genie_amd']
import torch
from genie import DynamicsModel
from genie.trainer import ParamArena
torch.manual_seed(0)
B = 4
desc = (('space-time_attn', {'n_rep': 8, 'n_head': 8, 'd_head': 64}),)
dyn = DynamicsModel(desc, tok_vocab=2 ** 18, act_vocab=8, embed_dim=512).cuda().train()
tok = torch.randint(0, 2 ** 18, (B, 16, 8, 8), device='cuda'); act = torch.randint(0, 8, (B, 16), device='cuda')
g = torch.Generator().manual_seed(1)
mask = (torch.rand(B, 16, 8, 8, generator=g) < 0.75)
arena = ParamArena(dyn); arena.attach_weight_packs(dyn)
def step(m):
    dyn.compute_loss(tok, act, mask=m).backward(); arena.adamw_step()
for m, name in ((mask, 'host mask'), (mask.cuda(), 'device mask')):
    for _ in range(3): step(m)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): step(m)
    torch.cuda.synchronize(); print(name, (time.perf_counter() - t0) / 10 * 1e3, 'ms/step')
    # host-side time of a step when nothing synchronises
    t0 = time.perf_counter()
    for _ in range(10): step(m)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(name, 'host issue time', (t1 - t0) / 10 * 1e3, 'ms/step; drain', (t2 - t1) * 1e3, 'ms')
pr = cProfile.Profile(); pr.enable()
for _ in range(10): step(mask)
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(28)
PY
