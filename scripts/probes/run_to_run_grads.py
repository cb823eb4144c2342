"""Probe: three eager runs of the small tokenizer step of tests/test_gpu_graph.py (default stream twice, a side stream once) in deterministic
mode; prints the loss sequences and which gradients differ run to run.  Round 6 used it to find the conv bias gradients (epilogue atomics)
as the seed of the graph-replay test's fresh-box failure; with the fixed-order bias sum it prints no differing gradient."""
import sys, os, torch
sys.path[:0]=['/root/repo','/root/repo/open-genie_amd','/root/repo/tests']
from genie import conv as gconv, functional as GF
from genie.trainer import ParamArena
from test_gpu_trainer import _model
g = torch.Generator(device='cuda').manual_seed(7)
xs = [torch.randn(2, 3, 4, 16, 16, device='cuda', generator=g) for _ in range(4)]
seq = [xs[0], xs[0], xs[0], xs[1]]
gconv.set_deterministic(True)
def run(tag):
    m = _model(); arena = ParamArena(m); arena.attach_weight_packs(m); arena.set_graph_hyperparameters(1e-3, 0.01)
    names = [n for n, _ in m.named_parameters()]
    grads = []
    losses = []
    for x in seq:
        loss, _ = m(x); loss.backward()
        grads.append({n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None})
        arena.adamw_step(graph_safe=True); losses.append(loss.item())
    print(tag, ' '.join(f'{v:.6f}' for v in losses), flush=True)
    return grads
a = run('A default stream, first in process')
b = run('B default stream, second        ')
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    c = run('C side stream                   ')
torch.cuda.current_stream().wait_stream(side)
for step in range(4):
    for other, tag in ((b, 'B'), (c, 'C')):
        bad = [(n, (a[step][n] - other[step][n]).abs().max().item()) for n in a[step] if not torch.equal(a[step][n], other[step][n])]
        print('step', step, 'A vs', tag, 'differing grads:', len(bad), bad[:4])
