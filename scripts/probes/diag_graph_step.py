import sys, os, torch
sys.path[:0]=['/root/repo','/root/repo/open-genie_amd','/root/repo/tests']
from genie import conv as gconv, functional as GF
from genie.graph import GraphedTrainStep
from genie.trainer import ParamArena
from test_gpu_trainer import _model
g = torch.Generator(device='cuda').manual_seed(7)
xs = [torch.randn(2, 3, 4, 16, 16, device='cuda', generator=g) for _ in range(4)]
seq = [xs[0], xs[0], xs[0], xs[1], xs[2], xs[3], xs[1]]
gconv.set_deterministic(True)
def eager():
    m = _model(); arena = ParamArena(m); arena.attach_weight_packs(m); arena.set_graph_hyperparameters(1e-3, 0.01)
    out = []
    for x in seq:
        loss, _ = m(x); loss.backward(); arena.adamw_step(graph_safe=True); out.append(loss.item())
    return out
def graphed():
    m2 = _model(); arena2 = ParamArena(m2); arena2.attach_weight_packs(m2)
    gs = GraphedTrainStep(m2, arena2, xs[0], lr=1e-3, weight_decay=0.01, warmup=2)
    out = [gs.loss.item()]
    for x in seq[3:]:
        out.append(gs(x).item())
    return out
order = sys.argv[1] if len(sys.argv) > 1 else 'eg'
for what in order:
    r = eager() if what == 'e' else graphed()
    print(('eager ' if what == 'e' else 'graph ') + ' '.join(f'{v:.6f}' for v in (r[2:] if what == 'e' else r)), flush=True)
