import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'open-genie_amd')]
import torch
from genie import _hip, conv as gconv
from genie.conv import same_spec, conv_forward, conv_dgrad, pack_weight_fwd, pack_weight_bwd
from scripts.microbench import timeit, rand_cl
B = 8
for (c, size) in [(128, (16, 64, 64)), (256, (16, 32, 32))]:
    spec = same_spec(c, c, (3, 3, 3))
    x = rand_cl(B, c, *size)
    wt = (torch.randn(c, c, 3, 3, 3, device='cuda') * 0.05).contiguous(memory_format=torch.channels_last_3d)
    wf, wb = pack_weight_fwd(wt, spec), pack_weight_bwd(wt, spec)
    y = conv_forward(x, wf, None, spec)
    dyr = rand_cl(B, c, *size)
    for rep in range(3):
        print(c, 'fwd      ', round(timeit(lambda: conv_forward(x, wf, None, spec), 20), 4))
        print(c, 'dgrad(y) ', round(timeit(lambda: conv_dgrad(y, wb, spec, size), 20), 4))
        print(c, 'dgrad(rnd)', round(timeit(lambda: conv_dgrad(dyr, wb, spec, size), 20), 4))
        print(c, 'fwd(wb)  ', round(timeit(lambda: conv_forward(x, wb, None, spec), 20), 4))
    print(torch.isfinite(y.float()).all().item(), y.float().abs().max().item(), y.float().std().item())
