import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'open-genie_amd')):
    sys.path.insert(0, p)
import torch
from genie import conv as gconv, functional as GF
from genie.cl import to_cl
torch.manual_seed(0)
n, t, h, w = [int(v) for v in os.environ.get('DBG_SIZE', '1,4,8,64').split(',')]
cout = 3
spec = gconv.causal_spec(128, cout, (3, 3, 3))
x = torch.randn(n, 128, t, h, w).bfloat16().float()
xc = to_cl(x.cuda())
wfull = (torch.randn(cout, 128, 3, 3, 3) / 10).bfloat16().float()
op = GF.ConvOp(spec)
def run(wt, b):
    out = GF.conv3d(xc, wt.cuda(), b.cuda(), op).float().cpu()
    gen = gconv.conv_forward(xc, gconv.pack_weight_fwd(wt.cuda(), spec), b.cuda(), spec).float().cpu()
    return out, gen
b0 = torch.zeros(cout)
out, gen = run(wfull, b0)
print('full: max err', (out - gen).abs().max().item(), 'ref max', gen.abs().max().item())
e = (out - gen).abs()
print(' err by co', e.amax((0, 2, 3, 4)).tolist())
print(' err by t', e.amax((0, 1, 3, 4)).tolist())
print(' err by h', e.amax((0, 1, 2, 4)).tolist())
print(' err by w', [round(v, 2) for v in e.amax((0, 1, 2, 3)).tolist()])
for dt in range(3):
    for dh in range(3):
        for dw in range(3):
            wt = torch.zeros_like(wfull); wt[:, :, dt, dh, dw] = wfull[:, :, dt, dh, dw]
            out, gen = run(wt, b0)
            e = (out - gen).abs()
            print('tap', dt, dh, dw, 'err', round(e.max().item(), 3), 'of', round(gen.abs().max().item(), 3), 'by co', [round(v, 2) for v in e.amax((0, 2, 3, 4)).tolist()],
                  'bad cols', (e.amax((0, 1, 2, 3)) > 0.05).nonzero().flatten().tolist()[:8], 'bad t', (e.amax((0, 1, 3, 4)) > 0.05).nonzero().flatten().tolist())
