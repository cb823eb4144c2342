"""Probe: per-parameter gradient errors of drawn VideoResidualBlock cases (tests/test_gpu_random_geometry.py) against the oracle."""
import sys
import torch
sys.path[:0] = ['/root/repo', '/root/repo/open-genie_amd', '/root/repo/tests']
from test_gpu_random_geometry import draw_resblock, _randomise, rel_rms
from util import bf16_round
from oracle import genie_oracle as O
from genie.module.video import VideoResidualBlock

for i in [int(a) for a in sys.argv[1:]] or [7, 39]:
    cin, cout, groups, causal, down, blur, act, (n, t, h, w) = draw_resblock(i)
    torch.manual_seed(i)
    kw = dict(num_groups=groups, use_causal=causal, downsample=down, use_blur=blur, act_fn=act)
    m = VideoResidualBlock(cin, cout, **kw)
    _randomise(m)
    x = bf16_round(torch.randn(n, cin, t, h, w))
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    names = dict(m.named_parameters())
    sd_req = {k: (v.clone().requires_grad_(True) if k in names else v) for k, v in sd.items()}
    xr = x.clone().requires_grad_(True)
    ref = O.video_residual_block(xr, sd_req, '', cin, cout, **kw)
    dy = bf16_round(torch.randn_like(ref))
    ref.backward(dy)
    m = m.cuda()
    xc = x.cuda().requires_grad_(True)
    out = m(xc)
    out.backward(dy.cuda())
    print(i, cin, cout, kw, (n, t, h, w), 'out', f'{rel_rms(out, ref):.4f}', 'dx', f'{rel_rms(xc.grad, xr.grad):.4f}')
    for name, p in m.named_parameters():
        g, r = p.grad.float().cpu(), sd_req[name].grad
        print(f'   {name:28s} rel {rel_rms(g, r):.4f}  ref rms {r.pow(2).mean().sqrt():.4e}  ours rms {g.pow(2).mean().sqrt():.4e}  ratio of sums {(g.sum() / (r.sum() + 1e-30)).item():.4f}')
