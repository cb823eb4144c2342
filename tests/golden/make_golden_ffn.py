"""Golden fixture for the feed-forward options of SpaceTimeAttention (reference attention.py:429-455, misc.py:71-104): a hidden layer with
GELU (`hid_dim`), two hidden layers, and an output width different from the block's (`d_out`: the 1x1x1 `ffn_skip` projection) -- by RUNNING
THE REAL REFERENCE (needs /root/reference; build container only):

    python tests/golden/make_golden_ffn.py   ->  tests/golden/st_block_ffn.pt

Inputs, the reference's state_dict, its output, and its input / parameter gradients for a fixed output gradient."""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

from oracle.ref_import import ref_module  # noqa: E402


def bf16r(t):
    return t.to(torch.bfloat16).float()


def main():
    A = ref_module('module.attention')
    out = {}
    torch.manual_seed(211)
    cases = {'hid48': dict(hid_dim=48), 'hid_40_24': dict(hid_dim=(40, 24)), 'hid48_dout24': dict(hid_dim=48, d_out=24), 'dout40': dict(d_out=40),
             'hid48_bias': dict(hid_dim=48, bias=True)}
    for name, kw in cases.items():
        m = A.SpaceTimeAttention(n_head=2, d_head=16, transpose=True, **kw)
        with torch.no_grad():
            for n_, p in m.named_parameters():
                if 'freq' in n_:
                    continue
                torch.nn.init.normal_(p, std=0.5 if p.dim() < 2 else 0.06)
                if p.dim() >= 2:
                    p.copy_(bf16r(p))                       # matrix-like parameters bf16-representable: both sides multiply identical numbers
        x = bf16r(torch.randn(2, 32, 3, 4, 6)).requires_grad_(True)
        y = m(x)
        dy = bf16r(torch.randn_like(y))
        y.backward(dy)
        out[name] = dict(kw=kw, x=x.detach(), dy=dy, sd={k: v.detach().clone() for k, v in m.state_dict().items()}, out=y.detach(), dx=x.grad.detach(),
                         grads={k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None})
    path = os.path.join(HERE, 'st_block_ffn.pt')
    torch.save(out, path)
    print('wrote', path, {k: tuple(v['out'].shape) for k, v in out.items()}, os.path.getsize(path), 'bytes')


if __name__ == '__main__':
    main()
