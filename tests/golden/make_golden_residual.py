"""Golden fixture for the VideoResidualBlock options the shipped blueprints do not use but the reference's own tests do (test/test_video.py:130-166):
act_fn 'leaky' / 'relu' / 'gelu', a downsample through BlurPooling3d, GroupNorm groups > 1 handed on to the blur (grouped conv), causal convs --
by RUNNING THE REAL REFERENCE (needs /root/reference; build container only):

    python tests/golden/make_golden_residual.py   ->  tests/golden/residual_options.pt

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
    V = ref_module('module.video')
    out = {}
    torch.manual_seed(307)
    cases = {
        'leaky_down': dict(in_channels=16, out_channels=32, downsample=(2, 4), act_fn='leaky'),                                   # test_video.py:130-146
        'leaky_causal_groups2_down': dict(in_channels=16, out_channels=32, num_groups=2, use_causal=True, act_fn='leaky', downsample=(2, 4)),   # :148-166
        'relu': dict(in_channels=16, out_channels=32, act_fn='relu'),
        'gelu_groups2': dict(in_channels=16, out_channels=16, num_groups=2, act_fn='gelu'),
        'silu_groups2_down': dict(in_channels=16, out_channels=32, num_groups=2, downsample=(1, 2)),
        # causal convs with the non-zero F.pad modes (video.py:560-566 `partial(CausalConv3d, pad_mode=pad_mode)`, :160-164 F.pad): round 4's
        # block ran these convs unpadded -- appended AFTER the cases above so that their random draws are unchanged
        'causal_reflect': dict(in_channels=16, out_channels=32, use_causal=True, pad_mode='reflect'),
        'causal_replicate_same_width': dict(in_channels=16, out_channels=16, use_causal=True, pad_mode='replicate'),
        'causal_circular_leaky': dict(in_channels=16, out_channels=32, use_causal=True, pad_mode='circular', act_fn='leaky'),
    }
    for name, kw in cases.items():
        m = V.VideoResidualBlock(**kw)
        with torch.no_grad():
            for n_, p in m.named_parameters():
                torch.nn.init.normal_(p, std=0.5 if p.dim() < 2 else 0.08)
                if p.dim() >= 2:
                    p.copy_(bf16r(p))
        x = bf16r(torch.randn(2, 16, 4, 8, 8)).requires_grad_(True)
        y = m(x)
        dy = bf16r(torch.randn_like(y))
        y.backward(dy)
        out[name] = dict(kw=kw, x=x.detach(), dy=dy, sd={k: v.detach().clone() for k, v in m.state_dict().items()}, out=y.detach(), dx=x.grad.detach(),
                         grads={k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None})
    path = os.path.join(HERE, 'residual_options.pt')
    torch.save(out, path)
    print('wrote', path, {k: tuple(v['out'].shape) for k, v in out.items()}, os.path.getsize(path), 'bytes')


if __name__ == '__main__':
    main()
