"""Golden fixture for the GAN critic path: outputs of the REAL reference's ImageResidualBlock,
FrameDiscriminator and GANLoss (reference genie/module/image.py, discriminator.py, loss.py) on seeded inputs, with the random
frame choice of GANLoss injected through a stubbed torch.randperm.

    python tests/golden/make_golden_gan.py          (build container only: needs /root/reference)
"""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle.ref_import import ref_module  # noqa: E402


def bf16r(t):
    return t.to(torch.bfloat16).float()


def sd_of(m):
    return {k: v.detach().clone() for k, v in m.state_dict().items()}


def round_weights(m):
    with torch.no_grad():
        for p in m.parameters():
            if p.dim() >= 2:
                p.copy_(bf16r(p))


def main():
    I, D, L = ref_module('module.image'), ref_module('module.discriminator'), ref_module('module.loss')
    out = {}
    torch.manual_seed(300)
    for i, kw in enumerate([dict(inp_channel=16, out_channel=32, num_groups=2, downsample=2), dict(inp_channel=16, out_channel=16),
                            dict(inp_channel=16, out_channel=None, num_groups=4)]):
        m = I.ImageResidualBlock(**kw); round_weights(m)
        x = bf16r(torch.randn(3, 16, 12, 12))
        out[f'image_residual_{i}'] = dict(kw=kw, x=x, sd=sd_of(m), out=m(x).detach())
    disc_kw = dict(inp_size=(32, 32), model_dim=16, dim_mults=(1, 2, 4), down_step=(None, 2, 2), num_groups=2)
    d = D.FrameDiscriminator(**disc_kw); round_weights(d)
    img = bf16r(torch.randn(6, 3, 32, 32))
    out['frame_discriminator'] = dict(kw=disc_kw, x=img, sd=sd_of(d), out=d(img).detach())
    g = L.GANLoss(discriminate='frames', num_frames=2, **disc_kw); round_weights(g)
    rec, vid = bf16r(torch.randn(2, 3, 5, 32, 32)), bf16r(torch.randn(2, 3, 5, 32, 32))
    real = torch.randperm
    res = {}
    for train_gen in (True, False):
        perms = [real(5) for _ in range(2)]
        it = iter(perms + [real(5) for _ in range(8)])          # pick_frames draws (and discards) more, utils.py:45-49
        torch.randperm = lambda n, **k: next(it)
        try:
            loss = g(rec, vid, train_gen=train_gen)
        finally:
            torch.randperm = real
        res[train_gen] = dict(frame_idxs=torch.cat([p[:2] for p in perms]), loss=loss.detach())
    out['gan_loss'] = dict(kw=disc_kw, num_frames=2, rec=rec, video=vid, sd=sd_of(g), gen=res[True], dis=res[False])
    path = os.path.join(HERE, 'gan.pt')
    torch.save(out, path)
    print(f'gan.pt: {os.path.getsize(path) / 1024:.0f} KiB')


if __name__ == '__main__':
    main()
