"""GAN critic path on the HIP kernels (-m gpu): 2-D image modules, FrameDiscriminator, hinge GANLoss and the
VideoTokenizer training step with gan_loss_weight > 0 -- against the committed outputs of the real reference (tests/golden/gan.pt)
and against oracle autograd."""
import os

import pytest
import torch

from util import assert_close_bf16, bf16_round

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def rel_rms(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return ((a - b).pow(2).mean().sqrt() / (b.pow(2).mean().sqrt() + 1e-20)).item()


def test_image_modules_vs_reference_outputs():
    from genie.module.discriminator import FrameDiscriminator
    from genie.module.image import ImageResidualBlock
    from genie.module.loss import GANLoss
    g = torch.load(os.path.join(GOLD, 'gan.pt'), weights_only=False)
    for i in range(3):
        e = g[f'image_residual_{i}']
        m = ImageResidualBlock(**e['kw'])
        m.load_state_dict(e['sd'])
        out = m.cuda()(e['x'].cuda())
        assert tuple(out.shape) == tuple(e['out'].shape)
        assert rel_rms(out, e['out']) < 1e-2, (i, rel_rms(out, e['out']))
    e = g['frame_discriminator']
    d = FrameDiscriminator(**e['kw'])
    d.load_state_dict(e['sd'])
    out = d.cuda()(e['x'].cuda())
    assert tuple(out.shape) == (6,)
    assert rel_rms(out, e['out']) < 2e-2, rel_rms(out, e['out'])
    e = g['gan_loss']
    crit = GANLoss(discriminate='frames', num_frames=e['num_frames'], **e['kw'])
    crit.load_state_dict(e['sd'])
    crit = crit.cuda()
    for train_gen, key in ((True, 'gen'), (False, 'dis')):
        loss = crit(e['rec'].cuda(), e['video'].cuda(), train_gen=train_gen, frame_idxs=e[key]['frame_idxs'])
        assert abs(loss.item() - e[key]['loss'].item()) < 2e-2 * abs(e[key]['loss'].item()) + 2e-3, (key, loss.item(), e[key]['loss'].item())


@pytest.mark.parametrize('kw,size', [(dict(inp_channel=16, out_channel=32, num_groups=2, downsample=2), (3, 16, 12, 12)),
                                      (dict(inp_channel=64, out_channel=64, num_groups=8), (2, 64, 16, 16)),
                                      (dict(inp_channel=24, out_channel=None, num_groups=4), (2, 24, 9, 7))])
def test_image_residual_block_forward_backward(kw, size):
    from genie.module.image import ImageResidualBlock
    from oracle import genie_oracle as O
    torch.manual_seed(5)
    m = ImageResidualBlock(**kw)
    with torch.no_grad():
        for n, p in m.named_parameters():
            p.copy_(bf16_round(p) if p.dim() >= 2 else torch.randn_like(p) * 0.3 + (1. if n.endswith('weight') else 0.))
    sd = {k: v.detach().clone().requires_grad_(True) for k, v in m.state_dict().items()}
    m = m.cuda()
    x = bf16_round(torch.randn(size))
    xr = x.clone().requires_grad_(True)
    ref = O.image_residual_block(xr, sd, '', **kw)
    dy = bf16_round(torch.randn_like(ref))
    ref.backward(dy)
    xc = x.cuda().requires_grad_(True)
    out = m(xc)
    assert tuple(out.shape) == tuple(ref.shape) and out.dim() == 4
    assert rel_rms(out, ref) < 1e-2, rel_rms(out, ref)
    out.backward(dy.cuda())
    assert rel_rms(xc.grad, xr.grad) < 3e-2, rel_rms(xc.grad, xr.grad)
    for n, p in m.named_parameters():
        assert rel_rms(p.grad, sd[n].grad) < (0.2 if p.dim() == 1 else 6e-2), (n, rel_rms(p.grad, sd[n].grad))     # per-channel sums of bf16 gradients over few pixels


def test_blur_pool2d_and_registry():
    from genie.module import get_module
    from genie.module.image import BlurPooling2d, ImageResidualBlock, SpaceDownsample
    assert get_module('blur_pool') is BlurPooling2d and get_module('space_downsample') is SpaceDownsample and get_module('image-residual') is ImageResidualBlock
    torch.manual_seed(2)
    m = BlurPooling2d(3, stride=2).cuda()
    x = bf16_round(torch.randn(2, 16, 9, 10))
    ker = m.blur.cpu()[None, None].expand(16, 16, 3, 3)
    ref = torch.nn.functional.conv2d(x, ker, stride=2, padding=m.padding)          # image.py:75-84
    assert_close_bf16(m(x.cuda()), ref, 'blur_pool2d', rms_frac=4e-3)
    # num_groups > 1 (conv2d(groups = g), image.py:75-84): aligned groups as views, any other width on a copy of the slice -- forward and backward
    for c, g, stride in ((16, 2, 2), (12, 3, 2), (10, 5, (1, 2)), (6, 6, 2)):
        mg = BlurPooling2d(3, stride=stride, num_groups=g).cuda()
        xg = bf16_round(torch.randn(2, c, 9, 10))
        xr = xg.clone().requires_grad_(True)
        kg = mg.blur.cpu()[None, None].expand(c, c // g, 3, 3)
        rg = torch.nn.functional.conv2d(xr, kg, stride=stride, padding=mg.padding, groups=g)
        dy = bf16_round(torch.randn_like(rg))
        rg.backward(dy)
        xd = xg.cuda().requires_grad_(True)
        og = mg(xd)
        assert_close_bf16(og, rg, f'blur_pool2d groups={g}', rms_frac=4e-3)
        og.backward(dy.cuda())
        assert_close_bf16(xd.grad, xr.grad, f'blur_pool2d groups={g} dx', rms_frac=4e-3)
    with pytest.raises(ValueError):
        BlurPooling2d(3, num_groups=5).cuda()(torch.randn(1, 16, 8, 8, device='cuda'))


ENC = (('causal-conv3d', {'in_channels': 3, 'out_channels': 32, 'kernel_size': 3}),
       ('video-residual', {'in_channels': 32}),
       ('spacetime_downsample', {'in_channels': 32, 'out_channels': 32, 'kernel_size': 3, 'time_factor': 2, 'space_factor': 2}),
       ('group_norm', {'num_groups': 8, 'num_channels': 32}), ('silu', {}),
       ('causal-conv3d', {'in_channels': 32, 'out_channels': 8, 'kernel_size': 1}))
DEC = (('causal-conv3d', {'in_channels': 8, 'out_channels': 32, 'kernel_size': 3}),
       ('video-residual', {'in_channels': 32}),
       ('depth2spacetime_upsample', {'in_channels': 32, 'kernel_size': 3, 'time_factor': 2, 'space_factor': 2}),
       ('group_norm', {'num_groups': 8, 'num_channels': 32}), ('silu', {}),
       ('causal-conv3d', {'in_channels': 32, 'out_channels': 3, 'kernel_size': 3}))


def test_tokenizer_training_step_with_gan_critic():
    """VideoTokenizer.forward with gan_loss_weight > 0 (reference tokenizer.py:352-387 minus the perceptual term): total loss, the
    generator / critic hinge terms and the gradients of the critic and the decoder against oracle autograd; the random frame
    choice is pinned by stubbing torch.randperm the same way for both."""
    from genie import VideoTokenizer
    from oracle import genie_oracle as O
    disc_kw = dict(inp_size=(32, 32), model_dim=16, dim_mults=(1, 2, 4), down_step=(None, 2, 2), num_groups=2)
    torch.manual_seed(7)
    m = VideoTokenizer(ENC, DEC, disc_kwargs=disc_kw, d_codebook=8, gan_frames_per_batch=2, gan_loss_weight=0.5, perc_loss_weight=0.)
    with torch.no_grad():
        for p in m.parameters():
            if p.dim() >= 2:
                p.copy_(bf16_round(p))
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    sd_req = {k: (v.clone().requires_grad_(True) if v.is_floating_point() else v) for k, v in sd.items()}
    m = m.cuda().train()
    x = bf16_round(torch.randn(2, 3, 4, 32, 32))
    perms = [torch.randperm(4) for _ in range(4)]
    idx_gen, idx_dis = torch.cat([p[:2] for p in perms[:2]]), torch.cat([p[:2] for p in perms[2:]])
    real = torch.randperm
    it = iter(perms)
    torch.randperm = lambda n, **k: next(it).to(k.get('device', 'cpu'))
    try:
        loss_full, aux = m(x.cuda())                                  # the module's own forward: frame choice through torch.randperm
    finally:
        torch.randperm = real
    # the same step spelled out, so that the decoder + critic part can share its input (the quantised latent) with the oracle:
    # an LFQ bit that flips between the bf16 and the fp32 encoder changes the decoder's input, not the decoder (cf. the three-stage
    # scheme of test_tokenizer_training_step_parity)
    from genie import functional as GF
    e = m.encode(x.cuda())
    (qh, _), ql = m.quant(e, transpose=True)
    rec = m.decode(qh)
    rec_loss = GF.mse_loss(rec, x.cuda())
    gen = m.gan_crit(rec, x.cuda(), train_gen=True, frame_idxs=idx_gen)
    dis = m.gan_crit(rec, x.cuda(), train_gen=False, frame_idxs=idx_dis)
    loss = rec_loss + gen * 0.5 + dis * 0.5 + ql
    assert abs(loss.item() - loss_full.item()) < 1e-4 * abs(loss.item()) + 1e-5, (loss.item(), loss_full.item())
    assert abs(aux[1].item() - gen.item()) < 1e-4 + 1e-4 * abs(gen.item()) and abs(aux[2].item() - dis.item()) < 1e-4 + 1e-4 * abs(dis.item())
    loss.backward()
    q_in = qh.detach().float().cpu()
    rec_o = O.tokenizer_decode(q_in, sd_req, DEC)
    rec_l = torch.nn.functional.mse_loss(rec_o, x)
    gen_l = O.gan_loss(rec_o, x, True, idx_gen, sd_req, **disc_kw)
    dis_l = O.gan_loss(rec_o, x, False, idx_dis, sd_req, **disc_kw)
    (rec_l + gen_l * 0.5 + dis_l * 0.5).backward()
    assert abs(rec_loss.item() - rec_l.item()) < 3e-2 * abs(rec_l.item()), (rec_loss.item(), rec_l.item())
    assert abs(gen.item() - gen_l.item()) < 3e-2 * abs(gen_l.item()) + 3e-3, (gen.item(), gen_l.item())
    assert abs(dis.item() - dis_l.item()) < 3e-2 * abs(dis_l.item()) + 3e-3, (dis.item(), dis_l.item())
    # end to end (encoder included) the total still agrees with the oracle's forward
    ref, _, _ = O.tokenizer_forward_gan(x, sd, ENC, DEC, 8, idx_gen, idx_dis, disc_kw, gan_loss_weight=0.5)
    assert abs(loss_full.item() - ref.item()) < 4e-2 * abs(ref.item()), (loss_full.item(), ref.item())
    worst = 0.
    for n, p in m.named_parameters():
        if not (n.startswith('gan_crit') or n.startswith('dec_layers')) or sd_req[n].grad is None or sd_req[n].grad.abs().max() == 0:
            continue
        assert p.grad is not None, n
        r = rel_rms(p.grad, sd_req[n].grad)
        worst = max(worst, r)
        assert r < (0.2 if p.dim() == 1 else 0.12), (n, r)      # 1-D parameters: sums of bf16 gradients over a few thousand pixels
    print('worst critic / decoder gradient rel-RMS', worst)
    assert sum(p.numel() for p in m.gan_crit.parameters()) > 0 and all(p.grad is not None for p in m.gan_crit.parameters())
