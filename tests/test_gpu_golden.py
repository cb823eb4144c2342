"""The HIP path compared DIRECTLY with the committed outputs of the real reference (tests/golden/*.pt, produced by
tests/golden/make_golden*.py in the build container) -- no oracle in between (-m gpu).  Tolerances: one bf16 rounding per
operator (util.assert_close_bf16), relative-RMS bounds for multi-layer models, bit-exact for LFQ indices."""
import os

import pytest
import torch

from util import assert_close_bf16

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def load(name):
    return torch.load(os.path.join(GOLD, name), weights_only=False)


def rel_rms(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return ((a - b).pow(2).mean().sqrt() / (b.pow(2).mean().sqrt() + 1e-20)).item()


def test_ops_vs_reference_outputs():
    from genie.module.norm import AdaptiveGroupNorm
    from genie.module.video import CausalConv3d, DepthToSpaceTimeUpsample, VideoResidualBlock
    g = load('ops.pt')
    for i in range(4):
        e = g[f'causal_conv3d_{i}']
        m = CausalConv3d(e['cin'], e['cout'], e['kernel'], stride=e['stride'])
        m.load_state_dict({'conv3d.weight': e['weight'], 'conv3d.bias': e['bias']})
        assert_close_bf16(m.cuda()(e['x'].cuda()), e['out'], f'causal_conv3d_{i}')
    for i in range(4):
        e = g[f'video_residual_{i}']
        m = VideoResidualBlock(**e['kw'])
        m.load_state_dict(e['sd'])
        out = m.cuda()(e['x'].cuda())
        assert rel_rms(out, e['out']) < 1e-2, (i, rel_rms(out, e['out']))
    e = g['depth2spacetime']
    m = DepthToSpaceTimeUpsample(16, out_channels=8, time_factor=2, space_factor=2, kernel_size=3)
    m.load_state_dict(e['sd'])
    assert_close_bf16(m.cuda()(e['x'].cuda()), e['out'], 'depth2spacetime')
    e = g['adagn']
    m = AdaptiveGroupNorm(6, 4, 16)
    m.load_state_dict(e['sd'])
    assert_close_bf16(m.cuda()(e['x'].cuda(), e['cond'].cuda()), e['out'], 'adagn', rel=2 ** -6, rms_frac=4e-3)


def test_lfq_indices_vs_reference_outputs():
    from genie.module.quantization import LookupFreeQuantization
    g = load('lfq.pt')
    for name, e in g.items():
        if 'eval_idx' not in e:
            continue
        m = LookupFreeQuantization(e['d'], e['n'], input_dim=e['inp'])
        m.load_state_dict(e['sd'], strict=False)
        m = m.cuda().eval()
        (out, idx), loss = m(e['x'].cuda(), transpose=True)
        assert loss is None
        if e['inp'] == e['d'] * e['n']:                 # no projection: the operator boundary -> bit-exact
            assert torch.equal(idx.cpu(), e['eval_idx']), name
            assert torch.equal(out.float().cpu(), e['eval_out']), name
        else:                                           # fp32 Linear in front: signs can flip only where |z| ~ 0
            assert (idx.cpu() == e['eval_idx']).float().mean() > 0.95, name
        if 'train_loss' in e and e['inp'] == e['d'] * e['n']:
            m.train()
            (_, it), lt = m(e['x'].cuda(), transpose=True)
            assert torch.equal(it.cpu(), e['train_idx'])
            assert abs(lt.item() - e['train_loss'].item()) < 1e-5 + 1e-5 * abs(e['train_loss'].item()), (name, lt.item(), e['train_loss'].item())


def test_st_block_vs_reference_outputs():
    """SpaceTimeAttention with the reference's shipped head shape (4 heads x 16, genie/__init__.py:15-50): channels-last,
    channels-first and temporally conditioned."""
    from genie.module.attention import SpaceTimeAttention
    g = load('st_block.pt')
    for name, e in g.items():
        m = SpaceTimeAttention(n_head=4, d_head=16, transpose=e['transpose'], **e['kw'])
        m.load_state_dict(e['sd'])
        m = m.cuda()
        out = m(e['x'].cuda(), cond=(None, e['cond'].cuda())) if e['cond'] is not None else m(e['x'].cuda())
        assert tuple(out.shape) == tuple(e['out'].shape)
        assert rel_rms(out, e['out']) < 1.5e-2, (name, rel_rms(out, e['out']))


def test_st_block_feed_forward_options_vs_reference():
    """SpaceTimeAttention(hid_dim=..., d_out=...) -- hidden Conv3d layers with GELU between them (reference misc.py:86-98) and the 1x1x1
    skip projection when the width changes (attention.py:453) -- against outputs AND gradients of the real reference modules
    (tests/golden/make_golden_ffn.py).  Round 3 raised NotImplementedError for these constructor options."""
    from genie.module.attention import SpaceTimeAttention
    g = load('st_block_ffn.pt')
    assert set(g) >= {'hid48', 'hid_40_24', 'hid48_dout24', 'dout40', 'hid48_bias'}
    for name, e in g.items():
        m = SpaceTimeAttention(n_head=2, d_head=16, transpose=True, **e['kw'])
        assert sorted(m.state_dict()) == sorted(e['sd']), name          # the reference's keys (ffn.1.net.<i>.0.weight, ffn_skip.weight / .bias)
        m.load_state_dict(e['sd'])
        m = m.cuda()
        x = e['x'].cuda().requires_grad_(True)
        out = m(x)
        assert tuple(out.shape) == tuple(e['out'].shape), name
        assert rel_rms(out, e['out']) < 1.5e-2, (name, rel_rms(out, e['out']))
        out.backward(e['dy'].cuda())
        assert rel_rms(x.grad, e['dx']) < 2e-2, (name, 'dx', rel_rms(x.grad, e['dx']))
        for k, p in m.named_parameters():
            if k in e['grads']:
                assert p.grad is not None, (name, k)
                assert rel_rms(p.grad, e['grads'][k]) < 2e-2, (name, k, rel_rms(p.grad, e['grads'][k]))


def test_residual_block_options_vs_reference():
    """VideoResidualBlock with the options the reference's own tests exercise (test/test_video.py:130-166) and round 3 refused: act_fn 'leaky' /
    'relu' / 'gelu', GroupNorm groups > 1 handed on to BlurPooling3d (the grouped blur: per-group channel sums), causal convs with a
    downsample -- outputs AND gradients of the real reference modules (tests/golden/make_golden_residual.py)."""
    from genie.module.video import VideoResidualBlock
    g = load('residual_options.pt')
    assert set(g) >= {'leaky_down', 'leaky_causal_groups2_down', 'relu', 'gelu_groups2', 'silu_groups2_down',
                      'causal_reflect', 'causal_replicate_same_width', 'causal_circular_leaky'}      # (the last three: F.pad modes inside the block)
    for name, e in g.items():
        m = VideoResidualBlock(**e['kw'])
        assert sorted(m.state_dict()) == sorted(e['sd']), (name, sorted(m.state_dict()), sorted(e['sd']))
        m.load_state_dict(e['sd'])
        m = m.cuda()
        x = e['x'].cuda().requires_grad_(True)
        out = m(x)
        assert tuple(out.shape) == tuple(e['out'].shape), name
        assert rel_rms(out, e['out']) < 1.5e-2, (name, rel_rms(out, e['out']))
        out.backward(e['dy'].cuda())
        assert rel_rms(x.grad, e['dx']) < 2e-2, (name, 'dx', rel_rms(x.grad, e['dx']))
        for k, p in m.named_parameters():
            if k in e['grads']:
                assert p.grad is not None, (name, k)
                # 3 %: ReLU / LeakyReLU have a kink -- an activation input that the bf16 GroupNorm pass rounds across zero switches its gradient
                # on or off, and these are sums over only 2 x 4 x 8 x 8 positions (measured worst: 2.1 % on a GroupNorm bias under ReLU)
                assert rel_rms(p.grad, e['grads'][k]) < 3e-2, (name, k, rel_rms(p.grad, e['grads'][k]))
    with pytest.raises(ValueError):
        VideoResidualBlock(16, act_fn='tanh')


def test_gelu_kernel_matches_torch():
    """genie_gelu_fwd / _bwd == nn.GELU() (exact erf form) and its autograd on a channels-last video tensor, incl. a channel count that is
    not a multiple of 8 (pad channels stay zero: gelu(0) = 0)."""
    from genie import functional as GF
    torch.manual_seed(9)
    for c in (16, 20):
        x = (torch.randn(2, c, 3, 5, 4) * 2).to(torch.bfloat16).float()
        xr = x.clone().requires_grad_(True)
        ref = torch.nn.functional.gelu(xr)
        dy = torch.randn_like(ref).to(torch.bfloat16).float()
        ref.backward(dy)
        xc = x.cuda().requires_grad_(True)
        y = GF.gelu(xc)
        y.backward(dy.cuda())
        assert (y.float().cpu() - ref.detach()).abs().max().item() <= 2 ** -7 * ref.abs().max().item() + 1e-3
        assert (xc.grad.float().cpu() - xr.grad).abs().max().item() <= 2 ** -7 * xr.grad.abs().max().item() + 1e-3


def test_dynamics_vs_reference_outputs():
    from genie.dynamics import DynamicsModel
    g = load('dynamics_small.pt')
    m = DynamicsModel(g['desc'], tok_vocab=g['tok_vocab'], act_vocab=g['act_vocab'], embed_dim=g['embed_dim'])
    m.load_state_dict(g['sd'])
    m = m.cuda()
    logits, last = m(g['tokens'].cuda(), g['act'].cuda())
    assert rel_rms(logits, g['logits']) < 2e-2, rel_rms(logits, g['logits'])
    loss = m.compute_loss(g['tokens'].cuda(), g['act'].cuda(), mask=g['mask'].cuda())
    assert abs(loss.item() - g['loss'].item()) < 2e-2 * abs(g['loss'].item()) + 1e-3
    assert m.get_schedule(10, (16, 16)).tolist() == g['schedule_linear_10_16x16']
    assert m.get_schedule(7, (8, 8), 'cosine').tolist() == g['schedule_cosine_7_8x8']
    assert m.get_schedule(7, (8, 8), 'arccos').tolist() == g['schedule_arccos_7_8x8']


def test_tokenizer_small_vs_reference_outputs():
    from genie import VideoTokenizer
    g = load('tokenizer_small.pt')
    m = VideoTokenizer(g['enc_desc'], g['dec_desc'], d_codebook=g['d_codebook'], gan_loss_weight=0., perc_loss_weight=0.)
    m.load_state_dict(g['sd'], strict=False)
    m = m.cuda().train()
    x = g['x'].cuda()
    enc = m.encode(x)
    assert rel_rms(enc, g['enc']) < 2e-2, rel_rms(enc, g['enc'])
    rec = m.decode(g['quant'].cuda())
    assert rel_rms(rec, g['rec']) < 3e-2, rel_rms(rec, g['rec'])
    q, idx = m.tokenize(x)
    agree = (idx.cpu() == g['idx']).float().mean().item()
    assert agree > 0.97, agree                         # end to end through bf16 layers: flips only at |z| ~ 0 (operator-level test is exact)
    loss, aux = m(x)
    assert abs(loss.item() - g['rfwd_loss'].item()) < 3e-2 * abs(g['rfwd_loss'].item()), (loss.item(), g['rfwd_loss'].item())


def test_reference_test_dynamics_config():
    """SURVEY.md 8c model-level procedure: the reference's own test_dynamics.py configuration (4 x ST(4 x 16), 16 tokens, 4 actions,
    (2, 10, 16, 16) tokens; `n_embd` dropped as in every runnable form of the reference) -- shapes of test_dynamics.py:37-81 plus the
    numbers of the real reference (last-frame logits of the committed generate fixture)."""
    from genie.dynamics import DynamicsModel
    g = load('dynamics_generate.pt')['linear5']
    m = DynamicsModel(g['desc'], tok_vocab=16, act_vocab=4, embed_dim=64)
    m.load_state_dict(g['sd'])
    m = m.cuda()
    tok, act = g['tokens'].cuda(), g['act'].cuda()
    logits, last = m(tok, act)
    assert tuple(logits.shape) == (2, 10, 16, 16, 16) and tuple(last.shape) == (2, 16, 16, 16)
    tok1 = torch.cat([tok, torch.zeros(2, 1, 16, 16, dtype=tok.dtype, device='cuda')], 1)
    act1 = torch.cat([act, torch.zeros(2, 1, dtype=act.dtype, device='cuda')], 1)
    _, last1 = m(tok1, act1)
    assert rel_rms(last1, g['last_logits']) < 2e-2, rel_rms(last1, g['last_logits'])
    loss = m.compute_loss(tok, act)
    assert loss.shape == () and loss.item() >= 0
    gen = m.generate(tok, act, steps=5)
    assert tuple(gen.shape) == (2, 11, 16, 16)
    sch = m.get_schedule(10, (16, 16))
    assert tuple(sch.shape) == (10,) and int(sch.sum()) == 256
