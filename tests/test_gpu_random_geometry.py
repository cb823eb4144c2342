"""Seeded random sweeps over the geometry the public modules accept (-m gpu): `CausalConv3d` (channels, kernel, stride, dilation, frame / image sizes, with
and without bias, both spatial paddings) and `SpaceTimeAttention` (heads, head width, T x H x W, layout, temporal condition) against the oracle -- outputs,
input gradients and every parameter gradient.  The hand-picked cases of test_gpu_kernels.py / test_gpu_attention.py pin each kernel family on the shapes it was
written for; these draws cross the DISPATCH predicates (narrow / kw-triple / lean / pointwise / windowed / packed / general) at shapes nobody picked.
The draws are a pure function of the case index, so a failure reproduces from its parameter id."""
import random

import pytest
import torch

from util import assert_close_bf16, bf16_round, report

pytestmark = pytest.mark.gpu


def rel_rms(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return ((a - b).pow(2).mean().sqrt() / (b.pow(2).mean().sqrt() + 1e-20)).item()


def draw_conv(i):
    r = random.Random(1000 + i)
    chans = [1, 2, 3, 4, 8, 10, 16, 18, 24, 64, 72, 128, 136, 256]
    cin, cout = r.choice(chans), r.choice(chans)
    kt = r.choice([1, 1, 2, 3, 3, 3, 5])
    ks = r.choice([1, 3, 3, 3, 5])
    kernel = (kt, ks, ks) if r.random() < 0.8 else (kt, r.choice([1, 3]), r.choice([1, 3, 5]))
    stride = r.choice([(1, 1, 1)] * 5 + [(1, 2, 2), (2, 2, 2), (2, 1, 1), (1, 1, 2)])
    dilation = r.choice([(1, 1, 1)] * 4 + [(2, 1, 1), (1, 2, 2)])
    w = r.choice([3, 5, 7, 8, 11, 16, 20, 32, 32, 64, 64, 128])
    h = r.choice([2, 3, 4, 6, 8, 9, 16]) if w >= 32 else r.choice([3, 4, 5, 8, 12, 17])
    t = r.choice([1, 2, 3, 4, 6, 9])
    n = r.choice([1, 1, 2, 3])
    if stride[0] == 2 and kt == 1:
        t = max(t, 3)                                    # the negative causal pad of this combination crops a frame
    # keep the oracle cheap and the effective kernel inside the padded image
    while n * t * h * w * max(cin, cout) > 6_000_000:
        n, t = max(1, n - 1), max(1, t - 1)
        if n == 1 and t == 1:
            h = max(2, h // 2)
    tp = (kt - 1) * dilation[0] + (1 - stride[0])         # causal front padding (negative: crops)
    while t + tp < dilation[0] * (kt - 1) + 1:
        t += 1                                            # the padded clip must hold one (dilated) kernel in time
    bias = r.random() < 0.7
    return cin, cout, kernel, stride, dilation, (n, t, h, w), bias


@pytest.mark.parametrize('i', range(96))
def test_causal_conv3d_random_geometry(i):
    from oracle import genie_oracle as O
    from genie.module.video import CausalConv3d
    cin, cout, kernel, stride, dilation, (n, t, h, w), bias = draw_conv(i)
    eff = [dilation[a] * (kernel[a] - 1) + 1 for a in range(3)]
    if eff[1] > h + 2 * ((kernel[1] - 1) // 2) or eff[2] > w + 2 * ((kernel[2] - 1) // 2):
        pytest.skip('dilated kernel larger than the padded image')
    torch.manual_seed(i)
    m = CausalConv3d(cin, cout, kernel, stride=stride, dilation=dilation, bias=bias)
    fan = cin * kernel[0] * kernel[1] * kernel[2]
    with torch.no_grad():
        m.conv3d.weight.copy_(bf16_round(torch.randn_like(m.conv3d.weight) / fan ** 0.5))
        if bias:
            m.conv3d.bias.copy_(torch.randn_like(m.conv3d.bias))
    wt = m.conv3d.weight.detach().clone().requires_grad_(True)
    bt = m.conv3d.bias.detach().clone().requires_grad_(True) if bias else None
    x = bf16_round(torch.randn(n, cin, t, h, w))
    xr = x.clone().requires_grad_(True)
    ref = O.causal_conv3d(xr, wt, bt, stride=stride, dilation=dilation)
    if ref.numel() == 0:
        pytest.skip('empty output')
    dy = bf16_round(torch.randn_like(ref))
    ref.backward(dy)
    m = m.cuda()
    xc = x.cuda().requires_grad_(True)
    out = m(xc)
    assert tuple(out.shape) == tuple(ref.shape), (tuple(out.shape), tuple(ref.shape))
    assert_close_bf16(out, ref, f'conv fwd {cin}->{cout} k{kernel} s{stride} d{dilation} @{(n, t, h, w)}')
    out.backward(dy.cuda())
    errs = {'dx': rel_rms(xc.grad, xr.grad), 'dw': rel_rms(m.conv3d.weight.grad, wt.grad)}
    if bias:
        errs['db'] = rel_rms(m.conv3d.bias.grad, bt.grad)
    assert errs['dx'] < 1e-2 and errs['dw'] < 5e-3 and errs.get('db', 0.0) < 5e-3, (errs, cin, cout, kernel, stride, dilation, (n, t, h, w))
    report('random_conv', i=i, cin=cin, cout=cout, kernel=kernel, stride=stride, dilation=dilation, size=(n, t, h, w), **errs)


def draw_attn(i):
    r = random.Random(5000 + i)
    n_head = r.choice([1, 2, 4, 8])
    d_head = r.choice([8, 16, 32, 64, 64, 64, 128])
    while n_head * d_head > 512:
        n_head //= 2
    t = r.choice([1, 2, 3, 5, 8, 12, 16, 17, 33])
    h, w = r.choice([(1, 1), (2, 3), (3, 3), (4, 4), (5, 7), (8, 8), (9, 9), (12, 12), (16, 16)])
    if t * h * w * n_head * d_head > 1_500_000:
        t = max(1, 1_500_000 // (h * w * n_head * d_head))
    transpose = r.random() < 0.5
    cond = r.random() < 0.3
    b = r.choice([1, 2, 3])
    return n_head, d_head, (b, t, h, w), transpose, cond


@pytest.mark.parametrize('i', range(64))
def test_space_time_attention_random_geometry(i):
    from oracle import genie_oracle as O
    from genie.module.attention import SpaceTimeAttention
    n_head, d_head, (b, t, h, w), transpose, cond = draw_attn(i)
    c = n_head * d_head
    torch.manual_seed(i)
    kw = {'time_attn_kw': {'key_dim': 8}} if cond else {}
    m = SpaceTimeAttention(n_head=n_head, d_head=d_head, transpose=transpose, **kw)
    with torch.no_grad():
        for name, p in m.named_parameters():
            if 'freq' in name:
                continue
            if name.endswith('attn.norm.weight'):
                # a small LayerNorm gain in front of q = k = v: with gamma ~ 1 the self-score |u|^2 * scale dwarfs the cross scores and the softmax
                # is the identity -- a sweep through which a wrong mask would pass; here the weights are spread over the keys
                p.copy_(torch.randn_like(p) * 0.1 + 0.45)
            elif p.dim() < 2:
                p.copy_(torch.randn_like(p) * 0.3 + (1.0 if name.endswith('weight') else 0.0))
            else:
                p.copy_(bf16_round(torch.randn_like(p) * (0.02 if p.dim() == 5 else 0.3)))
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    m = m.cuda()
    x = bf16_round(torch.randn(b, c, t, h, w) if transpose else torch.randn(b, t, h, w, c))
    cd = bf16_round(torch.randn(b, t, 8)) if cond else None
    sd_req = {k: (v.clone().requires_grad_(True) if 'freq' not in k else v) for k, v in sd.items()}
    xr = x.clone().requires_grad_(True)
    cr = cd.clone().requires_grad_(True) if cond else None
    ref = O.space_time_block(xr, sd_req, '', n_head, d_head, transpose=transpose, cond=(None, cr) if cond else None)
    dy = bf16_round(torch.randn_like(ref))
    ref.backward(dy)
    xc = x.cuda().requires_grad_(True)
    cc = cd.cuda().requires_grad_(True) if cond else None
    out = m(xc, cond=(None, cc)) if cond else m(xc)
    assert tuple(out.shape) == tuple(ref.shape)
    e_out = rel_rms(out, ref)
    assert e_out < 1.5e-2, (e_out, n_head, d_head, (b, t, h, w), transpose, cond)
    out.backward(dy.cuda())
    e_dx = rel_rms(xc.grad, xr.grad)
    assert e_dx < 5e-2, (e_dx, n_head, d_head, (b, t, h, w), transpose, cond)
    worst = ('', 0.0)
    for name, p in m.named_parameters():
        if 'freq' in name:
            continue
        assert p.grad is not None, name
        want = sd_req[name].grad
        if want.float().pow(2).mean().sqrt().item() < 1e-7:
            # T = 1 with a temporal condition: one key, softmax = 1 -- the gradients of everything in front of the scores (LayerNorm, to_k) are exactly
            # zero in the reference; here they are the rounding noise of p - 1
            assert p.grad.float().pow(2).mean().sqrt().item() < 1e-4, name
            continue
        e = rel_rms(p.grad, want)
        worst = max(worst, (name, e), key=lambda v: v[1])
    assert worst[1] < 6e-2, (worst, n_head, d_head, (b, t, h, w), transpose, cond)
    if cond:
        assert rel_rms(cc.grad, cr.grad) < 6e-2
    report('random_st_attention', i=i, n_head=n_head, d_head=d_head, size=(b, t, h, w), transpose=transpose, cond=cond, out=e_out, dx=e_dx,
           worst_param=worst[0], worst_param_err=worst[1])


def draw_gn(i):
    r = random.Random(9000 + i)
    c = r.choice([8, 16, 24, 40, 64, 72, 128, 136, 256, 512])
    divs = [g for g in (1, 2, 3, 4, 5, 8, 9, 16, 17, 32, 64) if c % g == 0]
    g = r.choice(divs + [1, 1, c])
    n = r.choice([1, 2, 3, 5])
    thw = (r.choice([1, 2, 3, 4, 8]), r.choice([1, 3, 4, 8, 13, 16, 32]), r.choice([1, 2, 5, 8, 16, 31, 64]))
    while n * c * thw[0] * thw[1] * thw[2] > 8_000_000:
        thw = (max(1, thw[0] // 2), thw[1], thw[2])
        n = max(1, n - 1)
    return n, c, g, thw, r.random() < 0.4, r.random() < 0.6


@pytest.mark.parametrize('i', range(48))
def test_groupnorm_random_geometry(i):
    """GroupNorm (+ adaptive scale / shift, + SiLU) through the C ABI on drawn (N, C, groups, T x H x W): forward, dx, affine and adaptive gradients."""
    from oracle import genie_oracle as O
    from genie import _hip, cl
    n, c, g, thw, ada, act = draw_gn(i)
    torch.manual_seed(i)
    x = bf16_round(torch.randn(n, c, *thw) * 1.5 + 0.3)
    dy = bf16_round(torch.randn(n, c, *thw))
    gamma, beta = torch.randn(c), torch.randn(c)
    ada_s = torch.randn(n, c) if ada else None
    ada_b = torch.randn(n, c) if ada else None
    leaves = [t.clone().requires_grad_(True) for t in (x, gamma, beta)] + ([ada_s.clone().requires_grad_(True), ada_b.clone().requires_grad_(True)] if ada else [])
    ref = O.group_norm(leaves[0], g, leaves[1], leaves[2])
    if ada:
        ref = ref * leaves[3].reshape(n, c, 1, 1, 1) + leaves[4].reshape(n, c, 1, 1, 1)
    if act:
        ref = O.silu(ref)
    ref.backward(dy)
    lib = _hip.load_library()
    P = _hip.ptr
    xc, dyc = cl.to_cl(x.cuda()), cl.to_cl(dy.cuda())
    y, dx = cl.empty_like_cl(xc), cl.empty_like_cl(xc)
    cp, npix = cl.pitch_of(xc), thw[0] * thw[1] * thw[2]
    dev = lambda t: None if t is None else t.cuda().contiguous()
    g_, b_, as_, ab_ = dev(gamma), dev(beta), dev(ada_s), dev(ada_b)
    mean, rstd = torch.empty(n * g, device='cuda'), torch.empty(n * g, device='cuda')
    ws = torch.empty(lib.genie_groupnorm_ws_floats(n, c, g), device='cuda')
    _hip.check(lib.genie_groupnorm_fwd(P(xc), P(y), n, npix, c, cp, g, P(g_), P(b_), P(as_), P(ab_), 1e-5, int(act), P(mean), P(rstd), P(ws), _hip.stream_ptr()), 'gn fwd')
    tag = f'gn N={n} C={c} G={g} thw={thw} ada={ada} act={act}'
    assert_close_bf16(y, ref, tag + ' fwd')
    if npix * (c // g) > 1:                               # (a one-element group has zero variance: the gradient is rounding noise on both sides)
        dgamma, dbeta = torch.zeros(c, device='cuda'), torch.zeros(c, device='cuda')
        das = torch.empty(n, c, device='cuda') if ada else None
        dab = torch.empty(n, c, device='cuda') if ada else None
        _hip.check(lib.genie_groupnorm_bwd(P(xc), P(dyc), P(dx), n, npix, c, cp, g, P(g_), P(b_), P(as_), P(ab_), int(act), P(mean), P(rstd),
                                           P(dgamma), P(dbeta), P(das), P(dab), P(ws), _hip.stream_ptr()), 'gn bwd')
        assert_close_bf16(dx, leaves[0].grad, tag + ' dx', rms_frac=5e-3)
        for got, want, nm in [(dgamma, leaves[1].grad, 'dgamma'), (dbeta, leaves[2].grad, 'dbeta')] + ([(das, leaves[3].grad, 'dada_s'), (dab, leaves[4].grad, 'dada_b')] if ada else []):
            torch.testing.assert_close(got.cpu(), want, rtol=2e-3, atol=2e-3 * want.abs().max().item(), msg=tag + ' ' + nm)
    report('random_groupnorm', i=i, N=n, C=c, G=g, thw=thw, ada=ada, act=act)


def _randomise(m, scale=0.1):
    with torch.no_grad():
        for name, p in m.named_parameters():
            if p.dim() >= 2:
                p.copy_(bf16_round(torch.randn_like(p) * scale / max(1.0, (p[0].numel() / 27.0) ** 0.5)))
            else:
                p.copy_(torch.randn_like(p) * 0.3 + (1.0 if 'weight' in name and p.dim() == 1 and 'conv' not in name else 0.0))


def _compare_module(m, ref_fn, x, tag, out_tol=2e-2, dx_tol=3e-2, p_tol=3e-2, rounding=None):
    """m: CPU module with its parameters set; ref_fn(x, sd) -> oracle output.  Output, dx and every parameter gradient (relative RMS).
    `rounding='bf16_at_stores'`: the oracle rounds where the HIP path stores a tensor (oracle.set_rounding), so that the bounds measure the
    implementation and not the bf16 representation -- needed where a non-smooth activation sits behind a normalisation (a ReLU mask computed from
    a bf16-stored pre-norm tensor differs from the fp32 one on every element near zero: 3-10 % on the gradients in front of it, measured)."""
    from oracle import genie_oracle as O
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    sd_req = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and k in dict(m.named_parameters()) else v) for k, v in sd.items()}
    xr = x.clone().requires_grad_(True)
    with O.rounding(rounding):
        ref = ref_fn(xr, sd_req)
        dy = bf16_round(torch.randn_like(ref))
        ref.backward(dy)
    m = m.cuda()
    xc = x.cuda().requires_grad_(True)
    out = m(xc)
    assert tuple(out.shape) == tuple(ref.shape), (tag, tuple(out.shape), tuple(ref.shape))
    errs = {'out': rel_rms(out, ref)}
    assert errs['out'] < out_tol, (tag, errs)
    out.backward(dy.cuda())
    errs['dx'] = rel_rms(xc.grad, xr.grad)
    assert errs['dx'] < dx_tol, (tag, errs)
    # a parameter gradient is measured against the larger of its own RMS and 2 % of the block's typical (median) parameter-gradient RMS: a gradient
    # that is analytically zero (a conv bias in front of a channel-summing blur + one-group GroupNorm, which remove it) is rounding noise on both sides
    rms = lambda v: v.detach().float().pow(2).mean().sqrt().item()
    refs = {name: sd_req[name].grad for name, _ in m.named_parameters()}
    assert all(v is not None for v in refs.values()), tag
    floor = 0.02 * sorted(rms(v) for v in refs.values())[len(refs) // 2]
    for name, p in m.named_parameters():
        assert p.grad is not None, (tag, name)
        errs[name] = rms(p.grad.cpu() - refs[name]) / max(rms(refs[name]), floor, 1e-20)
        assert errs[name] < p_tol, (tag, name, errs[name], rms(refs[name]), floor)
    return errs


def draw_resblock(i):
    r = random.Random(13000 + i)
    cin = r.choice([8, 16, 24, 64, 72, 128, 256])
    cout = r.choice([None, None, 8, 16, 64, 128, 136, 256])
    groups = r.choice([g for g in (1, 1, 1, 2, 4, 8) if cin % g == 0 and (cout or cin) % g == 0])
    causal = r.random() < 0.5
    down = r.choice([None, None, None, 2, (1, 2), (2, 2), (2, 1)])
    blur = True if down is not None else r.random() < 0.6      # downsample with use_blur=False raises TypeError in the reference too (video.py:600-615)
    act = r.choice(['swish', 'swish', 'swish', 'leaky', 'relu', 'gelu'])
    n, t = r.choice([1, 2]), r.choice([2, 3, 4, 6])
    h, w = r.choice([(4, 4), (6, 5), (8, 8), (4, 32), (3, 64), (9, 16), (16, 16)])
    while n * t * h * w * max(cin, cout or cin) > 3_000_000:
        t = max(2, t - 1)
        n = 1
        if t == 2:
            h = max(2, h // 2)
    return cin, cout, groups, causal, down, blur, act, (n, t, h, w)


@pytest.mark.parametrize('i', range(40))
def test_video_residual_block_random_geometry(i):
    from oracle import genie_oracle as O
    from genie.module.video import VideoResidualBlock
    cin, cout, groups, causal, down, blur, act, (n, t, h, w) = draw_resblock(i)
    torch.manual_seed(i)
    kw = dict(num_groups=groups, use_causal=causal, downsample=down, use_blur=blur, act_fn=act)
    m = VideoResidualBlock(cin, cout, **kw)
    _randomise(m)
    x = bf16_round(torch.randn(n, cin, t, h, w))
    tag = f'resblock {cin}->{cout} {kw} @{(n, t, h, w)}'
    sd_cpu = {k: v.detach().clone() for k, v in m.state_dict().items()}
    # against the oracle rounding at the HIP path's stores: output 4e-3 (one bf16 ulp on some elements; measured: bit-equal ... 2.5e-3 with GELU), dx 1e-2, parameter gradients 6e-2 (the conv bias in
    # front of the channel-summing blur + GroupNorm is a cancellation residual: 4-5 % measured; every other one < 2 %) ...
    errs = _compare_module(m, lambda xx, sd: O.video_residual_block(xx, sd, '', cin, cout, **kw), x, tag, out_tol=4e-3, dx_tol=1e-2, p_tol=6e-2,
                           rounding='bf16_at_stores')
    # ... and the representation error against the reference's fp32 arithmetic, output and input gradient only
    xr = x.clone().requires_grad_(True)
    ref = O.video_residual_block(xr, sd_cpu, '', cin, cout, **kw)
    mc = m.cuda()
    xc = x.cuda().requires_grad_(True)
    out = mc(xc)
    g = bf16_round(torch.randn_like(ref))
    ref.backward(g)
    out.backward(g.cuda())
    assert rel_rms(out, ref) < 2e-2 and rel_rms(xc.grad, xr.grad) < 5e-2, (tag, rel_rms(out, ref), rel_rms(xc.grad, xr.grad))
    errs['out_fp32'], errs['dx_fp32'] = rel_rms(out, ref), rel_rms(xc.grad, xr.grad)
    report('random_resblock', i=i, cin=cin, cout=cout, size=(n, t, h, w), **kw, out=errs['out'], dx=errs['dx'], out_fp32=errs['out_fp32'], dx_fp32=errs['dx_fp32'],
           worst_param=max((v for k, v in errs.items() if k not in ('out', 'dx', 'out_fp32', 'dx_fp32')), default=0.0))


def draw_resample(i):
    r = random.Random(17000 + i)
    cin, cout = r.choice([3, 8, 16, 64, 72, 128]), r.choice([8, 16, 24, 64, 128])
    tf, sf = r.choice([1, 2]), r.choice([1, 2, 2, 4])
    up = r.random() < 0.5
    n, t = r.choice([1, 2]), r.choice([1, 2, 3, 5])
    h, w = r.choice([(4, 4), (4, 8), (6, 6), (8, 16), (2, 32), (5, 7)])
    if not up:
        t, h, w = max(t, tf + 1), h * sf, w * sf
    return up, cin, cout, tf, sf, (n, t, h, w)


@pytest.mark.parametrize('i', range(32))
def test_spacetime_resample_random_geometry(i):
    """SpaceTimeDownsample (strided CausalConv3d, video.py:457-483) / DepthToSpaceTimeUpsample (conv to C * tf * sf^2 channels + rearrange, video.py:379-430)."""
    from oracle import genie_oracle as O
    from genie.module.video import DepthToSpaceTimeUpsample, SpaceTimeDownsample
    up, cin, cout, tf, sf, (n, t, h, w) = draw_resample(i)
    torch.manual_seed(i)
    cls, ofn = (DepthToSpaceTimeUpsample, O.depth2spacetime_upsample) if up else (SpaceTimeDownsample, O.spacetime_downsample)
    m = cls(in_channels=cin, out_channels=cout, kernel_size=3, time_factor=tf, space_factor=sf)
    _randomise(m)
    x = bf16_round(torch.randn(n, cin, t, h, w))
    tag = f'{"up" if up else "down"} {cin}->{cout} tf={tf} sf={sf} @{(n, t, h, w)}'
    errs = _compare_module(m, lambda xx, sd: ofn(xx, sd, '', time_factor=tf, space_factor=sf), x, tag, out_tol=1e-2, dx_tol=1.5e-2, p_tol=1e-2)
    report('random_resample', i=i, up=up, cin=cin, cout=cout, tf=tf, sf=sf, size=(n, t, h, w), out=errs['out'], dx=errs['dx'])


def draw_lfq(i):
    r = random.Random(21000 + i)
    d = r.choice([1, 2, 3, 5, 8, 8, 10, 12, 18])
    ncb = r.choice([1, 1, 1, 2, 3])
    training = r.random() < 0.6 and d <= 12
    # input_dim = d * ncb: no projection (what the tokenizer builds); None: the reference's default, 2^d (quantization.py:47) -- kept small here
    input_dim = r.choice([d * ncb, d * ncb, d * ncb + r.choice([1, 6, 22]), 64, 512] + ([None] if d <= 8 else []))
    transpose = r.random() < 0.5
    shape = r.choice([(2, 4, 4, 4), (1, 16, 8, 8), (3, 2, 5, 7), (2, 1, 1, 9), (4, 4, 2, 2)])
    return d, ncb, training, input_dim, transpose, shape


@pytest.mark.parametrize('i', range(40))
def test_lfq_module_random_geometry(i):
    """LookupFreeQuantization (quantization.py:32-133) as a module: ids BIT-EXACT and the quantised code exact at the operator boundary (the projected
    latent the kernel saw), training loss and input gradient against the oracle."""
    from oracle import genie_oracle as O
    from genie.module.quantization import LookupFreeQuantization
    d, ncb, training, input_dim, transpose, (b, t, h, w) = draw_lfq(i)
    torch.manual_seed(i)
    m = LookupFreeQuantization(codebook_dim=d, num_codebook=ncb, input_dim=input_dim)
    project = isinstance(m.proj_inp, torch.nn.Linear)
    cin = m.proj_inp.in_features if project else d * ncb
    with torch.no_grad():
        for p in m.parameters():
            p.copy_(bf16_round(torch.randn_like(p) * (0.3 if p.dim() == 1 else 1.0 / p.shape[-1] ** 0.5)))
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    x = bf16_round(torch.randn(b, cin, t, h, w) if transpose else torch.randn(b, t, h, w, cin)) * 0.4
    xr = x.clone().requires_grad_(True)
    sd_req = {k: (v.clone().requires_grad_(True) if k in dict(m.named_parameters()) else v) for k, v in sd.items()}
    (o_ref, i_ref), l_ref = O.lfq_forward(xr, sd_req, '', d, ncb, training=training, transpose=transpose)
    m = m.cuda().train(training)
    xc = x.cuda().requires_grad_(True)
    (o_hip, i_hip), l_hip = m(xc, transpose=transpose)
    assert tuple(o_hip.shape) == tuple(o_ref.shape) and tuple(i_hip.shape) == tuple(i_ref.shape) and i_hip.dtype == i_ref.dtype
    # ids: equal wherever the projected latent is not within rounding of zero (fp32 Linear on both sides, different summation order)
    if not project:
        assert torch.equal(i_hip.cpu(), i_ref), 'LFQ ids differ without a projection in front'
    else:
        z = torch.nn.functional.linear(x.movedim(1, -1) if transpose else x, sd['proj_inp.weight'], sd.get('proj_inp.bias'))
        decided = (z.abs() > 1e-4 * z.abs().max()).reshape(*z.shape[:-1], ncb, d).all(-1)
        decided = decided.reshape(i_ref.shape) if decided.numel() == i_ref.numel() else decided.squeeze()
        assert torch.equal(i_hip.cpu()[decided], i_ref[decided]) and decided.float().mean() > 0.9
    assert rel_rms(o_hip, o_ref) < 1e-2 or (o_hip.float().cpu() - o_ref).abs().max().item() < 2e-2 * o_ref.abs().max().item() + 1e-6
    if training:
        assert l_hip is not None and abs(l_hip.item() - l_ref.item()) < 1e-4 + 2e-3 * abs(l_ref.item()), (l_hip.item(), l_ref.item())
        dy = bf16_round(torch.randn_like(o_ref))
        (l_ref + (o_ref * dy).sum()).backward()
        (l_hip + (o_hip.float() * dy.cuda()).sum()).backward()
        assert rel_rms(xc.grad, xr.grad) < 2e-2, rel_rms(xc.grad, xr.grad)
    else:
        assert l_hip is None and l_ref is None
    report('random_lfq', i=i, d=d, ncb=ncb, training=training, input_dim=input_dim, transpose=transpose, shape=(b, t, h, w))


@pytest.mark.parametrize('cin,cout,tf,sf,size', [(16, 8, 1, 4, (1, 2, 4, 4)), (32, 16, 2, 4, (2, 2, 3, 5)), (64, 24, 1, 4, (1, 1, 6, 6)), (16, 8, 2, 3, (1, 2, 4, 4))])
def test_depth2spacetime_upsample_with_many_sub_pixels(cin, cout, tf, sf, size):
    """The backward-data pass of an upsample conv gathers through the depth-to-space-time rearrange with (taps x sub-pixels) entries: 27 x 16 = 432 at
    space_factor 4, 864 with time_factor 2 -- more than the 256 the generic gather's table holds (`genie_conv_igemm: ntaps 432 out of range`, found by the
    sweep above).  conv_dgrad un-shuffles the gradient and runs the plain 27-tap conv instead whenever the table would overflow."""
    from oracle import genie_oracle as O
    from genie.module.video import DepthToSpaceTimeUpsample
    torch.manual_seed(3)
    m = DepthToSpaceTimeUpsample(in_channels=cin, out_channels=cout, kernel_size=3, time_factor=tf, space_factor=sf)
    _randomise(m)
    n, t, h, w = size
    x = bf16_round(torch.randn(n, cin, t, h, w))
    errs = _compare_module(m, lambda xx, sd: O.depth2spacetime_upsample(xx, sd, '', time_factor=tf, space_factor=sf), x,
                           f'upsample {cin}->{cout} tf={tf} sf={sf}', out_tol=1e-2, dx_tol=1.5e-2, p_tol=1e-2)
    report('upsample_many_sub_pixels', cin=cin, cout=cout, tf=tf, sf=sf, **{k: v for k, v in errs.items() if k in ('out', 'dx')})


def draw_blueprint(i):
    """A random but well-formed (encoder, decoder) blueprint pair in the reference's vocabulary (genie/module/__init__.py:23-93): stem conv, stages of
    `video-residual` (n_rep, widening, groups, causal or not) with `spacetime_downsample` between them, optionally a `space-time_attn` block at the lowest
    resolution, GroupNorm + SiLU + 1x1x1 conv to the code; the decoder mirrors it with `adaptive_group_norm(has_ext)` and `depth2spacetime_upsample`."""
    r = random.Random(31000 + i)
    c = r.choice([16, 32, 64])
    d = r.choice([4, 6, 8, 10])
    enc = [('causal-conv3d', {'in_channels': 3, 'out_channels': c, 'kernel_size': 3})]
    factors, widths = [], [c]
    for _ in range(r.choice([1, 2, 2, 3])):
        c2 = c * 2 if (r.random() < 0.5 and c < 128) else c
        kw = {'in_channels': c, 'out_channels': c2}
        if r.random() < 0.4:
            kw['n_rep'] = 2
            kw.pop('out_channels')
            c2 = c
        if r.random() < 0.3:
            kw['num_groups'] = r.choice([2, 4, 8])
        if r.random() < 0.5:
            kw['use_causal'] = True
        enc.append(('video-residual', kw))
        c = c2
        if r.random() < 0.7:
            tf, sf = r.choice([(1, 2), (2, 2), (2, 1), (1, 2)])
            enc.append(('spacetime_downsample', {'in_channels': c, 'out_channels': c, 'kernel_size': 3, 'time_factor': tf, 'space_factor': sf}))
            factors.append((tf, sf))
        else:
            factors.append(None)
        widths.append(c)
    attn = r.random() < 0.4
    if attn:
        dh = r.choice([16, 32])
        enc.append(('space-time_attn', {'n_head': c // dh, 'd_head': dh, 'transpose': True}))
    enc += [('group_norm', {'num_groups': 8, 'num_channels': c}), ('silu', {}), ('causal-conv3d', {'in_channels': c, 'out_channels': d, 'kernel_size': 1})]
    dec = [('causal-conv3d', {'in_channels': d, 'out_channels': c, 'kernel_size': 3})]
    if attn:
        dec.append(('space-time_attn', {'n_head': c // dh, 'd_head': dh, 'transpose': True}))
    for f, cw in zip(reversed(factors), reversed(widths[:-1])):
        dec.append(('video-residual', {'in_channels': c}))
        if r.random() < 0.6:
            dec.append(('adaptive_group_norm', {'dim_cond': d, 'num_groups': 8, 'num_channels': c, 'has_ext': True}))
        if f is not None:
            dec.append(('depth2spacetime_upsample', {'in_channels': c, 'kernel_size': 3, 'time_factor': f[0], 'space_factor': f[1]}))
        if cw != c:
            dec.append(('video-residual', {'in_channels': c, 'out_channels': cw}))
            c = cw
    dec += [('group_norm', {'num_groups': 8, 'num_channels': c}), ('silu', {}), ('causal-conv3d', {'in_channels': c, 'out_channels': 3, 'kernel_size': 3})]
    ft = 1
    fs = 1
    for f in factors:
        if f is not None:
            ft, fs = ft * f[0], fs * f[1]
    t, hw = ft * r.choice([1, 2, 3]), fs * r.choice([3, 4, 6, 8, 16])
    return tuple(enc), tuple(dec), d, (r.choice([1, 2]), 3, max(t, 2 if ft == 1 else t), hw, hw)


@pytest.mark.parametrize('i', range(32))
def test_tokenizer_random_blueprints(i):
    """`VideoTokenizer` built from drawn blueprints: encoder output, LFQ ids at the operator boundary (bit-exact), decoder output from the same code, and the
    training forward's loss, against the oracle's layer loops (`tokenizer_encode / _decode / _forward_hotpath`): the registry, `parse_blueprint`, `n_rep`,
    `has_ext` routing and every module pairing nobody wrote a test for."""
    from oracle import genie_oracle as O
    from genie import VideoTokenizer
    enc, dec, d, shape = draw_blueprint(i)
    torch.manual_seed(i)
    m = VideoTokenizer(enc, dec, d_codebook=d, gan_loss_weight=0., perc_loss_weight=0.)
    for n, p in m.named_parameters():
        if '.std.' in n or '.avg.' in n:
            torch.nn.init.normal_(p, std=0.3)
    with torch.no_grad():
        for p in m.parameters():
            if p.dim() >= 2:
                p.copy_(bf16_round(p))
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    m = m.cuda()
    x = bf16_round(torch.randn(shape))
    xc = x.cuda()
    enc_ref = O.tokenizer_encode(x, sd, enc)
    enc_hip = m.encode(xc)
    assert tuple(enc_hip.shape) == tuple(enc_ref.shape), (enc, shape)
    e_enc = rel_rms(enc_hip, enc_ref)
    assert e_enc < 3e-2, (e_enc, enc)
    q_hip, idx_hip = m.tokenize(xc)
    (q_o, idx_o), _ = O.lfq_forward(enc_hip.float().cpu(), sd, 'quant.', d, 1, training=False, transpose=True)
    assert torch.equal(idx_hip.cpu(), idx_o) and torch.equal(q_hip.float().cpu(), q_o)
    rec_ref = O.tokenizer_decode(q_o, sd, dec)
    rec_hip = m.decode(q_hip)
    assert tuple(rec_hip.shape) == tuple(rec_ref.shape) == tuple(shape), (dec, shape)
    e_dec = rel_rms(rec_hip, rec_ref)
    assert e_dec < 4e-2, (e_dec, dec)
    m.train()
    loss, _ = m(xc)
    loss.backward()
    assert torch.isfinite(loss) and all(p.grad is not None and torch.isfinite(p.grad).all() for p in m.parameters() if p.requires_grad)
    report('random_blueprint', i=i, layers=(len(enc), len(dec)), d=d, shape=shape, enc=e_enc, dec=e_dec, loss=loss.item())


def draw_dynamics(i):
    r = random.Random(41000 + i)
    n_head, d_head = r.choice([(2, 16), (2, 32), (4, 16), (1, 64), (2, 64), (4, 32), (8, 8)])
    vocab = r.choice([64, 256, 1000, 4096, 1 << 14])
    return (r.choice([1, 2]), n_head, d_head, vocab, r.choice([2, 3, 5, 8]), n_head * d_head,
            (r.choice([2, 3]), r.choice([2, 3, 5, 6]), r.choice([2, 4, 5, 8]), r.choice([2, 3, 4, 10])),
            r.choice([3, 5, 7, 10]), r.choice(['linear', 'cosine', 'arccos']), r.choice([0.7, 1.0, 1.3]))


@pytest.mark.parametrize('i', range(12))
def test_dynamics_model_random_configs(i):
    """`DynamicsModel` on drawn (blocks, heads, vocabulary, action count, token grid, MaskGIT steps / schedule / temperature): logits and the masked
    cross-entropy against the oracle, and `generate` through tests/test_gpu_maskgit.py's checker (sampler bit-exact on identical logits; every id that
    differs from the fp32 oracle end to end explained by the logits' noise)."""
    from oracle import genie_oracle as O
    from test_gpu_maskgit import _build, _check_generate
    n_rep, n_head, d_head, vocab, n_act, dim, shape, steps, which, temp = draw_dynamics(i)
    while int(O.maskgit_schedule(steps, shape[2:], which).min()) < 0:
        # more steps than positions: `clamp(min=1)` over-asks and the reference's last entry goes NEGATIVE (dynamics.py:187-193) -- its behaviour is then
        # the tie order of topk among -inf confidences (an already painted position is painted again): not a contract, not swept
        steps -= 1
    desc = (('space-time_attn', {'n_rep': n_rep, 'n_head': n_head, 'd_head': d_head}),)
    m, sd = _build(desc, vocab, n_act, dim, seed=i)
    torch.manual_seed(300 + i)
    b, t, h, w = shape
    tok, act = torch.randint(0, vocab, shape), torch.randint(0, n_act, (b, t))
    logits, last = m(tok.cuda(), act.cuda())
    ref, _ = O.dynamics_forward(tok, act, sd, desc)
    assert tuple(logits.shape) == (b, t, h, w, vocab) and tuple(last.shape) == (b, h, w, vocab)
    e_logits = rel_rms(logits, ref)
    assert e_logits < 2e-2, (e_logits, desc, vocab, shape)
    mask = torch.rand(shape) < 0.7
    mask[0, 0, 0, 0] = True
    m.train()
    loss = m.compute_loss(tok.cuda(), act.cuda(), mask=mask.cuda())
    loss_ref = O.dynamics_loss(tok, act, mask, sd, desc)
    assert abs(loss.item() - loss_ref.item()) < 2e-2 * abs(loss_ref.item()) + 1e-3, (loss.item(), loss_ref.item())
    loss.backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for n, p in m.named_parameters() if 'freq' not in n and p.requires_grad)
    m.eval()
    u = torch.rand(steps, b * h * w)
    match = _check_generate(m, sd, desc, tok, act, u, steps, which, temp)
    report('random_dynamics', i=i, n_rep=n_rep, heads=(n_head, d_head), vocab=vocab, shape=shape, steps=steps, which=which, temp=temp,
           logits=e_logits, loss=loss.item(), loss_ref=loss_ref.item(), id_match=match)


def draw_lam(i):
    r = random.Random(51000 + i)
    n_head, d_head = r.choice([(2, 16), (2, 32), (4, 16), (1, 64), (4, 32), (2, 64)])
    c = n_head * d_head
    d = r.choice([2, 3, 4, 8])
    st = lambda ext: ('space-time_attn', {'n_head': n_head, 'd_head': d_head, 'transpose': True,
                                          **({'has_ext': True, 'time_attn_kw': {'key_dim': d}} if ext else {})})
    down = r.random() < 0.7
    enc = [st(False)] * r.choice([1, 2])
    dec = [st(True)]
    if down:
        enc = enc + [('spacetime_downsample', {'in_channels': c, 'kernel_size': 3, 'time_factor': 1, 'space_factor': 2}), st(False)]
        dec = dec + [('depth2spacetime_upsample', {'in_channels': c, 'kernel_size': 3, 'time_factor': 1, 'space_factor': 2}), st(r.random() < 0.7)]
    hw = r.choice([(8, 8), (16, 16), (8, 16), (4, 12), (32, 32)])
    return tuple(enc), tuple(dec), d, c, hw, (r.choice([1, 2]), 3, r.choice([2, 3, 5, 8]), *hw)


@pytest.mark.parametrize('i', range(12))
def test_latent_action_random_configs(i):
    """R-lam (`LatentAction` with the SURVEY 8c repairs) on drawn blueprints / widths / frame sizes / codebooks: the training forward against
    `oracle.latent_action_forward` -- action ids where the pre-sign latent is decided, reconstruction and quantiser losses -- and a finite backward."""
    from oracle import genie_oracle as O
    from genie import LatentAction
    enc, dec, d, c, hw, shape = draw_lam(i)
    torch.manual_seed(i)
    m = LatentAction(enc, dec, d_codebook=d, inp_shape=hw, n_embd=c)
    with torch.no_grad():
        for n, p in m.named_parameters():
            if p.dim() >= 2:
                p.copy_(bf16_round(p))
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    m = m.cuda().train()
    x = bf16_round(torch.randn(shape))
    idxs, loss, (rec_loss, q_loss) = m(x.cuda())
    tr = {}
    idx_ref, loss_ref, (rec_ref, q_ref), _ = O.latent_action_forward(x, sd, enc, dec, d, training=True, trace=tr)
    assert tuple(idxs.shape) == tuple(idx_ref.shape)
    assert abs(rec_loss.item() - rec_ref.item()) < 4e-2 * abs(rec_ref.item()), (rec_loss.item(), rec_ref.item(), enc, shape)
    act = tr['act']                                                     # (B, T, d) pre-sign latent of the oracle
    decided = (act.abs() > 0.03 * act.abs().mean()).all(-1).reshape(idx_ref.shape)        # every bit of the id further from 0 than the latent's noise
    assert torch.equal(idxs.cpu()[decided], idx_ref[decided]) and decided.float().mean() >= 0.25, (idxs.cpu(), idx_ref, decided)
    loss.backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for n, p in m.named_parameters() if 'freq' not in n)
    report('random_lam', i=i, width=c, d=d, shape=shape, layers=(len(enc), len(dec)), rec=rec_loss.item(), rec_ref=rec_ref.item(),
           ids_decided=decided.float().mean().item())


def draw_conv_options(i):
    r = random.Random(61000 + i)
    kind = r.choice(['pad', 'pad', 'groups', 'groups', 'transpose'])
    if kind == 'groups':
        g = r.choice([2, 3, 4, 8, 16])
        cin, cout = g * r.choice([1, 2, 4, 8, 16]), g * r.choice([1, 2, 8, 12])
    else:
        g = 1
        cin, cout = r.choice([8, 16, 24, 64, 128]), r.choice([8, 16, 48, 64, 128])
    k = r.choice([3, 3, (3, 3, 3), (1, 3, 3), (3, 1, 1), (2, 3, 3)])
    stride = r.choice([(1, 1, 1), (1, 1, 1), (1, 2, 2), (2, 2, 2), (2, 1, 1)])
    n, t, h, w = r.choice([1, 2]), r.choice([3, 4, 6]), r.choice([4, 6, 8, 9]), r.choice([4, 5, 8, 16])
    mode = r.choice(['replicate', 'reflect', 'circular']) if kind == 'pad' else 'constant'
    return kind, cin, cout, g, k, stride, mode, (n, t, h, w)


@pytest.mark.parametrize('i', range(40))
def test_causal_conv3d_options_random(i):
    """The less-travelled constructor options on drawn shapes: `pad_mode` (reference video.py:154-192: F.pad(..., mode) then an unpadded conv), `groups`
    (video.py:168-175) and `causal-conv3d-transpose` (video.py:202-277) -- outputs, input and parameter gradients against those compositions in fp32."""
    from oracle import genie_oracle as O
    from genie.module import get_module
    from genie.module.video import CausalConv3d
    F = torch.nn.functional
    kind, cin, cout, g, k, stride, mode, (n, t, h, w) = draw_conv_options(i)
    kt, kh, kw = (k, k, k) if isinstance(k, int) else k
    torch.manual_seed(i)
    x = bf16_round(torch.randn(n, cin, t, h, w))
    xr = x.clone().requires_grad_(True)
    if kind == 'transpose':
        m = get_module('causal-conv3d-transpose')(cin, cout, kernel_size=k, stride=stride)
        with torch.no_grad():
            m.weight.copy_(bf16_round(m.weight))
        wt, bt = m.weight.detach().clone().requires_grad_(True), m.bias.detach().clone().requires_grad_(True)
        ref = O.causal_conv_transpose3d(xr, wt, bt, stride, (1, 1, 1), None)
        grads = lambda mm: (mm.weight.grad, mm.bias.grad)
        out_kw = dict(rel=2 ** -6, rms_frac=4e-3)
    else:
        tp = (kt - 1) + (1 - stride[0])
        if tp < 0 and mode != 'constant':
            pytest.skip('a negative causal pad with a non-constant pad_mode is not implemented (DESIGN section 7)')
        if mode in ('reflect', 'circular') and ((kh - 1) // 2 >= h or (kw - 1) // 2 >= w or tp >= t):
            pytest.skip('padding wider than the image: torch refuses it as well')
        m = CausalConv3d(cin, cout, k, stride=stride, pad_mode=mode, groups=g)
        with torch.no_grad():
            m.conv3d.weight.copy_(bf16_round(m.conv3d.weight))
        wt, bt = m.conv3d.weight.detach().clone().requires_grad_(True), m.conv3d.bias.detach().clone().requires_grad_(True)
        pads = ((kw - 1) // 2, (kw - 1) // 2, (kh - 1) // 2, (kh - 1) // 2, tp, 0)
        ref = F.conv3d(F.pad(xr, pads, mode=mode) if mode != 'constant' else F.pad(xr, pads), wt, bt, stride=stride, groups=g)
        grads = lambda mm: (mm.conv3d.weight.grad, mm.conv3d.bias.grad)
        out_kw = {}
    if ref.numel() == 0:
        pytest.skip('empty output')
    dy = bf16_round(torch.randn_like(ref))
    ref.backward(dy)
    m = m.cuda()
    xc = x.cuda().requires_grad_(True)
    out = m(xc)
    tag = f'{kind} {cin}->{cout} g={g} k={k} s={stride} {mode} @{(n, t, h, w)}'
    assert tuple(out.shape) == tuple(ref.shape), tag
    assert_close_bf16(out, ref, tag, **out_kw)
    out.backward(dy.cuda())
    gw, gb = grads(m)
    errs = {'dx': rel_rms(xc.grad, xr.grad), 'dw': rel_rms(gw, wt.grad), 'db': rel_rms(gb, bt.grad)}
    assert errs['dx'] < 8e-3 and errs['dw'] < 3e-3 and errs['db'] < 3e-3, (tag, errs)
    report('random_conv_options', i=i, kind=kind, cin=cin, cout=cout, groups=g, kernel=k, stride=stride, mode=mode, size=(n, t, h, w), **errs)


def draw_image_block(i):
    r = random.Random(71000 + i)
    cin = r.choice([8, 16, 24, 32, 64, 128])
    cout = r.choice([None, cin, 16, 32, 64, 128])
    down = r.choice([None, None, 2]) if cout is not None else None
    groups = r.choice([g for g in (1, 2, 4, 8) if cin % g == 0 and (cout or cin) % g == 0])
    n, h, w = r.choice([1, 2, 3]), r.choice([4, 6, 8, 12, 16]), r.choice([4, 6, 8, 10, 32])
    return dict(inp_channel=cin, out_channel=cout, num_groups=groups, **({'downsample': down} if down else {})), (n, cin, h, w)


@pytest.mark.parametrize('i', range(24))
def test_image_residual_block_random(i):
    """`image-residual` (reference image.py:105-163; the FrameDiscriminator's block) on drawn widths / groups / sizes, with and without the
    pixel-unshuffle downsample, against the oracle."""
    from genie.module.image import ImageResidualBlock
    from oracle import genie_oracle as O
    kw, size = draw_image_block(i)
    torch.manual_seed(i)
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
    e_out = rel_rms(out, ref)
    assert e_out < 1e-2, (e_out, kw, size)
    out.backward(dy.cuda())
    e_dx = rel_rms(xc.grad, xr.grad)
    assert e_dx < 3e-2, (e_dx, kw, size)
    rms = lambda v: v.detach().float().pow(2).mean().sqrt().item()
    floor = 0.02 * sorted(rms(sd[n].grad) for n, _ in m.named_parameters())[len(list(m.parameters())) // 2]
    for n, p in m.named_parameters():
        e = rms(p.grad.cpu() - sd[n].grad) / max(rms(sd[n].grad), floor)
        assert e < (0.2 if p.dim() == 1 else 6e-2), (n, e, kw, size)     # per-channel sums of bf16 gradients over few pixels (the bound of test_gpu_gan.py)
    report('random_image_block', i=i, size=size, out=e_out, dx=e_dx, **{k: v for k, v in kw.items()})
