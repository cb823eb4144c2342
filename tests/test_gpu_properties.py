"""Size-independent properties of the hot path AT THE BENCHMARKED SIZES (-m gpu).

The oracle finishes a 2-clip MAGVIT2 step in seconds (tests/test_gpu_tokenizer.py) but not the 32 / 64-clip launches bench.py times,
whose kernels take other code paths (256 x 256 tiles instead of K splits, every row tile of a layer in flight, > 2^31-byte tensors).
What the domain offers without a reference result: linearity of the convolution in its input, invariances of GroupNorm, the
lookup-free quantiser's fixed points and scale invariance, determinism and the row-wise nature of the vocabulary head, the MaskGIT
paint step's bookkeeping.  Each check runs the SAME entry points bench.py drives, on the bench's tensor shapes.
"""
import pytest
import torch

from util import bf16_round, report

pytestmark = pytest.mark.gpu

CLIPS = 32          # bench.py's per-GPU batch is 64; 32 already takes every large-launch path and halves the memory of the checks


@pytest.fixture(scope='module')
def G():
    from genie import _hip, cl, conv
    from genie import functional as GF
    _hip.load_library()

    class NS:
        pass
    ns = NS()
    ns.hip, ns.cl, ns.conv, ns.GF = _hip, cl, conv, GF
    return ns


def _rel(a, b):
    return ((a.float() - b.float()).norm() / (b.float().norm() + 1e-30)).item()


@pytest.mark.parametrize('cin,cout,size', [(128, 128, (16, 64, 64)), (256, 256, (16, 32, 32))])
def test_conv_is_linear_at_bench_size(G, cin, cout, size):
    """conv(a x + b y) = a conv(x) + b conv(y) for the residual 3x3x3 layers at 32 clips (the kw-triple 256-row / 256 x 256 kernels,
    8192 / 4096 row tiles): forward, backward-data and the weight gradient (linear in dy).  bf16 outputs: compared in norm."""
    torch.manual_seed(5)
    t, h, w = size
    spec = G.conv.same_spec(cin, cout, (3, 3, 3))
    wt = (torch.randn(cout, cin, 3, 3, 3, device='cuda') * (cin * 27) ** -0.5).contiguous(memory_format=torch.channels_last_3d)
    wf, wb = G.conv.pack_weight_fwd(wt, spec), G.conv.pack_weight_bwd(wt, spec)
    x = G.cl.to_cl(torch.randn(CLIPS, cin, t, h, w, device='cuda'))
    y = G.cl.to_cl(torch.randn(CLIPS, cin, t, h, w, device='cuda'))
    a, b = 0.5, -2.0                                    # powers of two: a x + b y is exact in bf16 up to the final rounding of the sum
    z = G.cl.to_cl((a * x.float() + b * y.float()))
    fx, fy, fz = (G.conv.conv_forward(v, wf, None, spec) for v in (x, y, z))
    comb = a * fx.float() + b * fy.float()
    e_fwd = _rel(fz, comb)
    del fx, fy
    dx, dy_, dz = (G.conv.conv_dgrad(v, wb, spec, size) for v in (x, y, z))      # any CL tensor of the output's shape is a gradient
    e_dg = _rel(dz, a * dx.float() + b * dy_.float())
    del dx, dy_, dz
    g = []
    for v in (x, y, z):
        dw = torch.zeros_like(wt)
        G.conv.conv_wgrad(fz, v, spec, dw, None)       # dW is linear in the output gradient; fz plays the input
        g.append(dw)
    # swap roles: conv_wgrad(x=input, dy=gradient): here input = fz (cout channels == cin for these layers), gradient = v
    e_wg = _rel(g[2], a * g[0] + b * g[1])
    report('conv_linearity', layer=f'{cin}->{cout}@{size}', clips=CLIPS, fwd=e_fwd, dgrad=e_dg, wgrad=e_wg)
    assert e_fwd < 6e-3 and e_dg < 6e-3, (e_fwd, e_dg)   # three bf16 roundings (z, and the two terms of the combination)
    assert e_wg < 6e-3, e_wg


def test_groupnorm_invariances_at_bench_size(G):
    """GroupNorm(alpha x + beta) = GroupNorm(x) for alpha > 0 (one group, 32 x 128 x 16 x 64 x 64: 8.4 M elements per sample, 128 blocks
    per sample), its output has zero mean / unit variance per sample, and the backward of a per-sample constant gradient is zero."""
    torch.manual_seed(6)
    GF = G.GF
    x = G.cl.to_cl(torch.randn(CLIPS, 128, 16, 64, 64, device='cuda'))
    gamma = torch.ones(128, device='cuda', requires_grad=True)
    beta = torch.zeros(128, device='cuda', requires_grad=True)
    with torch.no_grad():
        y = GF.group_norm(x, 1, gamma, beta, 1e-5, act=False)
    x2 = G.cl.to_cl(4.0 * x.float() + 0.0)              # exact in bf16
    with torch.no_grad():
        y2 = GF.group_norm(x2, 1, gamma, beta, 1e-5, act=False)
    e_scale = _rel(y2, y)
    yf = y.float()
    m = yf.mean(dim=(1, 2, 3, 4))
    v = yf.var(dim=(1, 2, 3, 4), unbiased=False)
    xr = x.detach().requires_grad_(True)
    yy = GF.group_norm(xr, 1, gamma, beta, 1e-5, act=False)
    (gx,) = torch.autograd.grad(yy, [xr], [G.cl.to_cl(torch.ones_like(yf))])
    gmax = gx.float().abs().max().item()
    report('groupnorm_invariances', clips=CLIPS, scale_rel=e_scale, mean_max=m.abs().max().item(), var_dev=(v - 1).abs().max().item(), const_grad_max=gmax)
    assert e_scale < 1e-3
    assert m.abs().max().item() < 2e-3 and (v - 1).abs().max().item() < 5e-3
    assert gmax < 2e-3                                   # d/dx of sum(y) = 0: the normalisation removes the mean


def test_lfq_fixed_points_and_scale_invariance_at_bench_size(G):
    """Lookup-free quantisation on the bench's latent grid (32 x 18 x 4 x 8 x 8 -> 8192 tokens of 18 bits): indices depend on signs
    only (invariant under any positive scale), quantising the quantised latent returns it (idempotence), and the indices reproduce
    the sign pattern MSB first (reference quantization.py:77-108) -- exact integer comparisons."""
    torch.manual_seed(7)
    GF = G.GF
    z = torch.randn(CLIPS * 4 * 8 * 8, 18, device='cuda').to(torch.bfloat16)
    with torch.no_grad():
        q1, i1, _ = GF.lfq_rows(z, 1, 18, False, 100., 0.25, 0.1, 1.)
        q2, i2, _ = GF.lfq_rows((z.float() * 37.5).to(torch.bfloat16), 1, 18, False, 100., 0.25, 0.1, 1.)
        q3, i3, _ = GF.lfq_rows(q1.reshape(z.shape).to(torch.bfloat16).contiguous(), 1, 18, False, 100., 0.25, 0.1, 1.)
    bits = (z.float() > 0).long()
    want = (bits << torch.arange(17, -1, -1, device='cuda')).sum(-1)
    assert torch.equal(i1.reshape(-1), want)
    assert torch.equal(i1, i2) and torch.equal(i1, i3)
    assert torch.equal(q3.reshape(z.shape).float(), q1.reshape(z.shape).float())
    assert set(q1.float().unique().tolist()) <= {-1.0, 1.0, 0.0}


def test_vocabulary_head_is_row_wise_and_deterministic(G):
    """The Dynamics head at full size (3072 gathered rows x 512 -> 2^18 logits, the 256 x 256 GEMM with the interleaved 16-byte-store
    epilogue): identical rows give identical logits wherever they sit, a permutation of the rows permutes the logits, and two runs are
    bit-identical.  This is what lets compute_loss run the head on the masked rows only."""
    torch.manual_seed(8)
    GF, conv = G.GF, G.conv
    R, D, V = 3072, 512, 1 << 18
    op = GF.ConvOp(conv.ConvSpec(D, V, (1, 1, 1)))
    w = (torch.randn(V, D, device='cuda') * D ** -0.5)
    bias = torch.randn(V, device='cuda') * 0.1
    x = bf16_round(torch.randn(R, D)).cuda()
    x[7] = x[1999]                                       # a repeated row
    perm = torch.randperm(R, device='cuda')

    def head(rows):
        xc = G.cl.to_cl(rows.t().reshape(1, D, 1, R // 256, 256))
        with torch.no_grad():
            y = GF.conv3d(xc, w[:, :, None, None, None], bias, op)
        return y.permute(0, 2, 3, 4, 1).reshape(R, V)

    a = head(x)
    b = head(x)
    assert torch.equal(a, b)
    assert torch.equal(a[7], a[1999])
    c = head(x[perm])
    assert torch.equal(c, a[perm])


def test_maskgit_paint_bookkeeping_at_bench_size(G):
    """genie_maskgit_paint on 4 x 64 positions drawn from 2^18 codes, over a whole schedule: every step unmasks exactly k
    positions per sample, only masked positions change, painted codes equal the sampled tokens, and the positions
    taken are the most confident ones (ties -> lower index), as in reference dynamics.py:146-158."""
    torch.manual_seed(9)
    GF = G.GF
    b, n = 4, 64
    code = torch.zeros(b, n, dtype=torch.int64, device='cuda')
    mask = torch.ones(b, n, dtype=torch.uint8, device='cuda')
    left = n
    for step, k in enumerate([7, 9, 16, 20, 12]):           # sums to n, as get_schedule's counts do (dynamics.py:167-195)
        conf = torch.rand(b, n, device='cuda')
        conf[:, 3] = conf[:, 5]                          # a tie inside every sample
        pred = torch.randint(0, 1 << 18, (b, n), device='cuda')
        code0, mask0 = code.clone(), mask.clone()
        GF.maskgit_paint(conf, pred, k, code, mask)
        took = (mask0 == 1) & (mask == 0)
        kk = min(k, left)
        assert (took.sum(1) == kk).all(), (step, took.sum(1).tolist(), kk)
        assert torch.equal(code[~took], code0[~took]) and torch.equal(code[took], pred[took])
        assert ((mask0 == 0) <= (mask == 0)).all()      # nothing is re-masked
        # the taken set = top-kk of the masked confidences, ties to the lower index
        cm = torch.where(mask0 == 1, conf, torch.full_like(conf, -1.0))
        order = torch.sort(cm.double() - torch.arange(n, device='cuda').double() * 1e-12, dim=1, descending=True).indices[:, :kk]
        exp = torch.zeros_like(took)
        exp.scatter_(1, order, True)
        assert torch.equal(took, exp)
        left -= kk
    assert left == 0 and int(mask.sum()) == 0


def test_gradient_determinism(G):
    """Same step, same state, twice.  Default mode: split-K partial sums of the weight-gradient kernels meet in fp32
    atomics (conv_wgrad3.hip / conv_wgrad.hip / conv_narrow.hip), so a gradient is reproducible only up to the order of those adds --
    bounded here at 1e-5 relative RMS per parameter (measured ~1e-7; reported).  GENIE_DETERMINISTIC / conv.set_deterministic(True):
    one owner per output tile, fixed order, and the bias gradients summed in a fixed tree instead of the kernels' epilogue atomics -- the conv
    weight and bias gradients are bit-identical run to run.  Forward values, input gradients,
    GroupNorm statistics and LFQ indices are bit-identical in both modes (no atomics on those paths)."""
    from genie import VideoTokenizer
    from genie.trainer import ParamArena
    enc = (('causal-conv3d', {'in_channels': 3, 'out_channels': 128, 'kernel_size': 3}),
           ('video-residual', {'n_rep': 2, 'in_channels': 128}),
           ('spacetime_downsample', {'in_channels': 128, 'out_channels': 128, 'kernel_size': 3, 'time_factor': 2, 'space_factor': 2}),
           ('video-residual', {'in_channels': 128, 'out_channels': 256}),
           ('group_norm', {'num_groups': 8, 'num_channels': 256}), ('silu', {}),
           ('causal-conv3d', {'in_channels': 256, 'out_channels': 10, 'kernel_size': 1}))
    dec = (('causal-conv3d', {'in_channels': 10, 'out_channels': 256, 'kernel_size': 3}),
           ('video-residual', {'in_channels': 256}),
           ('adaptive_group_norm', {'dim_cond': 10, 'num_groups': 8, 'num_channels': 256, 'has_ext': True}),
           ('depth2spacetime_upsample', {'in_channels': 256, 'kernel_size': 3, 'time_factor': 2, 'space_factor': 2}),
           ('video-residual', {'in_channels': 256, 'out_channels': 128}),
           ('group_norm', {'num_groups': 8, 'num_channels': 128}), ('silu', {}),
           ('causal-conv3d', {'in_channels': 128, 'out_channels': 3, 'kernel_size': 3}))
    torch.manual_seed(0)
    m = VideoTokenizer(enc, dec, d_codebook=10, gan_loss_weight=0., perc_loss_weight=0.).cuda().train()
    arena = ParamArena(m)
    arena.attach_weight_packs(m)
    x = bf16_round(torch.randn(4, 3, 8, 64, 64)).cuda()

    def step():
        arena.zero_grad()
        loss, _ = m(x)
        loss.backward()
        torch.cuda.synchronize()
        return loss.item(), {n: p.grad.detach().clone() for n, p in m.named_parameters()}

    out = {}
    for det in (False, True):
        old = G.conv.set_deterministic(det)
        try:
            l1, g1 = step()
            l2, g2 = step()
            _, g3 = step()
        finally:
            G.conv.set_deterministic(old)
        assert l1 == l2, (det, l1, l2)                          # the forward pass has no atomics: the loss is bit-identical
        spread = {n: ((g1[n] - g2[n]).float().norm() / (g1[n].float().norm() + 1e-30)).item() for n in g1}
        worst = max(spread, key=spread.get)
        top = sorted(spread.items(), key=lambda kv: -kv[1])[:5]
        out['deterministic' if det else 'default'] = (spread[worst], worst, sum(1 for v in spread.values() if v == 0), len(spread), top)
        assert spread[worst] < 1e-5, (det, top)
        if det:
            conv_w = [n for n in g1 if n.endswith('weight') and g1[n].dim() == 5]
            conv_p = conv_w + [n[:-6] + 'bias' for n in conv_w if n[:-6] + 'bias' in g1]
            moving = [n for n in conv_p if not (torch.equal(g1[n], g2[n]) and torch.equal(g1[n], g3[n]))]
            assert not moving, moving                           # every conv weight AND bias gradient bit-identical (bias: fixed-order sum)
    report('gradient_determinism', default_worst_rel=out['default'][0], default_worst_param=out['default'][1], default_bit_identical=out['default'][2],
           deterministic_worst_rel=out['deterministic'][0], deterministic_bit_identical=out['deterministic'][2], params=out['default'][3],
           default_top5=out['default'][4], deterministic_top5=out['deterministic'][4])
