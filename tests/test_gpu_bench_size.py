"""The launches `bench.py` times, compared with the ORACLE at the bench's own batch (64 clips) -- VERDICT r5 item 1c.

The kernel-level files check every kernel against the oracle at sizes the oracle walks in a second, and `test_gpu_properties.py` checks
size-independent properties at 32 clips; neither compares a 64-clip launch with the oracle.  Here the launch has the bench's geometry
(64 clips, the layer's channels and resolution) and the oracle is applied where the operator's own structure makes a small comparison EXACT:

  * forward / backward-data / GroupNorm apply: clips are independent (no operator on the path mixes clips: SURVEY 8e), so clip k of the
    64-clip output is the oracle's output on clip k alone.  Compared on the first, a middle and the LAST clip (the tail of every index
    range: 2^28..2^30 elements in, the last row tile, the last workgroup of every XCD).
  * weight / bias / affine-parameter gradients sum over the 64 clips.  The inputs are built so that the sum is known from FOUR oracle clips:
    x_n = X[n mod 4], dy_n = c_n * DY[n mod 4] with c_n a signed power of two (exact in bf16), hence
    dW = sum_k (sum_{n = k mod 4} c_n) * dW_k -- every one of the 64 clips carries non-zero data, and a slice that is dropped, doubled or read from
    the wrong clip changes the result.

Tolerances are the operator tests' own (one bf16 rounding of the fp32 oracle for activations, 1e-3 of the tensor maximum for fp32
parameter gradients accumulated with split-K atomics)."""
import pytest
import torch
import torch.nn.functional as F

from util import assert_close_bf16, bf16_round, report

pytestmark = pytest.mark.gpu

N = 64
CHECK = (0, 37, N - 1)


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


def _coeffs():
    """c_n for n = 0..63: signed powers of two, different in every group of four, summing to non-trivial values per residue class."""
    g = torch.Generator().manual_seed(5)
    e = torch.randint(-2, 2, (N,), generator=g).float()
    s = torch.randint(0, 2, (N,), generator=g).float() * 2 - 1
    return s * torch.pow(2.0, e)


def _periodic(base: torch.Tensor, c=None) -> torch.Tensor:
    """(4, C, T, H, W) fp32 base clips -> CL bf16 (64, C, T, H, W) on the GPU with clip n = c_n * base[n mod 4]."""
    from genie.cl import to_cl
    b = to_cl(base.cuda())                                            # bf16 CL
    x = b.repeat(N // 4, 1, 1, 1, 1)
    x = to_cl(x)
    if c is not None:
        x = to_cl((x.float() * c.cuda().view(N, 1, 1, 1, 1)))         # powers of two: exact
    return x


def _run_conv(G, cin, cout, causal, thw, seed, resid=False):
    """One 3x3x3 conv layer at 64 clips through the module-level path (functional.conv3d -> _Conv3dFn: forward, backward-data, weight
    gradient) with the periodic inputs; returns everything the checks need."""
    from oracle import genie_oracle as O
    torch.manual_seed(seed)
    t, h, w = thw
    kernel = (3, 3, 3)
    X = bf16_round(torch.randn(4, cin, t, h, w))
    DY = bf16_round(torch.randn(4, cout, t, h, w))
    wt = bf16_round(torch.randn(cout, cin, *kernel) / (cin * 27) ** 0.5)
    b = torch.randn(cout)
    c = _coeffs()
    spec = G.conv.causal_spec(cin, cout, kernel) if causal else G.conv.same_spec(cin, cout, kernel)
    x = _periodic(X).requires_grad_(True)
    dy = _periodic(DY, c)
    op = G.GF.ConvOp(spec)
    wd, bd = wt.cuda().requires_grad_(True), b.cuda().requires_grad_(True)
    G.conv.PROFILER = prof = G.conv.LaunchProfiler()
    try:
        out = G.GF.conv3d(x, wd, bd, op)
        out.backward(dy)
        torch.cuda.synchronize()
    finally:
        G.conv.PROFILER = None
    names = sorted(prof.summary())
    # oracle on the four base clips
    Xr = X.clone().requires_grad_(True)
    wr, br = wt.clone().requires_grad_(True), b.clone().requires_grad_(True)
    ref = O.causal_conv3d(Xr, wr, br) if causal else O.conv3d_same(Xr, wr, br)
    csum = torch.stack([c[k::4].sum() for k in range(4)])
    # d/dW of sum_n <dy_n, conv(x_n)> = sum_k csum_k <DY_k, conv(X_k)>
    (ref * (DY * csum.view(4, 1, 1, 1, 1))).sum().backward()
    dw_ref, db_ref = wr.grad.clone(), br.grad.clone()
    # backward-data per clip: dx_n = c_n * dgrad(DY[n mod 4])
    Xr.grad = None
    ref2 = O.causal_conv3d(Xr, wt, b) if causal else O.conv3d_same(Xr, wt, b)
    ref2.backward(DY)
    return dict(out=out, dx=x.grad, dw=wd.grad, db=bd.grad, ref=ref.detach(), dx_ref=Xr.grad, dw_ref=dw_ref, db_ref=db_ref, c=c, names=names)


def _check_conv(tag, r, expect_kernels):
    for k in expect_kernels:
        assert any(k in n for n in r['names']), (k, r['names'])
    worst = {}
    for n in CHECK:
        assert_close_bf16(r['out'][n], r['ref'][n % 4], f'{tag} forward clip {n}')
        assert_close_bf16(r['dx'][n], r['dx_ref'][n % 4] * r['c'][n], f'{tag} backward-data clip {n}')
    sw, sb = r['dw_ref'].abs().max().item(), r['db_ref'].abs().max().item()
    ew = (r['dw'].float().cpu() - r['dw_ref']).abs().max().item() / sw
    eb = (r['db'].float().cpu() - r['db_ref']).abs().max().item() / sb
    report(f'bench_size_{tag}', clips=N, kernels=r['names'], dw_max_err_of_max=ew, db_max_err_of_max=eb)
    # fp32 accumulation of bf16 products over 64 x T x H x W pixels, split-K partial sums in atomics: 1e-3 of the tensor's largest element
    assert ew < 1e-3, (tag, ew)
    assert eb < 1e-3, (tag, eb)


def test_bench_size_conv_256_at_16x32x32(G):
    """256 -> 256 @ 16x32x32, 64 clips: forward and backward-data on the 256 x 256 tile (`igemm3w_kernel`, the bench line's `roofline` kernel),
    weight gradient on `wgrad3l_kernel`."""
    r = _run_conv(G, 256, 256, False, (16, 32, 32), 41)
    _check_conv('conv256_16x32x32', r, ('igemm3_kernel<256x256>', 'wgrad3l_kernel'))


def test_bench_size_conv_128_at_16x64x64(G):
    """128 -> 128 @ 16x64x64, 64 clips (0.5 G elements per tensor): forward and backward-data on the 256 x 128 tile with 32-channel K tiles
    (`igemm3h_kernel`), weight gradient on `wgrad3l_kernel`."""
    r = _run_conv(G, 128, 128, False, (16, 64, 64), 43)
    _check_conv('conv128_16x64x64', r, ('igemm3_kernel<256,k32>', 'wgrad3l_kernel'))


def test_bench_size_stem_conv(G):
    """Stem CausalConv3d(3 -> 128) @ 16x64x64, 64 clips: `conv_narrow_in_kernel` forward, `conv_narrow_wgrad` straight into dW / db."""
    r = _run_conv(G, 3, 128, True, (16, 64, 64), 47)
    _check_conv('stem_3_128', r, ('conv_narrow_in_kernel', 'conv_narrow_wgrad'))


def test_bench_size_head_conv(G):
    """Head CausalConv3d(128 -> 3) @ 16x64x64, 64 clips: `conv_narrow_out` forward, backward-data = the narrow-in kernel with flipped taps,
    weight gradient accumulated straight into dW / db."""
    r = _run_conv(G, 128, 3, True, (16, 64, 64), 53)
    _check_conv('head_128_3', r, ('conv_narrow_out', 'conv_narrow_wgrad'))


def test_bench_size_groupnorm_silu(G):
    """GroupNorm(1 group, 128 channels) + SiLU on 64 clips of 16x64x64 (the first norm of the full-resolution residual blocks): forward and input
    gradient per clip, dgamma / dbeta through the periodic construction."""
    from oracle import genie_oracle as O
    torch.manual_seed(59)
    C, thw = 128, (16, 64, 64)
    X = bf16_round(torch.randn(4, C, *thw) * 1.5 + 0.3)
    DY = bf16_round(torch.randn(4, C, *thw))
    gamma, beta = torch.randn(C) * 0.3 + 1, torch.randn(C) * 0.2
    c = _coeffs()
    x = _periodic(X).requires_grad_(True)
    dy = _periodic(DY, c)
    gd, bd = gamma.cuda().requires_grad_(True), beta.cuda().requires_grad_(True)
    y = G.GF.group_norm(x, 1, gd, bd, 1e-5, act=True)
    y.backward(dy)
    torch.cuda.synchronize()
    Xr, gr, br = X.clone().requires_grad_(True), gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    ref = O.silu(O.group_norm(Xr, 1, gr, br))
    csum = torch.stack([c[k::4].sum() for k in range(4)])
    (ref * (DY * csum.view(4, 1, 1, 1, 1))).sum().backward()
    dg_ref, db_ref = gr.grad.clone(), br.grad.clone()
    Xr.grad = None
    O.silu(O.group_norm(Xr, 1, gamma, beta)).backward(DY)
    for n in CHECK:
        assert_close_bf16(y[n], ref[n % 4].detach(), f'GroupNorm+SiLU forward clip {n}')
        assert_close_bf16(x.grad[n], Xr.grad[n % 4] * c[n], f'GroupNorm+SiLU input gradient clip {n}', rel=2 ** -6, rms_frac=4e-3)
    eg = (gd.grad.cpu() - dg_ref).abs().max().item() / dg_ref.abs().max().item()
    eb = (bd.grad.cpu() - db_ref).abs().max().item() / db_ref.abs().max().item()
    report('bench_size_groupnorm_silu', clips=N, dgamma_max_err_of_max=eg, dbeta_max_err_of_max=eb)
    assert eg < 2e-3 and eb < 2e-3, (eg, eb)
