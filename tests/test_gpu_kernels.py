"""Operator-level parity of the HIP kernels against the CPU oracle (run on the MI355X: -m gpu).

Every comparison feeds the oracle the SAME bf16-representable inputs/weights the kernel sees, computes
in fp32 on the CPU and bounds the difference by one bf16 rounding of the result (tests/util.py)."""
import json
import os

import pytest
import torch
import torch.nn.functional as F

from util import ROOT, assert_close_bf16, bf16_round

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def G():
    from genie import _hip, cl, conv
    _hip.load_library()
    class NS: pass
    ns = NS(); ns.hip, ns.cl, ns.conv = _hip, cl, conv
    return ns


def test_probe_ds_read_tr16(G):
    """Pin the lane semantics of ds_read_b64_tr_b16 the wgrad kernel relies on: within each 16-lane group the
    lanes' 8-byte rows form a 16x4 matrix M[i][j] (i = lane in group); lane i receives ... (recorded)."""
    img = torch.arange(2048, dtype=torch.int16, device='cuda')
    res = {}
    for name, addr in {
        'linear8': [l * 8 for l in range(64)],
        'rows64B': [(l >> 2) * 64 + (l & 3) * 8 for l in range(64)],
    }.items():
        a = torch.tensor(addr, dtype=torch.int32, device='cuda')
        out = torch.zeros(64 * 4, dtype=torch.int16, device='cuda')
        G.hip.check(G.hip.load_library().genie_probe_ds_read_tr16(img.data_ptr(), a.data_ptr(), out.data_ptr(), G.hip.stream_ptr()), 'probe')
        torch.cuda.synchronize()
        res[name] = out.cpu().reshape(64, 4).tolist()
    os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
    with open(os.path.join(ROOT, 'gpurun_out', 'probe_tr16.json'), 'w') as f:
        json.dump(res, f)
    # semantics assumed by conv_wgrad.hip: lane (16g + i) supplies row i of its group's 16x4 (u16) matrix and
    # receives column-of-blocks: element j of lane (16g + i) = M[4*(i//4)... ] -- see the kernel header; here:
    lin = res['linear8']
    # with addr = 8*lane, u16 index = 4*lane + j before the transpose; group g holds u16 [64g, 64g+64)
    # viewed as a 4 x 16 row-major block B[r][c] = 64g + 16r + c; the transposed read gives lane i column i
    for l in range(64):
        g, i = l // 16, l % 16
        assert lin[l] == [64 * g + 16 * r + i for r in range(4)], (l, lin[l])


def test_layout_roundtrip(G):
    torch.manual_seed(0)
    for shape in [(2, 3, 4, 8, 8), (1, 18, 2, 4, 4), (2, 32, 3, 5, 7), (1, 128, 2, 8, 8)]:
        x = torch.randn(shape, device='cuda')
        y = G.cl.to_cl(x)
        assert G.cl.is_cl(y) and y.shape == x.shape
        assert torch.equal(y.float(), x.to(torch.bfloat16).float())
        cp = G.cl.pitch_of(y)
        if cp != shape[1]:   # pad channels are zero
            base = torch.as_strided(y, (shape[0], shape[2], shape[3], shape[4], cp), (y.stride(0), y.stride(2), y.stride(3), y.stride(4), 1))
            assert (base[..., shape[1]:] == 0).all()
        back = G.cl.from_cl(y)
        assert torch.equal(back, x.to(torch.bfloat16).float())
        # non-contiguous source
        xt = x.permute(0, 1, 2, 4, 3)
        assert torch.equal(G.cl.to_cl(xt).float(), xt.to(torch.bfloat16).float())


@pytest.mark.parametrize('d,ncb', [(18, 1), (8, 1), (6, 3), (10, 1)])
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_lfq_quantize_bit_exact(G, d, ncb, dtype):
    from oracle import genie_oracle as O
    torch.manual_seed(1)
    ntok = 1000
    z = torch.randn(ntok, ncb * d) * 0.5
    z[3, :] = 0.
    z[5, 0] = -0.
    z = z.to(dtype)
    pitch = (ncb * d + 7) & ~7
    zz = torch.zeros(ntok, pitch, dtype=dtype, device='cuda')
    zz[:, :ncb * d] = z.cuda()
    quant = torch.full_like(zz, 7.)
    idx = torch.zeros(ntok, ncb, dtype=torch.int64, device='cuda')
    lib = G.hip.load_library()
    G.hip.check(lib.genie_lfq_quantize(zz.data_ptr(), G.hip.GENIE_F32 if dtype == torch.float32 else G.hip.GENIE_BF16, ntok, ncb, d,
                                       pitch, quant.data_ptr(), idx.data_ptr(), G.hip.stream_ptr()), 'lfq')
    ref_idx = O.lfq_indices_numpy(z.float().reshape(ntok, ncb, d).numpy())
    assert (idx.cpu().numpy() == ref_idx).all()
    assert torch.equal(quant[:, :ncb * d].float().cpu(), torch.sign(z.float()))


def _gn_ref(x, G_, gamma, beta, ada_s, ada_b, act):
    from oracle import genie_oracle as O
    y = O.group_norm(x, G_, gamma, beta)
    if ada_s is not None:
        shape = ada_s.shape + (1, 1, 1)
        y = y * ada_s.reshape(shape) + ada_b.reshape(shape)
    return O.silu(y) if act else y


@pytest.mark.parametrize('N,C,G_,thw,ada,act', [
    (2, 128, 1, (4, 16, 16), False, True), (2, 512, 8, (2, 8, 8), True, False), (1, 24, 3, (3, 5, 7), False, True),
    (3, 64, 64, (2, 4, 4), True, True), (2, 16, 8, (1, 4, 4), False, False),
    (2, 512, 1, (4, 8, 8), True, True),          # the low-resolution 512-channel layers, adaptive scale / shift
    (1, 64, 1, (8, 64, 64), False, True),        # G = 1, 4 MiB sample: several blocks per sample
])
def test_groupnorm_fwd_bwd(G, N, C, G_, thw, ada, act):
    torch.manual_seed(2)
    x = bf16_round(torch.randn(N, C, *thw) * 1.5 + 0.3)
    dy = bf16_round(torch.randn(N, C, *thw))
    gamma, beta = torch.randn(C), torch.randn(C)
    ada_s = torch.randn(N, C) if ada else None
    ada_b = torch.randn(N, C) if ada else None
    leaves = [t.clone().requires_grad_(True) for t in (x, gamma, beta)] + ([ada_s.clone().requires_grad_(True), ada_b.clone().requires_grad_(True)] if ada else [])
    ref = _gn_ref(leaves[0], G_, leaves[1], leaves[2], leaves[3] if ada else None, leaves[4] if ada else None, act)
    ref.backward(dy)

    lib = G.hip.load_library()
    xc, dyc = G.cl.to_cl(x.cuda()), G.cl.to_cl(dy.cuda())
    y, dx = G.cl.empty_like_cl(xc), G.cl.empty_like_cl(xc)
    cp, npix = G.cl.pitch_of(xc), thw[0] * thw[1] * thw[2]
    dev = lambda t: None if t is None else t.cuda().contiguous()
    g_, b_, as_, ab_ = dev(gamma), dev(beta), dev(ada_s), dev(ada_b)
    mean = torch.empty(N * G_, device='cuda'); rstd = torch.empty(N * G_, device='cuda')
    ws = torch.empty(lib.genie_groupnorm_ws_floats(N, C, G_), device='cuda')
    P = G.hip.ptr
    G.hip.check(lib.genie_groupnorm_fwd(P(xc), P(y), N, npix, C, cp, G_, P(g_), P(b_), P(as_), P(ab_), 1e-5, int(act), P(mean), P(rstd), P(ws), G.hip.stream_ptr()), 'gn fwd')
    assert_close_bf16(y, ref, 'gn fwd')
    dgamma, dbeta = torch.zeros(C, device='cuda'), torch.zeros(C, device='cuda')
    das = torch.empty(N, C, device='cuda') if ada else None
    dab = torch.empty(N, C, device='cuda') if ada else None
    G.hip.check(lib.genie_groupnorm_bwd(P(xc), P(dyc), P(dx), N, npix, C, cp, G_, P(g_), P(b_), P(as_), P(ab_), int(act), P(mean), P(rstd),
                                        P(dgamma), P(dbeta), P(das), P(dab), P(ws), G.hip.stream_ptr()), 'gn bwd')
    assert_close_bf16(dx, leaves[0].grad, 'gn dx', rms_frac=5e-3)
    for got, want, nm in [(dgamma, leaves[1].grad, 'dgamma'), (dbeta, leaves[2].grad, 'dbeta')] + ([(das, leaves[3].grad, 'dada_s'), (dab, leaves[4].grad, 'dada_b')] if ada else []):
        torch.testing.assert_close(got.cpu(), want, rtol=2e-3, atol=2e-3 * want.abs().max().item(), msg=nm)


_ONE_PASS_CHECK = r"""
import sys, torch
sys.path[:0] = ['tests', '.', 'open-genie_amd']
from util import assert_close_bf16, bf16_round
from genie import _hip, cl
from oracle import genie_oracle as O
lib = _hip.load_library()
for N, C, thw, act in ((5, 128, (16, 64, 64), True), (70, 256, (4, 16, 16), False), (9, 512, (4, 8, 8), True)):
    torch.manual_seed(5)
    x = bf16_round(torch.randn(N, C, *thw) * 1.5 + 0.3 * torch.arange(N).float().reshape(N, 1, 1, 1, 1))
    gamma, beta = torch.randn(C), torch.randn(C)
    with torch.no_grad():
        ref = O.group_norm(x, 1, gamma, beta)
        ref = O.silu(ref) if act else ref
    xc = cl.to_cl(x.cuda()); y = cl.empty_like_cl(xc)
    npix = thw[0] * thw[1] * thw[2]
    mean = torch.empty(N, device='cuda'); rstd = torch.empty(N, device='cuda')
    ws = torch.empty(lib.genie_groupnorm_ws_floats(N, C, 1), device='cuda')
    g_, b_ = gamma.cuda(), beta.cuda()
    for _ in range(2):                                   # twice: the exchange table is re-armed per launch
        _hip.check(lib.genie_groupnorm_fwd(xc.data_ptr(), y.data_ptr(), N, npix, C, cl.pitch_of(xc), 1, g_.data_ptr(), b_.data_ptr(), None, None, 1e-5,
                                           int(act), mean.data_ptr(), rstd.data_ptr(), ws.data_ptr(), _hip.stream_ptr()), 'gn fwd')
    assert lib.genie_gn_fused_error() == 0
    assert_close_bf16(y, ref, 'gn one-pass fwd')
    xg = x.reshape(N, -1).double()
    torch.testing.assert_close(mean.cpu().double(), xg.mean(-1), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(rstd.cpu().double(), (xg.var(-1, unbiased=False) + 1e-5).rsqrt(), rtol=1e-4, atol=1e-5)
print('one-pass ok')
"""


def test_groupnorm_one_pass_forward_behind_its_switch():
    """The one-pass forward (gn_fused_fwd_kernel, GENIE_GN_FUSED=1; off by default because it measured slower): a clip's slice stays in
    registers between statistics and apply, the blocks of a clip exchange partial sums through device-scope memory.  Run in a process of its
    own (the switch is read once): multi-round, partial last round, many clips per round; outputs, mean / rstd, no exchange ever gave up."""
    import os
    import subprocess
    import sys
    from util import ROOT
    for mode in ('1', '2'):                              # two register sets with the exchange pipelined / one set, two clips de-phased
        env = dict(os.environ, GENIE_GN_FUSED=mode)
        r = subprocess.run([sys.executable, '-c', _ONE_PASS_CHECK], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and 'one-pass ok' in r.stdout, (mode, r.stdout[-2000:] + r.stderr[-4000:])


CONV_CASES = [
    # cin, cout, kernel, stride, causal, size
    (64, 128, (3, 3, 3), (1, 1, 1), False, (2, 4, 8, 8)),
    (128, 64, (3, 3, 3), (1, 1, 1), True, (1, 3, 6, 10)),
    (64, 64, (3, 3, 3), (2, 2, 2), True, (2, 4, 8, 8)),
    (64, 96, (3, 3, 3), (1, 2, 2), True, (1, 3, 9, 7)),
    (128, 256, (1, 1, 1), (1, 1, 1), True, (2, 2, 4, 4)),
    (3, 128, (3, 3, 3), (1, 1, 1), True, (2, 4, 16, 16)),       # stem: small_c path
    (16, 32, (3, 3, 3), (1, 1, 1), False, (1, 2, 5, 5)),        # small_c, cpt = 2
    (32, 16, (3, 3, 3), (2, 2, 2), True, (1, 4, 6, 6)),
    (128, 3, (3, 3, 3), (1, 1, 1), True, (1, 2, 8, 8)),         # head: narrow N tile
    (18, 64, (3, 3, 3), (1, 1, 1), True, (2, 2, 4, 4)),         # dec0: cin pitch 24
    (512, 18, (1, 1, 1), (1, 1, 1), True, (2, 2, 4, 4)),        # enc26
    (40, 72, (1, 3, 3), (1, 1, 1), False, (1, 2, 5, 6)),
]


def _conv_ref(x, w, b, stride, causal):
    from oracle import genie_oracle as O
    return O.causal_conv3d(x, w, b, stride=stride) if causal else O.conv3d_same(x, w, b)


@pytest.mark.parametrize('cin,cout,kernel,stride,causal,size', CONV_CASES)
def test_conv_forward_dgrad(G, cin, cout, kernel, stride, causal, size):
    torch.manual_seed(3)
    n, t, h, w = size
    x = bf16_round(torch.randn(n, cin, t, h, w))
    wt = bf16_round(torch.randn(cout, cin, *kernel) / (cin * kernel[0] * kernel[1] * kernel[2]) ** 0.5)
    b = torch.randn(cout)
    xr = x.clone().requires_grad_(True)
    ref = _conv_ref(xr, wt, b, stride, causal)
    dy = bf16_round(torch.randn_like(ref))
    ref.backward(dy)
    spec = G.conv.causal_spec(cin, cout, kernel, stride) if causal else G.conv.same_spec(cin, cout, kernel)
    wd = wt.cuda()
    wf, wb = G.conv.pack_weight_fwd(wd, spec), G.conv.pack_weight_bwd(wd, spec)
    out = G.conv.conv_forward(G.cl.to_cl(x.cuda()), wf, b.cuda(), spec)
    assert tuple(out.shape) == tuple(ref.shape)
    assert_close_bf16(out, ref, 'conv fwd')
    dx = G.conv.conv_dgrad(G.cl.to_cl(dy.cuda()), wb, spec, (t, h, w))
    assert_close_bf16(dx, xr.grad, 'conv dgrad')
    # channels_last_3d weights pack to the same thing
    wcl = wd.contiguous(memory_format=torch.channels_last_3d)
    assert torch.equal(G.conv.pack_weight_fwd(wcl, spec), wf)


@pytest.mark.parametrize('cin,cf,fac,size', [(64, 32, (2, 2, 2), (1, 2, 4, 4)), (128, 64, (1, 2, 2), (2, 2, 3, 5)), (64, 3, (1, 4, 4), (1, 2, 4, 4)),
                                             (256, 32, (2, 2, 2), (1, 2, 4, 8))])
def test_conv_shuffle_forward_dgrad(G, cin, cf, fac, size, monkeypatch):
    from oracle import genie_oracle as O
    torch.manual_seed(4)
    n, t, h, w = size
    P, Q, R = fac
    cout = cf * P * Q * R
    x = bf16_round(torch.randn(n, cin, t, h, w))
    wt = bf16_round(torch.randn(cout, cin, 3, 3, 3) / (cin * 27) ** 0.5)
    b = torch.randn(cout)
    xr = x.clone().requires_grad_(True)
    ref = O.depth_to_spacetime(O.causal_conv3d(xr, wt, b), P, Q)
    dy = bf16_round(torch.randn_like(ref))
    ref.backward(dy)
    spec = G.conv.causal_spec(cin, cout, (3, 3, 3), shuffle=fac)
    wf = G.conv.pack_weight_fwd(wt.cuda(), spec)
    out = G.conv.conv_forward(G.cl.to_cl(x.cuda()), wf, b.cuda(), spec)
    assert_close_bf16(out, ref, 'shuffle fwd')
    wb = G.conv.pack_weight_bwd(wt.cuda(), spec)          # cf % 8 != 0 takes the un-shuffle path
    dx = G.conv.conv_dgrad(G.cl.to_cl(dy.cuda()), wb, spec, (t, h, w))
    assert_close_bf16(dx, xr.grad, 'shuffle dgrad')
    # the two routes of the backward-data pass: gather through the shuffle (generic kernel) vs genie_unshuffle_cl + plain conv
    monkeypatch.setattr(G.conv, 'UPCONV_DGRAD_UNSHUFFLE', not G.conv.UPCONV_DGRAD_UNSHUFFLE)
    dx2 = G.conv.conv_dgrad(G.cl.to_cl(dy.cuda()), wb, spec, (t, h, w))
    assert_close_bf16(dx2, xr.grad, 'shuffle dgrad (other route)')


TRI_CASES = [
    # cin, cout, kernel, causal, size (n, t, h, w)
    (64, 128, (3, 3, 3), False, (2, 4, 8, 8)),
    (128, 64, (3, 3, 3), True, (1, 3, 16, 16)),
    (64, 160, (3, 3, 3), False, (1, 2, 32, 32)),        # two column tiles, the second partial
    (128, 128, (3, 3, 3), True, (1, 2, 2, 64)),         # one image row per 64 pixels
    (64, 96, (3, 3, 3), False, (3, 1, 5, 8)),           # M = 120: partial row tile, odd H
    (64, 128, (1, 3, 3), False, (1, 2, 3, 128)),        # W = 128
    (192, 72, (3, 1, 3), True, (2, 3, 4, 16)),          # three channel blocks, kh = 1
]


@pytest.mark.parametrize('bm', [128, 256])
@pytest.mark.parametrize('flags', [0, 1, 2, 256])     # 256: one wave per SIMD (4 waves of 128 x 64) for the 256-row tile
@pytest.mark.parametrize('cin,cout,kernel,causal,size', TRI_CASES)
def test_conv_triple_kernel(G, cin, cout, kernel, causal, size, bm, flags, monkeypatch):
    """conv_igemm3.hip (kw-triples share one staged activation tile): forward and backward-data against the oracle, both row
    tiles, counted-wait and drain-every-barrier variants; the library must report that the triple kernel really ran."""
    monkeypatch.setattr(G.conv, 'TRI_BM', bm)
    monkeypatch.setattr(G.conv, 'TRI_FLAGS', flags)
    torch.manual_seed(11)
    n, t, h, w = size
    x = bf16_round(torch.randn(n, cin, t, h, w))
    wt = bf16_round(torch.randn(cout, cin, *kernel) / (cin * kernel[0] * kernel[1] * kernel[2]) ** 0.5)
    b = torch.randn(cout)
    xr = x.clone().requires_grad_(True)
    if causal:
        from oracle import genie_oracle as O
        ref = O.causal_conv3d(xr, wt, b, stride=(1, 1, 1))
        spec = G.conv.causal_spec(cin, cout, kernel)
    else:
        ref = F.conv3d(xr, wt, b, padding=tuple((k - 1) // 2 for k in kernel))
        spec = G.conv.same_spec(cin, cout, kernel)
    dy = bf16_round(torch.randn_like(ref))
    ref.backward(dy)
    wd = wt.cuda()
    lib = G.hip.load_library()
    want = 5 if bm == 256 and 256 % w == 0 else 4
    out = G.conv.conv_forward(G.cl.to_cl(x.cuda()), G.conv.pack_weight_fwd(wd, spec), b.cuda(), spec)
    assert lib.genie_last_conv_variant() == want, lib.genie_last_conv_variant()
    assert_close_bf16(out, ref, 'triple fwd')
    dx = G.conv.conv_dgrad(G.cl.to_cl(dy.cuda()), G.conv.pack_weight_bwd(wd, spec), spec, (t, h, w))
    if cout % 64 == 0:
        assert lib.genie_last_conv_variant() == want, lib.genie_last_conv_variant()
    assert_close_bf16(dx, xr.grad, 'triple dgrad')


PW_CASES = [
    # cin, cout, (n, t, h, w), residual + SiLU epilogue
    (128, 640, (2, 8, 32, 32), False),       # wide tile: 64 x 3 tiles (the last column tile half empty), K = 128
    (192, 1280, (2, 7, 33, 31), True),       # M = 14322: partial last row tile; 56 x 5 tiles; K = 192
    (64, 2048, (1, 5, 32, 33), False),       # K = 64; 21 x 8 = 168 tiles: the last round is partial (the wide "vocabulary" side)
    (512, 128, (2, 16, 32, 40), True),       # 256 x 128 tile: 160 row tiles, one column tile, eight K tiles; wide dgrad
]


@pytest.mark.parametrize('cin,cout,size,fused', PW_CASES)
def test_conv_pointwise_gemm_kernel(G, cin, cout, size, fused):
    """conv_gemm.hip (persistent GEMM for 1x1x1 convolutions / Linear layers): forward (bias, residual, SiLU epilogue) and
    backward-data against F.conv3d; the library must report that the GEMM kernel really ran."""
    torch.manual_seed(13)
    n, t, h, w = size
    x = bf16_round(torch.randn(n, cin, t, h, w))
    wt = bf16_round(torch.randn(cout, cin, 1, 1, 1) / cin ** 0.5)
    b = torch.randn(cout)
    r = bf16_round(torch.randn(n, cout, t, h, w)) if fused else None
    xr = x.clone().requires_grad_(True)
    ref = F.conv3d(xr, wt, b)
    dy = bf16_round(torch.randn_like(ref))
    ref.backward(dy)
    if fused:
        ref = F.silu(ref + r)
    spec = G.conv.same_spec(cin, cout, (1, 1, 1))
    lib = G.hip.load_library()
    out = G.conv.conv_forward(G.cl.to_cl(x.cuda()), G.conv.pack_weight_fwd(wt.cuda(), spec), b.cuda(), spec,
                              resid=G.cl.to_cl(r.cuda()) if fused else None, act=1 if fused else 0)
    assert lib.genie_last_conv_variant() == 6, lib.genie_last_conv_variant()
    assert_close_bf16(out, ref, 'pointwise fwd')
    dx = G.conv.conv_dgrad(G.cl.to_cl(dy.cuda()), G.conv.pack_weight_bwd(wt.cuda(), spec), spec, (t, h, w))
    m = n * t * h * w
    if cout % 64 == 0 and cin >= 96 and -(-m // 256) * -(-cin // (256 if cin > 128 else 128)) >= 160:
        assert lib.genie_last_conv_variant() == 6, lib.genie_last_conv_variant()
    assert_close_bf16(dx, xr.grad, 'pointwise dgrad')


@pytest.mark.parametrize('cin,cout,size', [(128, 128, (2, 4, 16, 16)), (256, 192, (1, 8, 16, 8)), (512, 512, (8, 4, 8, 8))])
def test_conv_triple_split_k(G, cin, cout, size, monkeypatch):
    """Low-resolution layers: too few row tiles to fill the chip, so the kw-triple step table is split over blockIdx.y (fp32
    partial tiles + finish kernel).  Automatic tile choice must pick the 256-row triple kernel; forward (bias) and backward-data."""
    monkeypatch.setattr(G.conv, 'TRI_BM', 0)
    monkeypatch.setattr(G.conv, 'TRI_FLAGS', 0)
    torch.manual_seed(14)
    n, t, h, w = size
    x = bf16_round(torch.randn(n, cin, t, h, w))
    wt = bf16_round(torch.randn(cout, cin, 3, 3, 3) / (cin * 27) ** 0.5)
    b = torch.randn(cout)
    xr = x.clone().requires_grad_(True)
    ref = F.conv3d(xr, wt, b, padding=1)
    dy = bf16_round(torch.randn_like(ref))
    ref.backward(dy)
    spec = G.conv.same_spec(cin, cout, (3, 3, 3))
    lib = G.hip.load_library()
    out = G.conv.conv_forward(G.cl.to_cl(x.cuda()), G.conv.pack_weight_fwd(wt.cuda(), spec), b.cuda(), spec)
    assert lib.genie_last_conv_variant() == 7, lib.genie_last_conv_variant()
    assert_close_bf16(out, ref, 'split-K triple fwd')
    r = bf16_round(torch.randn_like(ref))                 # the finish kernel's residual + SiLU branch
    out = G.conv.conv_forward(G.cl.to_cl(x.cuda()), G.conv.pack_weight_fwd(wt.cuda(), spec), b.cuda(), spec, resid=G.cl.to_cl(r.cuda()), act=1)
    assert_close_bf16(out, F.silu(ref.detach() + r), 'split-K triple fwd + residual + SiLU')
    dx = G.conv.conv_dgrad(G.cl.to_cl(dy.cuda()), G.conv.pack_weight_bwd(wt.cuda(), spec), spec, (t, h, w))
    if cout % 64 == 0:
        assert lib.genie_last_conv_variant() == 7, lib.genie_last_conv_variant()
    assert_close_bf16(dx, xr.grad, 'split-K triple dgrad')


@pytest.mark.parametrize('cin,cout,causal,size', [(64, 128, False, (2, 9, 64, 64)), (128, 192, True, (2, 5, 64, 64)), (64, 128, True, (3, 11, 32, 32))])
def test_conv_triple_persistent_kernel(G, cin, cout, causal, size, monkeypatch):
    """More 256-row tiles than CUs: the persistent kw-triple kernel walks several tiles per block (the DMA stream runs through the
    tile boundaries; the last round of tiles is partial).  Forward with bias + residual, backward-data."""
    monkeypatch.setattr(G.conv, 'TRI_BM', 256)
    monkeypatch.setattr(G.conv, 'TRI_FLAGS', 128)
    torch.manual_seed(15)
    n, t, h, w = size
    x = bf16_round(torch.randn(n, cin, t, h, w))
    wt = bf16_round(torch.randn(cout, cin, 3, 3, 3) / (cin * 27) ** 0.5)
    b = torch.randn(cout)
    r = bf16_round(torch.randn(n, cout, t, h, w))
    xr = x.clone().requires_grad_(True)
    if causal:
        from oracle import genie_oracle as O
        ref = O.causal_conv3d(xr, wt, b, stride=(1, 1, 1))
        spec = G.conv.causal_spec(cin, cout, (3, 3, 3))
    else:
        ref = F.conv3d(xr, wt, b, padding=1)
        spec = G.conv.same_spec(cin, cout, (3, 3, 3))
    dy = bf16_round(torch.randn_like(ref))
    ref.backward(dy)
    lib = G.hip.load_library()
    out = G.conv.conv_forward(G.cl.to_cl(x.cuda()), G.conv.pack_weight_fwd(wt.cuda(), spec), b.cuda(), spec, resid=G.cl.to_cl(r.cuda()))
    assert lib.genie_last_conv_variant() == 5, lib.genie_last_conv_variant()
    assert_close_bf16(out, ref + r, 'persistent triple fwd')
    dx = G.conv.conv_dgrad(G.cl.to_cl(dy.cuda()), G.conv.pack_weight_bwd(wt.cuda(), spec), spec, (t, h, w))
    assert_close_bf16(dx, xr.grad, 'persistent triple dgrad')


def test_conv_triple_shuffle_and_residual(G, monkeypatch):
    """The triple kernel under the depth-to-space-time store pattern (upsample conv) and with the residual add in the epilogue."""
    from oracle import genie_oracle as O
    monkeypatch.setattr(G.conv, 'TRI_BM', 128)
    torch.manual_seed(12)
    x = bf16_round(torch.randn(1, 64, 2, 4, 8))
    wt = bf16_round(torch.randn(32 * 8, 64, 3, 3, 3) / (64 * 27) ** 0.5)
    b = torch.randn(32 * 8)
    ref = O.depth_to_spacetime(O.causal_conv3d(x, wt, b), 2, 2)
    spec = G.conv.causal_spec(64, 256, (3, 3, 3), shuffle=(2, 2, 2))
    out = G.conv.conv_forward(G.cl.to_cl(x.cuda()), G.conv.pack_weight_fwd(wt.cuda(), spec), b.cuda(), spec)
    assert G.hip.load_library().genie_last_conv_variant() == 4
    assert_close_bf16(out, ref, 'triple shuffle fwd')
    r = bf16_round(torch.randn(2, 128, 3, 8, 8))
    x2 = bf16_round(torch.randn(2, 64, 3, 8, 8))
    w2 = bf16_round(torch.randn(128, 64, 3, 3, 3) / 42.)
    spec2 = G.conv.same_spec(64, 128, (3, 3, 3))
    out2 = G.conv.conv_forward(G.cl.to_cl(x2.cuda()), G.conv.pack_weight_fwd(w2.cuda(), spec2), None, spec2, resid=G.cl.to_cl(r.cuda()))
    assert G.hip.load_library().genie_last_conv_variant() == 4
    assert_close_bf16(out2, O.conv3d_same(x2, w2, None) + r, 'triple conv + resid')


def test_conv_residual_epilogue(G):
    torch.manual_seed(5)
    x = bf16_round(torch.randn(2, 64, 3, 6, 6))
    r = bf16_round(torch.randn(2, 128, 3, 6, 6))
    wt = bf16_round(torch.randn(128, 64, 3, 3, 3) / 42.)
    from oracle import genie_oracle as O
    ref = O.conv3d_same(x, wt, None) + r
    spec = G.conv.same_spec(64, 128, (3, 3, 3))
    out = G.conv.conv_forward(G.cl.to_cl(x.cuda()), G.conv.pack_weight_fwd(wt.cuda(), spec), None, spec, resid=G.cl.to_cl(r.cuda()))
    assert_close_bf16(out, ref, 'conv + resid')


WGRAD_CASES = CONV_CASES + [
    (128, 128, (3, 3, 3), (1, 1, 1), False, (2, 4, 16, 16)),
    (256, 128, (3, 3, 3), (1, 1, 1), True, (1, 3, 8, 8)),
]


@pytest.mark.parametrize('cin,cout,kernel,stride,causal,size', WGRAD_CASES)
@pytest.mark.parametrize('wfmt', ['contiguous', 'channels_last_3d'])
def test_conv_wgrad(G, cin, cout, kernel, stride, causal, size, wfmt):
    torch.manual_seed(6)
    n, t, h, w = size
    x = bf16_round(torch.randn(n, cin, t, h, w))
    wt = torch.randn(cout, cin, *kernel, requires_grad=True)
    b = torch.randn(cout, requires_grad=True)
    ref = _conv_ref(x, wt, b, stride, causal)
    dy = bf16_round(torch.randn_like(ref))
    ref.backward(dy)
    spec = G.conv.causal_spec(cin, cout, kernel, stride) if causal else G.conv.same_spec(cin, cout, kernel)
    dw = torch.zeros(cout, cin, *kernel, device='cuda')
    if wfmt == 'channels_last_3d':
        dw = dw.contiguous(memory_format=torch.channels_last_3d)
    db = torch.zeros(cout, device='cuda')
    G.conv.conv_wgrad(G.cl.to_cl(x.cuda()), G.cl.to_cl(dy.cuda()), spec, dw, db)
    scale = wt.grad.abs().max().item()
    torch.testing.assert_close(dw.cpu(), wt.grad, rtol=1e-3, atol=1e-3 * scale)
    torch.testing.assert_close(db.cpu(), b.grad, rtol=1e-3, atol=1e-3 * b.grad.abs().max().item())
    # accumulation semantics
    G.conv.conv_wgrad(G.cl.to_cl(x.cuda()), G.cl.to_cl(dy.cuda()), spec, dw, None)
    torch.testing.assert_close(dw.cpu(), 2 * wt.grad, rtol=1e-3, atol=2e-3 * scale)


@pytest.mark.parametrize('cin,cout,size', [(256, 512, (2, 4, 16, 16)), (512, 256, (1, 3, 9, 11)), (320, 264, (1, 2, 5, 29)), (256, 256 * 260, (1, 1, 3, 97))])
def test_conv_wgrad_pointwise_kernel(G, cin, cout, size, monkeypatch):
    """conv_wgrad_pw.hip (pointwise weight gradient, 256 x 256 tile, transposing reads on both operands, VALU bias gradient) against
    autograd of F.conv3d: split-K with atomics (few output tiles), partial tiles in both channel dimensions and in the last 32-pixel
    chunk, and the single-split read-modify-write path (260 output tiles, the shape class of the vocabulary head)."""
    torch.manual_seed(14)
    n, t, h, w = size
    x = bf16_round(torch.randn(n, cin, t, h, w))
    wt = torch.randn(cout, cin, 1, 1, 1, requires_grad=True)
    b = torch.randn(cout, requires_grad=True)
    ref = F.conv3d(x, wt, b)
    dy = bf16_round(torch.randn_like(ref))
    ref.backward(dy)
    spec = G.conv.same_spec(cin, cout, (1, 1, 1))
    dw = torch.zeros(cout, cin, 1, 1, 1, device='cuda')
    db = torch.zeros(cout, device='cuda')
    monkeypatch.setattr(G.conv, 'WGRAD_PW', 2)           # few output tiles would go to the generic kernel by default
    G.conv.conv_wgrad(G.cl.to_cl(x.cuda()), G.cl.to_cl(dy.cuda()), spec, dw, db)
    assert G.hip.load_library().genie_last_conv_variant() == 13
    torch.testing.assert_close(dw.cpu(), wt.grad, rtol=1e-3, atol=1e-3 * wt.grad.abs().max().item())
    torch.testing.assert_close(db.cpu(), b.grad, rtol=1e-3, atol=1e-3 * b.grad.abs().max().item())
    G.conv.conv_wgrad(G.cl.to_cl(x.cuda()), G.cl.to_cl(dy.cuda()), spec, dw, None)      # accumulates
    torch.testing.assert_close(dw.cpu(), 2 * wt.grad, rtol=1e-3, atol=2e-3 * wt.grad.abs().max().item())


@pytest.mark.parametrize('cin,cout,causal,size', [(3, 128, True, (2, 5, 10, 64)), (4, 128, False, (1, 3, 8, 32)), (1, 128, True, (1, 2, 4, 128)),
                                                  (128, 3, True, (2, 4, 9, 64)), (128, 2, False, (1, 3, 6, 32)), (128, 1, True, (1, 2, 5, 128)),
                                                  (3, 128, True, (4, 16, 64, 64)), (128, 3, True, (4, 16, 64, 64)),
                                                  (3, 256, True, (2, 5, 10, 64)), (256, 3, True, (2, 4, 9, 64)), (4, 256, False, (1, 3, 8, 32)), (384, 2, True, (1, 3, 6, 64))])
def test_conv_narrow_wgrad_kernel(G, cin, cout, causal, size):
    """conv_narrow.hip, weight gradients of the stem ((<= 4) -> 128) and head (128 -> (<= 4)) convolutions in ONE pass over the
    128-channel tensor (im2col tile of the narrow tensor built in LDS, transposing reads on both MFMA operands), against autograd of
    the fp32 convolution: causal and symmetric padding, every supported width, bias gradients, accumulation into non-zero buffers."""
    from oracle import genie_oracle as O
    torch.manual_seed(21)
    n, t, h, w = size
    x = bf16_round(torch.randn(n, cin, t, h, w))
    wt = torch.randn(cout, cin, 3, 3, 3, requires_grad=True)
    b = torch.randn(cout, requires_grad=True)
    ref = O.causal_conv3d(x, wt, b) if causal else F.conv3d(x, wt, b, padding=1)
    dy = bf16_round(torch.randn_like(ref))
    ref.backward(dy)
    spec = G.conv.causal_spec(cin, cout, (3, 3, 3)) if causal else G.conv.same_spec(cin, cout, (3, 3, 3))
    xc, dyc = G.cl.to_cl(x.cuda()), G.cl.to_cl(dy.cuda())
    assert G.conv.narrow_wgrad_ok(spec, xc, dyc)
    sw, sb = wt.grad.abs().max().item(), b.grad.abs().max().item()
    # the gradient buffer in the layout the modules keep (channels_last_3d), dense (co, ci, taps), and an arbitrary strided view (through temporaries)
    for layout in ('channels_last', 'dense', 'strided'):
        if layout == 'channels_last':
            dw = torch.full((cout, cin, 3, 3, 3), 0.5, device='cuda').contiguous(memory_format=torch.channels_last_3d)
        elif layout == 'dense':
            dw = torch.full((cout, cin, 3, 3, 3), 0.5, device='cuda')
        else:
            dw = torch.full((cout, cin, 3, 3, 6), 0.5, device='cuda')[..., ::2]
        db = torch.full((cout,), -1.0, device='cuda')
        G.conv.PROFILER = prof = G.conv.LaunchProfiler()
        try:
            G.conv.conv_wgrad(xc, dyc, spec, dw, db)
        finally:
            G.conv.PROFILER = None
        assert 'conv_narrow_wgrad_kernel' in prof.summary(), list(prof.summary())
        torch.testing.assert_close(dw.cpu() - 0.5, wt.grad, rtol=1e-3, atol=1e-3 * sw, msg=lambda m: f'{layout}: {m}')
        torch.testing.assert_close(db.cpu() + 1.0, b.grad, rtol=1e-3, atol=1e-3 * sb, msg=lambda m: f'{layout}: {m}')


TRI_WGRAD_CASES = [
    # cin, cout, kernel, causal, size (n, t, h, w), shuffle
    (64, 128, (3, 3, 3), False, (2, 4, 8, 8), None),
    (128, 128, (3, 3, 3), True, (1, 3, 16, 16), None),
    (192, 160, (3, 3, 3), False, (1, 2, 4, 32), None),       # partial co / ci tiles
    (64, 64, (3, 3, 3), True, (1, 2, 3, 64), None),          # one image row per chunk
    (64, 96, (1, 3, 3), False, (1, 1, 3, 8), None),          # M = 24: a single partial chunk
    (128, 64, (3, 1, 3), True, (3, 5, 2, 16), None),         # M = 480: last chunk partial, kh = 1
    (64, 32 * 8, (3, 3, 3), True, (1, 2, 4, 8), (2, 2, 2)),  # upsample conv: dy is the shuffled high-resolution gradient
    (128, 64 * 4, (3, 3, 3), True, (2, 2, 4, 16), (1, 2, 2)),
    (128, 128, (3, 3, 3), False, (3, 5, 16, 32), None),       # several splits, (t, h) wrap inside a block's range, every dt / dh border
    (256, 128, (3, 3, 3), True, (2, 3, 64, 64), None),       # W = 64: one image row per chunk, two ci tiles
    (72, 200, (3, 3, 3), False, (1, 2, 8, 16), None),        # channel counts that are multiples of 8 only: OOB lanes of the descriptor
]


@pytest.mark.parametrize('cin,cout,kernel,causal,size,wfmt', [
    (64, 128, (3, 3, 3), True, (1, 3, 6, 128), 'cl'),
    (128, 64, (3, 3, 3), False, (2, 2, 5, 128), 'contig'),
    (256, 256, (3, 3, 3), True, (1, 4, 16, 128), 'cl'),        # the LatentAction feed-forward conv of BASELINE configs[4], fewer rows
    (72, 200, (1, 3, 3), False, (1, 2, 4, 128), 'cl'),         # partial channel tiles, kt = 1
    (64, 64, (3, 1, 3), True, (1, 3, 2, 128), 'cl'),           # kh = 1
])
def test_conv_wgrad_128_wide_as_two_windows(G, cin, cout, kernel, causal, size, wfmt, monkeypatch):
    """W = 128 (BASELINE configs[4]): two 64-column windows on the lean kw-triple kernel (GenieWgradDesc.row_px / px0) + the two seam products, against
    autograd of the oracle and against the generic kernel on the whole layer (GENIE_WGRAD_WINDOWS=0, what rounds 1-5 ran); the bias gradient is the sum
    of the two windows'.  A gradient that ignored the seam would be off by ~1 / 64 of the kw = +-1 taps: the bound below is 1e-3 of the maximum."""
    from oracle import genie_oracle as O
    torch.manual_seed(29)
    n, t, h, w = size
    x = bf16_round(torch.randn(n, cin, t, h, w))
    wt = torch.randn(cout, cin, *kernel, requires_grad=True)
    b = torch.randn(cout, requires_grad=True)
    if causal:
        ref = O.causal_conv3d(x, wt, b)
        spec = G.conv.causal_spec(cin, cout, kernel)
    else:
        ref = F.conv3d(x, wt, b, padding=tuple((k - 1) // 2 for k in kernel))
        spec = G.conv.same_spec(cin, cout, kernel)
    dy = bf16_round(torch.randn_like(ref))
    ref.backward(dy)
    xc, dyc = G.cl.to_cl(x.cuda()), G.cl.to_cl(dy.cuda())
    assert G.conv.wgrad_windows_ok(spec, xc, dyc)
    mk = lambda: (torch.zeros(cout, cin, *kernel, device='cuda').contiguous(memory_format=torch.channels_last_3d) if wfmt == 'cl'
                  else torch.zeros(cout, cin, *kernel, device='cuda'))
    dw, db = mk(), torch.zeros(cout, device='cuda')
    G.conv.conv_wgrad(xc, dyc, spec, dw, db)
    amax = wt.grad.abs().max().item()
    torch.testing.assert_close(dw.cpu(), wt.grad, rtol=1e-3, atol=1e-3 * amax)
    torch.testing.assert_close(db.cpu(), b.grad, rtol=1e-3, atol=1e-3 * b.grad.abs().max().item())
    # the seam is there: per-tap error of the kw = +-1 taps no larger than that of the kw = 0 taps
    err = (dw.cpu() - wt.grad).abs().amax((0, 1, 2, 3))
    assert err[0] < 4 * err[1] + 1e-4 * amax and err[2] < 4 * err[1] + 1e-4 * amax, err
    G.conv.conv_wgrad(xc, dyc, spec, dw, None)                                           # accumulates
    torch.testing.assert_close(dw.cpu(), 2 * wt.grad, rtol=1e-3, atol=2e-3 * amax)
    monkeypatch.setattr(G.conv, 'WGRAD_WINDOWS', False)
    dw2, db2 = mk(), torch.zeros(cout, device='cuda')
    G.conv.conv_wgrad(xc, dyc, spec, dw2, db2)
    torch.testing.assert_close(dw2, dw * 0.5, rtol=1e-3, atol=1e-3 * amax)
    torch.testing.assert_close(db2, db, rtol=1e-3, atol=1e-3 * b.grad.abs().max().item())
    # deterministic mode: one owner per element in both windows, launches serialised on the stream -> bit-identical run to run
    monkeypatch.setattr(G.conv, 'WGRAD_WINDOWS', True)
    old = G.conv.set_deterministic(True)
    try:
        runs = []
        for _ in range(2):
            dwd, dbd = mk(), torch.zeros(cout, device='cuda')
            G.conv.conv_wgrad(xc, dyc, spec, dwd, dbd)
            runs.append((dwd, dbd))
    finally:
        G.conv.set_deterministic(old)
    assert torch.equal(runs[0][0], runs[1][0]) and torch.equal(runs[0][1], runs[1][1])
    torch.testing.assert_close(runs[0][0].cpu(), wt.grad, rtol=1e-3, atol=1e-3 * amax)


@pytest.mark.parametrize('cin,cout,kernel,causal,size,shuffle', TRI_WGRAD_CASES)
def test_conv_wgrad_triple_kernel(G, cin, cout, kernel, causal, size, shuffle, monkeypatch):
    """conv_wgrad3.hip (one block per kw-triple: shared dy tile and x image) against autograd of the oracle."""
    from oracle import genie_oracle as O
    torch.manual_seed(13)
    n, t, h, w = size
    x = bf16_round(torch.randn(n, cin, t, h, w))
    wt = torch.randn(cout, cin, *kernel, requires_grad=True)
    b = torch.randn(cout, requires_grad=True)
    if causal:
        ref = O.causal_conv3d(x, wt, b)
        spec = G.conv.causal_spec(cin, cout, kernel, shuffle=shuffle)
    else:
        ref = F.conv3d(x, wt, b, padding=tuple((k - 1) // 2 for k in kernel))
        spec = G.conv.same_spec(cin, cout, kernel)
    if shuffle is not None:
        ref = O.depth_to_spacetime(ref, shuffle[0], shuffle[1])
    dy = bf16_round(torch.randn_like(ref))
    ref.backward(dy)
    dw = torch.zeros(cout, cin, *kernel, device='cuda').contiguous(memory_format=torch.channels_last_3d)
    db = torch.zeros(cout, device='cuda')
    monkeypatch.setattr(G.conv, 'TRI_WGRAD', 2)          # force the triple kernel whatever the problem size
    G.conv.conv_wgrad(G.cl.to_cl(x.cuda()), G.cl.to_cl(dy.cuda()), spec, dw, db)
    # whole image rows per 64-pixel chunk and a plain (un-shuffled) dy: the lean main loop (buffer-addressed LDS-DMA, wgrad3l_kernel, 14);
    # everything else: the general kernel (11)
    lean = shuffle is None and (h * w) % 64 == 0 and 64 // w <= h
    assert G.hip.load_library().genie_last_conv_variant() == (14 if lean else 11)
    torch.testing.assert_close(dw.cpu(), wt.grad, rtol=1e-3, atol=1e-3 * wt.grad.abs().max().item())
    torch.testing.assert_close(db.cpu(), b.grad, rtol=1e-3, atol=1e-3 * b.grad.abs().max().item())
    G.conv.conv_wgrad(G.cl.to_cl(x.cuda()), G.cl.to_cl(dy.cuda()), spec, dw, None)      # accumulates
    torch.testing.assert_close(dw.cpu(), 2 * wt.grad, rtol=1e-3, atol=2e-3 * wt.grad.abs().max().item())


@pytest.mark.parametrize('cin,cf,fac,size', [(128, 64, (1, 2, 2), (2, 2, 4, 16)), (64, 32, (2, 2, 2), (1, 3, 8, 8)), (256, 128, (2, 2, 2), (2, 2, 8, 16)), (72, 24, (1, 2, 2), (1, 2, 2, 32))])
def test_conv_wgrad_unshuffled_dy(G, cin, cf, fac, size):
    """Upsample convs in backward: ONE un-shuffle of the output gradient (genie_unshuffle_cl, sub-pixel-major channels) feeds the plain
    backward-data conv AND the lean kw-triple weight-gradient kernel, which permutes its rows back to the natural '(c p q r)' weight
    order (GenieWgradDesc.dy_unshuffled); both against autograd of conv + depth-to-space-time rearrange."""
    from oracle import genie_oracle as O
    torch.manual_seed(17)
    n, t, h, w = size
    P, Q, R = fac
    cout = cf * P * Q * R
    x = bf16_round(torch.randn(n, cin, t, h, w)).requires_grad_(True)
    wt = bf16_round(torch.randn(cout, cin, 3, 3, 3) / (cin * 27) ** 0.5).requires_grad_(True)
    b = torch.randn(cout, requires_grad=True)
    ref = O.depth_to_spacetime(O.causal_conv3d(x, wt, b), P, Q)
    dy = bf16_round(torch.randn_like(ref))
    ref.backward(dy)
    spec = G.conv.causal_spec(cin, cout, (3, 3, 3), shuffle=fac)
    xc, dyc = G.cl.to_cl(x.detach().cuda()), G.cl.to_cl(dy.cuda())
    assert G.conv.wgrad_unshuffled_ok(spec, xc)
    dyu = G.conv.unshuffle_dy(dyc, spec)
    assert tuple(dyu.shape) == (n, cout, t, h, w)
    dw = torch.zeros(cout, cin, 3, 3, 3, device='cuda').contiguous(memory_format=torch.channels_last_3d)
    db = torch.zeros(cout, device='cuda')
    G.conv.conv_wgrad(xc, dyu, spec, dw, db, True)
    assert G.hip.load_library().genie_last_conv_variant() == 14
    torch.testing.assert_close(dw.cpu(), wt.grad, rtol=1e-3, atol=1e-3 * wt.grad.abs().max().item())
    torch.testing.assert_close(db.cpu(), b.grad, rtol=1e-3, atol=1e-3 * b.grad.abs().max().item())
    dx = G.conv.conv_dgrad(dyc, G.conv.pack_weight_bwd(wt.detach().cuda(), spec), spec, (t, h, w), dy_unshuffled=dyu)
    assert_close_bf16(dx, x.grad, 'dgrad from the shared un-shuffled gradient')
    # and through the autograd binding: same numbers as the gather-through-the-shuffle path
    from genie.module.video import DepthToSpaceTimeUpsample
    m = DepthToSpaceTimeUpsample(cin, cf, time_factor=P, space_factor=Q, kernel_size=3).cuda()
    conv = m.go_up[0].conv3d
    with torch.no_grad():
        conv.weight.copy_(wt.detach())
        conv.bias.copy_(b.detach())
    xg = x.detach().cuda().requires_grad_(True)
    m(xg).backward(dy.cuda())
    torch.testing.assert_close(conv.weight.grad.cpu(), wt.grad, rtol=1e-3, atol=1e-3 * wt.grad.abs().max().item())
    assert_close_bf16(xg.grad, x.grad, 'module backward')


@pytest.mark.parametrize('cin,cf,fac,size', [(64, 32, (2, 2, 2), (1, 2, 4, 4)), (128, 64, (1, 2, 2), (2, 2, 3, 5)), (64, 256, (2, 2, 2), (1, 2, 4, 4)),
                                             (64, 3, (1, 4, 4), (1, 2, 4, 4))])
def test_conv_shuffle_wgrad(G, cin, cf, fac, size):
    from oracle import genie_oracle as O
    torch.manual_seed(7)
    n, t, h, w = size
    P, Q, R = fac
    cout = cf * P * Q * R
    x = bf16_round(torch.randn(n, cin, t, h, w))
    wt = torch.randn(cout, cin, 3, 3, 3, requires_grad=True)
    b = torch.randn(cout, requires_grad=True)
    ref = O.depth_to_spacetime(O.causal_conv3d(x, wt, b), P, Q)
    dy = bf16_round(torch.randn_like(ref))
    ref.backward(dy)
    spec = G.conv.causal_spec(cin, cout, (3, 3, 3), shuffle=fac)
    dw = torch.zeros(cout, cin, 3, 3, 3, device='cuda').contiguous(memory_format=torch.channels_last_3d)
    db = torch.zeros(cout, device='cuda')
    G.conv.conv_wgrad(G.cl.to_cl(x.cuda()), G.cl.to_cl(dy.cuda()), spec, dw, db)
    torch.testing.assert_close(dw.cpu(), wt.grad, rtol=1e-3, atol=1e-3 * wt.grad.abs().max().item())
    torch.testing.assert_close(db.cpu(), b.grad, rtol=1e-3, atol=1e-3 * b.grad.abs().max().item())


@pytest.mark.parametrize('d,ncb,ntok,scale', [(18, 1, 64, 0.05), (18, 1, 40, 0.002), (8, 1, 200, 0.3), (6, 3, 50, 0.01), (10, 1, 128, 0.02), (1, 1, 16, 0.01)])
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_lfq_loss_and_grad(G, d, ncb, ntok, scale, dtype):
    """Training loss of LFQ (entropy over all 2^d codes with the clamp inside the log + commitment) and its
    gradient, against the reference formulation evaluated by the oracle (which materialises N x 2^d)."""
    from oracle import genie_oracle as O
    torch.manual_seed(8)
    z = (torch.randn(ntok, ncb * d) * scale).to(dtype)
    z[0] = 0.
    zr = z.float().clone().requires_grad_(True)
    (_, _), ref_loss = O.lfq_forward(zr.reshape(1, ntok, ncb * d), {}, '', d, ncb, training=True, beta=100.)
    ref_loss.backward()
    pitch = (ncb * d + 7) & ~7
    zz = torch.zeros(ntok, pitch, dtype=dtype, device='cuda')
    zz[:, :ncb * d] = z.cuda()
    lib = G.hip.load_library()
    ws = torch.empty(lib.genie_lfq_loss_ws_floats(ntok, ncb, d), device='cuda')
    loss4 = torch.zeros(4, device='cuda')
    dz = torch.zeros(ntok, ncb * d, device='cuda')
    G.hip.check(lib.genie_lfq_loss(zz.data_ptr(), G.hip.GENIE_F32 if dtype == torch.float32 else G.hip.GENIE_BF16, ntok, ncb, d, pitch, 100., 0.25, 0.1, 1.,
                                   ws.data_ptr(), loss4.data_ptr(), dz.data_ptr(), G.hip.stream_ptr()), 'lfq loss')
    assert abs(loss4[0].item() - ref_loss.item()) < 1e-5 + 1e-5 * abs(ref_loss.item()), (loss4.tolist(), ref_loss.item())
    g = zr.grad
    err = (dz.cpu() - g).abs().max().item()
    assert err <= 2e-3 * g.abs().max().item() + 1e-7, (err, g.abs().max().item())


@pytest.mark.parametrize('c,o,ks,tf,sf,size', [(16, None, 3, 2, 2, (2, 4, 8, 8)), (40, 24, 3, 1, 2, (1, 3, 7, 9)), (128, None, (3, 3, 3), 2, (2, 2), (1, 5, 6, 6)),
                                                (8, 3, 5, 2, 2, (2, 6, 9, 8))])
def test_blur_pool3d(G, c, o, ks, tf, sf, size):
    """BlurPooling3d (video.py:487-537, the dense-conv behaviour of num_groups = 1: every output channel = strided blur of the
    channel sum), forward and backward against the oracle's F.conv3d restatement."""
    from genie.module.video import BlurPooling3d
    from oracle import genie_oracle as O
    torch.manual_seed(14)
    n, t, h, w = size
    x = bf16_round(torch.randn(n, c, t, h, w))
    xr = x.clone().requires_grad_(True)
    ref = O.blur_pool3d(xr, ks, tf, sf, out_channels=o)
    dy = bf16_round(torch.randn_like(ref))
    ref.backward(dy)
    m = BlurPooling3d(c, ks, out_channels=o, time_factor=tf, space_factor=sf).cuda()
    xd = x.cuda().requires_grad_(True)
    out = m(xd)
    assert tuple(out.shape) == tuple(ref.shape)
    assert_close_bf16(out, ref, 'blur pool fwd')
    out.backward(dy.cuda())
    assert_close_bf16(xd.grad, xr.grad, 'blur pool bwd', rms_frac=4e-3)


@pytest.mark.parametrize('c,o,g,tf,sf,size', [(12, None, 3, 2, 2, (2, 4, 6, 6)), (40, 20, 5, 1, 2, (1, 3, 8, 8)), (16, None, 4, 2, 1, (2, 4, 5, 5)),
                                              (64, 32, 4, 2, 2, (1, 4, 8, 8)), (6, 6, 6, 2, 2, (2, 2, 4, 4))])
def test_blur_pool3d_groups_of_any_width(G, c, o, g, tf, sf, size):
    """BlurPooling3d(num_groups = g) (video.py:520-533: F.conv3d(groups = g) with the Pascal kernel repeated over (o, c / g)): every output channel of a
    group is the strided blur of the SUM of that group's input channels.  Groups of a multiple of 8 channels run on channel-slice views; any other width
    (rounds 1-5 raised) on an aligned copy of the slice -- forward and backward against the oracle's grouped conv."""
    from genie.module.video import BlurPooling3d
    from oracle import genie_oracle as O
    torch.manual_seed(19)
    n, t, h, w = size
    x = bf16_round(torch.randn(n, c, t, h, w))
    xr = x.clone().requires_grad_(True)
    ref = O.blur_pool3d(xr, 3, tf, sf, num_groups=g, out_channels=o)
    dy = bf16_round(torch.randn_like(ref))
    ref.backward(dy)
    m = BlurPooling3d(c, 3, out_channels=o, time_factor=tf, space_factor=sf, num_groups=g).cuda()
    xd = x.cuda().requires_grad_(True)
    out = m(xd)
    assert tuple(out.shape) == tuple(ref.shape)
    assert_close_bf16(out, ref, 'grouped blur pool fwd')
    out.backward(dy.cuda())
    assert_close_bf16(xd.grad, xr.grad, 'grouped blur pool bwd', rms_frac=4e-3)
    with pytest.raises(ValueError):
        BlurPooling3d(c, 3, num_groups=c + 1).cuda()(x.cuda())


def test_residual_block_with_blur_downsample(G):
    """VideoResidualBlock(downsample=...) -- the README / test_tokenizer.py configuration of the reference (video.py:588-648):
    blur pooling in both branches -- forward against the oracle."""
    from genie.module.video import VideoResidualBlock
    from oracle import genie_oracle as O
    torch.manual_seed(15)
    m = VideoResidualBlock(64, 128, downsample=(2, 2))
    with torch.no_grad():
        for p in m.parameters():
            if p.dim() >= 2:
                p.copy_(bf16_round(p * 0.1))
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    x = bf16_round(torch.randn(2, 64, 4, 8, 8))
    ref = O.video_residual_block(x, sd, '', 64, 128, downsample=(2, 2))
    out = m.cuda()(x.cuda())
    assert tuple(out.shape) == tuple(ref.shape) == (2, 128, 2, 4, 4)
    rel = ((out.float().cpu() - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item()
    assert rel < 2e-2, rel


@pytest.mark.parametrize('cin,causal,size', [(3, True, (2, 5, 10, 64)), (3, False, (1, 3, 7, 32)), (4, True, (1, 2, 4, 128)), (1, True, (2, 4, 9, 64)),
                                              (3, True, (8, 16, 64, 64))])
def test_conv_narrow_in_kernel(G, cin, causal, size):
    """conv_narrow.hip: the HBM-bound stem CausalConv3d(<= 4 -> 128) forward, and the backward-data pass of the head conv (128 -> <= 4)
    as the same kernel with flipped taps, both through the module-level path (functional._Conv3dFn) against the oracle; the packed
    bias (bf16 hi + lo) must reproduce the fp32 bias."""
    from genie import functional as GF
    torch.manual_seed(17)
    n, t, h, w = size
    kernel = (3, 3, 3)
    # ---- forward: cin -> 128 ----
    x = bf16_round(torch.randn(n, cin, t, h, w))
    wt = bf16_round(torch.randn(128, cin, *kernel) / (cin * 27) ** 0.5)
    b = torch.randn(128) * 3
    spec = G.conv.causal_spec(cin, 128, kernel) if causal else G.conv.same_spec(cin, 128, kernel)
    xc = G.cl.to_cl(x.cuda())
    assert G.conv.narrow_fwd_ok(spec, xc)
    op = GF.ConvOp(spec)
    wd, bd = wt.cuda().requires_grad_(True), b.cuda().requires_grad_(True)
    G.conv.PROFILER = prof = G.conv.LaunchProfiler()
    try:
        out = GF.conv3d(xc, wd, bd, op)
    finally:
        G.conv.PROFILER = None
    assert 'conv_narrow_in_kernel' in prof.summary(), list(prof.summary())
    ref = _conv_ref(x, wt, b, (1, 1, 1), causal)
    assert tuple(out.shape) == tuple(ref.shape)
    assert_close_bf16(out, ref, 'narrow conv fwd')
    # the generic kernel on the same inputs agrees as well (same algorithm, other tiling)
    gen = G.conv.conv_forward(xc, G.conv.pack_weight_fwd(wd.detach(), spec), bd.detach(), spec)
    assert_close_bf16(out, gen.float().cpu(), 'narrow vs generic', rel=2 ** -7, rms_frac=3e-3)
    # weight / bias gradients still come from the generic wgrad kernel
    dy = bf16_round(torch.randn_like(ref))
    out.backward(dy.cuda())
    wr, br = wt.clone().requires_grad_(True), b.clone().requires_grad_(True)
    _conv_ref(x, wr, br, (1, 1, 1), causal).backward(dy)
    assert_close_bf16(wd.grad, wr.grad, 'narrow conv dW', rel=2 ** -6, rms_frac=1e-2)
    assert_close_bf16(bd.grad, br.grad, 'narrow conv db', rel=2 ** -6, rms_frac=1e-2)
    # ---- backward-data of 128 -> cin ----
    if n * t * h * w > 100000:
        return
    spec2 = G.conv.causal_spec(128, cin, kernel) if causal else G.conv.same_spec(128, cin, kernel)
    x2 = bf16_round(torch.randn(n, 128, t, h, w)).requires_grad_(True)
    w2 = bf16_round(torch.randn(cin, 128, *kernel) / (128 * 27) ** 0.5)
    y2 = _conv_ref(x2, w2, None, (1, 1, 1), causal)
    dy2 = bf16_round(torch.randn_like(y2))
    y2.backward(dy2)
    dyc = G.cl.to_cl(dy2.cuda())
    assert G.conv.narrow_dgrad_ok(spec2, dyc)
    x2c = G.cl.to_cl(x2.detach().cuda()).requires_grad_(True)
    op2 = GF.ConvOp(spec2)
    G.conv.PROFILER = prof = G.conv.LaunchProfiler()
    try:
        GF.conv3d(x2c, w2.cuda(), None, op2).backward(dyc)
    finally:
        G.conv.PROFILER = None
    assert 'conv_narrow_in_kernel' in prof.summary(), list(prof.summary())
    assert_close_bf16(x2c.grad, x2.grad, 'narrow conv dgrad')


WIDE_CASES = [
    # cin, cout, kernel, causal, size, residual epilogue
    (64, 256, (3, 3, 3), False, (2, 4, 8, 8), False),
    (128, 512, (3, 3, 3), True, (1, 3, 16, 16), True),
    (64, 320, (3, 3, 3), False, (1, 2, 32, 32), False),       # two column tiles, the second 64 wide
    (192, 256, (3, 1, 3), True, (2, 3, 4, 16), False),        # three channel blocks
    (128, 256, (3, 3, 3), False, (3, 1, 5, 8), True),         # M = 120: partial row tile
    (256, 256, (3, 3, 3), False, (2, 8, 16, 16), False),      # 16 row tiles, 4 channel blocks
]


@pytest.mark.parametrize('four_waves', [False, True])
@pytest.mark.parametrize('cin,cout,kernel,causal,size,resid', WIDE_CASES)
def test_conv_triple_wide_kernel(G, cin, cout, kernel, causal, size, resid, four_waves, monkeypatch):
    """igemm3w_kernel (256 x 256 tile, 32-channel weight half-tiles): forward (with and without the residual epilogue) and
    backward-data against the oracle; the library must report that the wide kernel ran (variant 12).  four_waves: the same tile as four
    waves of 128 x 128, one per SIMD (conv_igemm3x.hip, tri_flags bit 12)."""
    monkeypatch.setattr(G.conv, 'TRI_BM', 256)
    monkeypatch.setattr(G.conv, 'TRI_FLAGS', 1024 | 2048 | (4096 if four_waves else 0))
    torch.manual_seed(13)
    n, t, h, w = size
    x = bf16_round(torch.randn(n, cin, t, h, w))
    wt = bf16_round(torch.randn(cout, cin, *kernel) / (cin * kernel[0] * kernel[1] * kernel[2]) ** 0.5)
    b = torch.randn(cout)
    xr = x.clone().requires_grad_(True)
    if causal:
        from oracle import genie_oracle as O
        ref = O.causal_conv3d(xr, wt, b, stride=(1, 1, 1))
        spec = G.conv.causal_spec(cin, cout, kernel)
    else:
        ref = F.conv3d(xr, wt, b, padding=tuple((k - 1) // 2 for k in kernel))
        spec = G.conv.same_spec(cin, cout, kernel)
    dy = bf16_round(torch.randn_like(ref))
    ref.backward(dy)
    wd = wt.cuda()
    lib = G.hip.load_library()
    r = bf16_round(torch.randn_like(ref)) if resid else None
    out = G.conv.conv_forward(G.cl.to_cl(x.cuda()), G.conv.pack_weight_fwd(wd, spec), b.cuda(), spec, resid=None if r is None else G.cl.to_cl(r.cuda()))
    assert lib.genie_last_conv_variant() == 12, lib.genie_last_conv_variant()
    assert_close_bf16(out, ref.detach() + (r if resid else 0), 'wide triple fwd')
    if cin >= 256:                                            # backward-data = the same kernel over dy with cin output columns
        dx = G.conv.conv_dgrad(G.cl.to_cl(dy.cuda()), G.conv.pack_weight_bwd(wd, spec), spec, (t, h, w))
        assert lib.genie_last_conv_variant() == 12, lib.genie_last_conv_variant()
        assert_close_bf16(dx, xr.grad, 'wide triple dgrad')


K32_CASES = [
    # cin, cout, kernel, causal, size, residual epilogue
    (128, 128, (3, 3, 3), False, (2, 4, 16, 32), False),      # 16 row tiles, 4 sub-steps per (dt, dh)
    (128, 128, (3, 3, 3), True, (1, 3, 64, 64), True),        # W = 64: 4 image rows per tile; causal taps dt = -2..0; residual epilogue
    (64, 96, (3, 3, 3), False, (1, 2, 16, 16), False),        # one channel block, 96 columns (partial column tile)
    (192, 128, (3, 1, 3), True, (3, 3, 4, 16), False),        # M = 576: last tile partial; kh = 1
    (256, 64, (3, 3, 3), False, (2, 2, 8, 64), True),         # 8 sub-steps per (dt, dh), 64 columns
    (64, 128, (1, 3, 3), False, (5, 1, 16, 16), False),       # one frame per clip: every dt tap is zero padding of a clip border
]


@pytest.mark.parametrize('cin,cout,kernel,causal,size,resid', K32_CASES)
def test_conv_triple_k32_kernel(G, cin, cout, kernel, causal, size, resid, monkeypatch):
    """igemm3h_kernel (256 x 128 tile, 32-channel K-tiles, two blocks per CU, buffer-addressed staging: the Cout <= 128 layers): forward
    (with and without the residual epilogue) and backward-data against the oracle; the library must report variant 15."""
    monkeypatch.setattr(G.conv, 'TRI_BM', 256)
    monkeypatch.setattr(G.conv, 'TRI_FLAGS', 2048)
    torch.manual_seed(29)
    n, t, h, w = size
    x = bf16_round(torch.randn(n, cin, t, h, w))
    wt = bf16_round(torch.randn(cout, cin, *kernel) / (cin * kernel[0] * kernel[1] * kernel[2]) ** 0.5)
    b = torch.randn(cout)
    xr = x.clone().requires_grad_(True)
    if causal:
        from oracle import genie_oracle as O
        ref = O.causal_conv3d(xr, wt, b, stride=(1, 1, 1))
        spec = G.conv.causal_spec(cin, cout, kernel)
    else:
        ref = F.conv3d(xr, wt, b, padding=tuple((k - 1) // 2 for k in kernel))
        spec = G.conv.same_spec(cin, cout, kernel)
    dy = bf16_round(torch.randn_like(ref))
    ref.backward(dy)
    wd = wt.cuda()
    lib = G.hip.load_library()
    r = bf16_round(torch.randn_like(ref)) if resid else None
    out = G.conv.conv_forward(G.cl.to_cl(x.cuda()), G.conv.pack_weight_fwd(wd, spec), b.cuda(), spec, resid=None if r is None else G.cl.to_cl(r.cuda()))
    assert lib.genie_last_conv_variant() == 15, lib.genie_last_conv_variant()
    assert_close_bf16(out, ref.detach() + (r if resid else 0), 'k32 triple fwd')
    if cin <= 128 and cout % 64 == 0:                         # backward-data = the same kernel over dy with cin (<= 128) output columns
        dx = G.conv.conv_dgrad(G.cl.to_cl(dy.cuda()), G.conv.pack_weight_bwd(wd, spec), spec, (t, h, w))
        assert lib.genie_last_conv_variant() == 15, lib.genie_last_conv_variant()
        assert_close_bf16(dx, xr.grad, 'k32 triple dgrad')


@pytest.mark.parametrize('cout,causal,size', [(3, True, (2, 5, 10, 64)), (3, False, (1, 3, 7, 32)), (2, True, (1, 2, 4, 128)), (1, True, (2, 4, 9, 64)),
                                               (3, True, (8, 16, 64, 64)), (3, True, (1, 16, 64, 64)), (3, True, (32, 3, 8, 32)), (3, True, (1, 1, 3, 96)), (3, False, (1, 9, 6, 32))])
def test_conv_narrow_out_kernel(G, cout, causal, size):
    """conv_narrow.hip, 128 -> <= 3 channels (the head conv's forward): transposed 16x16x32 MFMA with the frame taps summed in the accumulator across
    input frames, against the oracle and against the generic kernel, over row strips / column blocks / frame segments / partial strips."""
    from genie import functional as GF
    torch.manual_seed(19)
    n, t, h, w = size
    kernel = (3, 3, 3)
    x = bf16_round(torch.randn(n, 128, t, h, w))
    wt = bf16_round(torch.randn(cout, 128, *kernel) / (128 * 27) ** 0.5)
    b = torch.randn(cout)
    spec = G.conv.causal_spec(128, cout, kernel) if causal else G.conv.same_spec(128, cout, kernel)
    xc = G.cl.to_cl(x.cuda())
    assert G.conv.narrow_out_ok(spec, xc)
    op = GF.ConvOp(spec)
    G.conv.PROFILER = prof = G.conv.LaunchProfiler()
    try:
        out = GF.conv3d(xc, wt.cuda(), b.cuda(), op)
    finally:
        G.conv.PROFILER = None
    assert 'conv_narrow_out_kernel' in prof.summary(), list(prof.summary())
    assert tuple(out.shape) == (n, cout, t, h, w) and G.cl.is_cl(out)
    if n * t * h * w <= 70000:
        ref = _conv_ref(x, wt, b, (1, 1, 1), causal)
        assert_close_bf16(out, ref, 'narrow-out conv fwd')
    gen = G.conv.conv_forward(xc, G.conv.pack_weight_fwd(wt.cuda(), spec), b.cuda(), spec)
    assert_close_bf16(out, gen.float().cpu(), 'narrow-out vs generic', rel=2 ** -7, rms_frac=3e-3)
    # pad channels of the 8-channel output pixels are zero (the CL invariant)
    raw = out.permute(0, 2, 3, 4, 1)
    base = torch.as_strided(raw, (n, t, h, w, 8), (t * h * w * 8, h * w * 8, w * 8, 8, 1))
    assert (base[..., cout:] == 0).all()


@pytest.mark.parametrize('cin,cout,size,force', [(128, 128, (2, 4, 16, 16), True), (128, 256, (1, 2, 16, 32), True), (256, 256, (4, 16, 32, 32), False)])
def test_residual_block_groupnorm_fused_into_conv_epilogues(G, cin, cout, size, force):
    """GroupNorm statistics from the producing conv's epilogue (GenieConvDesc.gn_sums) and the GroupNorm-backward reduce pass from the
    backward-data conv's epilogue (gnb_*): a chain of two residual blocks gives the same output, input gradient and parameter gradients
    as with the stand-alone passes, and the fused path really ran (it silently falls back when the kernel cannot do it)."""
    from genie import functional as GF
    from genie.module.video import VideoResidualBlock
    torch.manual_seed(23)
    n, t, h, w = size
    blocks = torch.nn.Sequential(VideoResidualBlock(cin, cout, num_groups=1), VideoResidualBlock(cout, cout, num_groups=1)).cuda()
    x0 = bf16_round(torch.randn(n, cin, t, h, w)).cuda()
    dy = bf16_round(torch.randn(n, cout, t, h, w)).cuda()
    res = []
    old_bm, old_fuse = G.conv.TRI_BM, G.conv.GN_FUSE
    try:
        if force:
            G.conv.TRI_BM = 256                          # small problem: force the 256-row kw-triple tile (the default would split K)
        for fuse in (0, 2):
            G.conv.GN_FUSE = fuse
            GF.GN_FUSE_COUNT['fwd'] = GF.GN_FUSE_COUNT['bwd'] = 0
            for p in blocks.parameters():
                p.grad = None
            x = G.cl.to_cl(x0.clone()).requires_grad_(True)
            out = blocks(x)
            out.backward(G.cl.to_cl(dy))
            torch.cuda.synchronize()
            res.append((out.detach().float().cpu(), x.grad.float().cpu(), [p.grad.float().cpu().clone() for p in blocks.parameters()],
                        dict(GF.GN_FUSE_COUNT)))
    finally:
        G.conv.TRI_BM, G.conv.GN_FUSE = old_bm, old_fuse
    (o0, g0, p0, c0), (o1, g1, p1, c1) = res
    assert c0 == {'fwd': 0, 'bwd': 0}
    assert c1['fwd'] == 3 and c1['bwd'] == 4, c1          # norm2 of both blocks + norm1 of the second; all four backward reduces
    # same arithmetic up to the summation order of the statistics; a statistic that moves by an fp32 ulp re-rounds a few bf16
    # intermediates, which two convs then spread: bound the error in norm and its worst element against the tensor RMS
    for a, b, what in ((o1, o0, 'output'), (g1, g0, 'input gradient')):
        rms = b.pow(2).mean().sqrt().item()
        assert (a - b).pow(2).mean().sqrt().item() < 2e-3 * rms, what
        assert (a - b).abs().max().item() < 4e-2 * rms, what
    for a, b, (name, _) in zip(p1, p0, blocks.named_parameters()):
        err = (a - b).norm().item() / (b.norm().item() + 1e-12)
        assert err < 5e-3, (name, err)


@pytest.mark.parametrize('n,d,v,dt', [(32768, 512, 1 << 18, torch.float32), (1000, 512, 8, torch.float32), (777, 64, 1000, torch.bfloat16), (5, 8, 3, torch.float32)])
def test_embedding_lookup_and_sparse_backward(G, n, d, v, dt):
    """genie_embedding_fwd / _bwd (nn.Embedding of DynamicsModel, dynamics.py:31-38) against F.embedding and a dense index_add_ on the CPU: the
    MaskGIT fill pattern (three quarters of the rows are token 0), ragged row counts, bf16 upstream gradients, accumulation into a non-zero
    buffer."""
    from genie import functional as GF
    torch.manual_seed(31)
    w = torch.randn(v, d)
    idx = torch.randint(0, v, (n,))
    idx[torch.rand(n) < 0.75] = 0
    wd = w.cuda().requires_grad_(True)
    out = GF.embedding(idx.cuda().view(1, n), wd)
    assert tuple(out.shape) == (1, n, d) and torch.equal(out.cpu()[0], F.embedding(idx, w))
    dy = torch.randn(1, n, d).to(dt)
    out.backward(dy.cuda().to(out.dtype) if dt == torch.float32 else dy.cuda().float())
    ref = torch.zeros(v, d).index_add_(0, idx, dy[0].float())
    torch.testing.assert_close(wd.grad.cpu(), ref, rtol=1e-5, atol=1e-5 * ref.abs().max().item())
    # the C entry point with a bf16 dy and a pre-filled gradient buffer
    g = torch.full((v, d), 0.25, device='cuda')
    dyb = dy[0].to(torch.bfloat16).cuda().contiguous()
    G.hip.check(G.hip.load_library().genie_embedding_bwd(idx.cuda().data_ptr(), dyb.data_ptr(), G.hip.GENIE_BF16, g.data_ptr(), n, d, v, G.hip.stream_ptr()), 'emb')
    ref_b = torch.full((v, d), 0.25).index_add_(0, idx, dyb.float().cpu())
    torch.testing.assert_close(g.cpu(), ref_b, rtol=1e-5, atol=1e-5 * ref_b.abs().max().item())


@pytest.mark.parametrize('m,k,n,xdt,odt,bias', [
    (64, 18, 512, torch.float32, torch.float32, True),          # AdaGN std / avg
    (5000, 512, 10, torch.bfloat16, torch.float32, True),       # LFQ proj_inp (R-yaml)
    (5000, 10, 512, torch.float32, torch.float32, True),        # LFQ proj_out
    (300, 8, 256, torch.float32, torch.bfloat16, False),        # cond to_k / to_v
    (64, 262144, 8, torch.bfloat16, torch.float32, False),      # LatentAction.to_act: 256 slices of the reduction axis
    (3, 40000, 8, torch.bfloat16, torch.float32, False),        # ragged slices, fewer rows than waves
    (777, 512, 18, torch.float32, torch.float32, False),        # out in (16, 32]: the 4-wide rowdot form (AdaGN's backward-data shape)
    (1, 5, 3, torch.float32, torch.float32, True),
])
def test_linear_small_forward_backward(G, m, k, n, xdt, odt, bias):
    """functional.linear (genie_linear_small_fwd / _wgrad, csrc/linear_small.hip) against F.linear in fp32 on the CPU: every call site's shape class,
    forward, backward-data (the same kernel with the weight's strides exchanged), weight / bias gradients accumulated into existing buffers, and
    run-to-run bit-reproducibility of the weight gradient (fixed-order partial sums, no atomics)."""
    from genie import functional as GF
    torch.manual_seed(7 + m + n)
    x = torch.randn(m, k)
    if xdt == torch.bfloat16:
        x = bf16_round(x)
    w = torch.randn(n, k) / k ** 0.5
    b = torch.randn(n) if bias else None
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    br = b.clone().requires_grad_(True) if bias else None
    ref = F.linear(xr, wr, br)
    dy = bf16_round(torch.randn(m, n))
    ref.backward(dy)
    xd = x.cuda().to(xdt).requires_grad_(True)
    wd = w.cuda().requires_grad_(True)
    bd = b.cuda().requires_grad_(True) if bias else None
    assert GF.linear_small_supported(k, n)
    y = GF.linear(xd, wd, bd, out_dtype=odt)
    assert y.dtype == odt and tuple(y.shape) == (m, n)
    tol = 1e-4 if odt == torch.float32 else 2 ** -7
    scale = ref.detach().abs().max().item() + 1e-6
    assert (y.float().cpu() - ref.detach()).abs().max().item() <= tol * scale, (y.float().cpu() - ref.detach()).abs().max().item() / scale
    y.backward(dy.cuda().to(y.dtype))
    gx_tol = 1e-4 if xdt == torch.float32 else 2 ** -7
    assert (xd.grad.float().cpu() - xr.grad).abs().max().item() <= gx_tol * (xr.grad.abs().max().item() + 1e-6)
    assert (wd.grad.cpu() - wr.grad).abs().max().item() <= 1e-4 * (wr.grad.abs().max().item() + 1e-6)
    if bias:
        assert (bd.grad.cpu() - br.grad).abs().max().item() <= 1e-4 * (br.grad.abs().max().item() + 1e-6)
    # bit-reproducible, and accumulating: a second backward doubles the gradient exactly
    g1 = wd.grad.clone()
    y2 = GF.linear(xd, wd, bd, out_dtype=odt)
    y2.backward(dy.cuda().to(y2.dtype))
    assert torch.equal(wd.grad, g1 + g1)


@pytest.mark.parametrize('wide,narrow,causal,size', [(256, 3, True, (2, 5, 10, 64)), (256, 4, False, (1, 3, 8, 32)), (384, 1, True, (1, 2, 4, 128))])
def test_conv_narrow_wide_side(G, wide, narrow, causal, size):
    """The narrow-conv kernels with a wide side of 256 / 384 channels (LatentAction.proj_in 3 -> 256 and the backward passes of proj_out 256 -> 3,
    action.py:60-70): one launch per 128-channel slab -- forward of narrow -> wide, backward-data of wide -> narrow, both weight gradients, through
    the module-level path against the oracle; the library must report the narrow kernels."""
    from genie import functional as GF
    from oracle import genie_oracle as O
    torch.manual_seed(23 + wide + narrow)
    n, t, h, w = size
    kernel = (3, 3, 3)
    ref_conv = (lambda x, wt, b: O.causal_conv3d(x, wt, b)) if causal else (lambda x, wt, b: F.conv3d(x, wt, b, padding=1))
    mk_spec = (lambda ci, co: G.conv.causal_spec(ci, co, kernel)) if causal else (lambda ci, co: G.conv.same_spec(ci, co, kernel))
    # ---- narrow -> wide: forward + weight gradient ----
    x = bf16_round(torch.randn(n, narrow, t, h, w))
    wt = bf16_round(torch.randn(wide, narrow, *kernel) / (narrow * 27) ** 0.5)
    b = torch.randn(wide)
    wr, br = wt.clone().requires_grad_(True), b.clone().requires_grad_(True)
    ref = ref_conv(x, wr, br)
    dy = bf16_round(torch.randn_like(ref))
    ref.backward(dy)
    spec = mk_spec(narrow, wide)
    xc = G.cl.to_cl(x.cuda())
    assert G.conv.narrow_fwd_ok(spec, xc)
    wd, bd = wt.cuda().requires_grad_(True), b.cuda().requires_grad_(True)
    G.conv.PROFILER = prof = G.conv.LaunchProfiler()
    try:
        out = GF.conv3d(xc, wd, bd, GF.ConvOp(spec))
        out.backward(G.cl.to_cl(dy.cuda()))
    finally:
        G.conv.PROFILER = None
    assert 'conv_narrow_in_kernel' in prof.summary() and 'conv_narrow_wgrad_kernel' in prof.summary(), list(prof.summary())
    assert_close_bf16(out, ref.detach(), 'wide stem fwd')
    torch.testing.assert_close(wd.grad.cpu(), wr.grad, rtol=1e-3, atol=1e-3 * wr.grad.abs().max().item())
    torch.testing.assert_close(bd.grad.cpu(), br.grad, rtol=1e-3, atol=1e-3 * br.grad.abs().max().item())
    # ---- wide -> narrow: backward-data + weight gradient (the forward of this direction stays on the generic kernel for wide > 128) ----
    x2 = bf16_round(torch.randn(n, wide, t, h, w))
    w2 = bf16_round(torch.randn(narrow, wide, *kernel) / (wide * 27) ** 0.5)
    b2 = torch.randn(narrow)
    x2r, w2r, b2r = x2.clone().requires_grad_(True), w2.clone().requires_grad_(True), b2.clone().requires_grad_(True)
    ref2 = ref_conv(x2r, w2r, b2r)
    dy2 = bf16_round(torch.randn_like(ref2))
    ref2.backward(dy2)
    spec2 = mk_spec(wide, narrow)
    x2c = G.cl.to_cl(x2.cuda()).requires_grad_(True)
    w2d, b2d = w2.cuda().requires_grad_(True), b2.cuda().requires_grad_(True)
    G.conv.PROFILER = prof = G.conv.LaunchProfiler()
    try:
        out2 = GF.conv3d(x2c, w2d, b2d, GF.ConvOp(spec2))
        out2.backward(G.cl.to_cl(dy2.cuda()))
    finally:
        G.conv.PROFILER = None
    assert 'conv_narrow_in_kernel' in prof.summary() and 'conv_narrow_wgrad_kernel' in prof.summary(), list(prof.summary())
    assert_close_bf16(out2, ref2.detach(), 'wide head fwd (generic kernel)')
    assert_close_bf16(x2c.grad, x2r.grad, 'wide head backward-data')
    torch.testing.assert_close(w2d.grad.cpu(), w2r.grad, rtol=1e-3, atol=1e-3 * w2r.grad.abs().max().item())
    torch.testing.assert_close(b2d.grad.cpu(), b2r.grad, rtol=1e-3, atol=1e-3 * b2r.grad.abs().max().item())
