"""Attention dropout (reference genie/module/attention.py:225-230: `dropout_p=self.dropout` handed to scaled_dot_product_attention) on the HIP
path (-m gpu).  torch's Philox stream is not reproducible outside torch (and differs between its own backends), so parity is THROUGH THE MASK: the
kernels' keep decisions are a pure function of (seed, sequence, head, query, key); `genie_attention_dropout_mask` writes them out and the fp32
reference / the oracle apply exactly those draws -- `softmax(S) o M / (1 - p) @ V`, sdpa's documented form."""
import pytest
import torch

from util import assert_close_bf16, bf16_round, report

pytestmark = pytest.mark.gpu


def rel_rms(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return ((a - b).pow(2).mean().sqrt() / (b.pow(2).mean().sqrt() + 1e-20)).item()


def export_mask(nseq, nhead, sq, sk, p, seed):
    from genie import _hip
    keep = torch.full((nseq, nhead, sq, sk), 7, dtype=torch.uint8, device='cuda')
    _hip.check(_hip.load_library().genie_attention_dropout_mask(_hip.ptr(keep), nseq, nhead, sq, sk, p, seed, _hip.stream_ptr()), 'mask')
    torch.cuda.synchronize()
    assert int(keep.max()) <= 1
    return keep.cpu()


@pytest.mark.parametrize('p', [0.1, 0.5, 0.9])
def test_dropout_mask_is_bernoulli_and_keyed(p):
    """The exported decisions: keep rate 1 - p overall (5 sigma), per (sequence, head) plane, per query row and per key column (no stripes); the planes of
    different sequences / heads / seeds are independent draws (agreement rate p^2 + (1 - p)^2, 5 sigma); the same seed gives the same mask."""
    nseq, nhead, sq, sk = 3, 4, 96, 160
    m = export_mask(nseq, nhead, sq, sk, p, 1234).float()
    n = m.numel()
    sig = lambda cnt: 5.0 * (p * (1 - p) / cnt) ** 0.5
    assert abs(m.mean().item() - (1 - p)) < sig(n)
    assert (m.mean((2, 3)) - (1 - p)).abs().max().item() < sig(sq * sk) * 1.2
    assert (m.mean((0, 1, 3)) - (1 - p)).abs().max().item() < sig(nseq * nhead * sk) * 1.3           # per query row
    assert (m.mean((0, 1, 2)) - (1 - p)).abs().max().item() < sig(nseq * nhead * sq) * 1.3           # per key column
    agree = p * p + (1 - p) * (1 - p)
    sa = 5.0 * (agree * (1 - agree) / (sq * sk)) ** 0.5
    planes = m.reshape(nseq * nhead, -1)
    for i in range(planes.shape[0] - 1):
        assert abs((planes[i] == planes[i + 1]).float().mean().item() - agree) < sa, i
    # neighbouring elements along both axes: no lag-1 structure
    assert abs((m[..., 1:] == m[..., :-1]).float().mean().item() - agree) < 5.0 * (agree * (1 - agree) / n) ** 0.5 * 1.5
    assert abs((m[..., 1:, :] == m[..., :-1, :]).float().mean().item() - agree) < 5.0 * (agree * (1 - agree) / n) ** 0.5 * 1.5
    m2 = export_mask(nseq, nhead, sq, sk, p, 1235).float()
    assert abs((m == m2).float().mean().item() - agree) < 5.0 * (agree * (1 - agree) / n) ** 0.5
    assert torch.equal(m, export_mask(nseq, nhead, sq, sk, p, 1234).float())
    report('attention_dropout_mask', p=p, keep_rate=m.mean().item())


@pytest.mark.parametrize('nseq,nhead,dh,sq,sk,causal,self_attn,p', [
    (3, 2, 64, 200, 200, False, True, 0.25),          # spatial self-attention, ragged multi-tile (would be a lean-kernel shape without dropout)
    (2, 4, 64, 320, 320, False, True, 0.1),
    (24, 2, 64, 16, 16, True, True, 0.3),             # temporal (the packed kernels' shape without dropout)
    (3, 2, 32, 70, 70, True, True, 0.5),
    (2, 1, 128, 130, 130, False, True, 0.2),
    (3, 2, 64, 96, 40, False, False, 0.25),           # separate K / V rows, Sq != Sk
    (2, 2, 32, 33, 150, False, False, 0.4),
    (3, 4, 16, 50, 50, True, True, 0.3),              # narrow heads: the fp32 kernels of attention_narrow.hip (the reference's own small blueprints are 4 x 16)
    (2, 2, 8, 40, 21, False, False, 0.25),
    (5, 4, 16, 10, 10, True, True, 0.5),
])
def test_attention_dropout_through_the_c_abi(nseq, nhead, dh, sq, sk, causal, self_attn, p):
    """genie_attention_fwd_dropout / genie_attention_bwd_dropout against fp32 softmax attention with the exported mask applied, and autograd of it."""
    from genie import _hip
    lib = _hip.load_library()
    P = _hip.ptr
    torch.manual_seed(7 * sq + sk + dh)
    c = nhead * dh
    scale = dh ** -0.5
    seed = 0x1234_5678_9ABC + sq
    u = bf16_round(torch.randn(nseq, sq, c) * 0.8)
    kx = u if self_attn else bf16_round(torch.randn(nseq, sk, c) * 0.8)
    vx = u if self_attn else bf16_round(torch.randn(nseq, sk, c))
    do = bf16_round(torch.randn(nseq, sq, c) * 0.5)
    keep = export_mask(nseq, nhead, sq, sk, p, seed)
    ur = u.clone().requires_grad_(True)
    kr = ur if self_attn else kx.clone().requires_grad_(True)
    vr = ur if self_attn else vx.clone().requires_grad_(True)
    heads = lambda t: t.reshape(t.shape[0], t.shape[1], nhead, dh).transpose(1, 2)
    sc = (heads(ur) @ heads(kr).transpose(-1, -2)) * scale
    if causal:
        sc = sc.masked_fill(~torch.ones(sq, sk, dtype=torch.bool).tril(), float('-inf'))
    w = sc.softmax(-1) * keep.float() / (1 - p)
    o_ref = (w @ heads(vr)).transpose(1, 2).reshape(nseq, sq, c)
    o_ref.backward(do)
    lse_ref = sc.logsumexp(-1).detach()                                                # (nseq, nhead, sq): the softmax's, no dropout in it

    dev = lambda t: t.cuda().to(torch.bfloat16).contiguous()
    ud, dod = dev(u), dev(do)
    kd, vd = (ud, ud) if self_attn else (dev(kx), dev(vx))
    qmap, kvmap = _hip.i64((1, sq * c, 0, c)), _hip.i64((1, sk * c, 0, c))
    ntok = nseq * sq
    out, oattn = torch.full_like(ud, 9.0), torch.full_like(ud, 9.0)
    lse = torch.full((ntok * nhead,), 9.0, device='cuda')
    resid = dev(bf16_round(torch.randn(nseq, sq, c)))
    _hip.check(lib.genie_attention_fwd_dropout(P(ud), P(kd), P(vd), P(resid), P(out), P(oattn), P(lse), nseq, nhead, dh, sq, sk, qmap, kvmap, qmap, scale,
                                               int(causal), c, p, seed, _hip.stream_ptr()), 'fwd')
    torch.cuda.synchronize()
    assert_close_bf16(oattn.float().cpu(), o_ref, 'dropout fwd', rms_frac=1e-2)
    assert rel_rms(out.float() - resid.float(), o_ref) < 2e-2                          # out = o_attn + resid (one bf16 store)
    torch.testing.assert_close(lse.cpu().reshape(nseq, sq, nhead).permute(0, 2, 1), lse_ref, rtol=2e-3, atol=2e-3)
    # p -> the plain call when dropout_p == 0 (bit-identical to genie_attention_fwd)
    o0, o1 = torch.empty_like(ud), torch.empty_like(ud)
    _hip.check(lib.genie_attention_fwd_dropout(P(ud), P(kd), P(vd), None, P(o0), None, None, nseq, nhead, dh, sq, sk, qmap, kvmap, qmap, scale,
                                               int(causal), c, 0.0, seed, _hip.stream_ptr()), 'fwd p=0')
    _hip.check(lib.genie_attention_fwd(P(ud), P(kd), P(vd), None, P(o1), None, None, nseq, nhead, dh, sq, sk, qmap, kvmap, qmap, scale,
                                       int(causal), c, _hip.stream_ptr()), 'fwd plain')
    assert torch.equal(o0, o1)

    D = torch.empty(3 * ntok * nhead, device='cuda')
    dq = torch.full_like(ud, 9.0)
    dk = dv = None
    if not self_attn:
        dk, dv = torch.full_like(kd, 9.0), torch.full_like(kd, 9.0)
    _hip.check(lib.genie_attention_bwd_dropout(P(ud), P(kd), P(vd), P(oattn), None, P(dod), P(lse), P(D), P(dq), P(dk), P(dv), nseq, nhead, dh, sq, sk,
                                               qmap, kvmap, qmap, None, scale, int(causal), c, ntok, p, seed, _hip.stream_ptr()), 'bwd')
    torch.cuda.synchronize()
    errs = {'dq': rel_rms(dq, ur.grad)}                                                # self-attention: du = dQ + dK + dV, autograd's sum as well
    if not self_attn:
        errs['dk'], errs['dv'] = rel_rms(dk, kr.grad), rel_rms(dv, vr.grad)
    assert max(errs.values()) < 2e-2, errs
    # a different seed is a different function: the same call must NOT reproduce the reference any more (guards against a mask that is ignored)
    o2 = torch.empty_like(ud)
    _hip.check(lib.genie_attention_fwd_dropout(P(ud), P(kd), P(vd), None, P(o2), None, None, nseq, nhead, dh, sq, sk, qmap, kvmap, qmap, scale,
                                               int(causal), c, p, seed + 1, _hip.stream_ptr()), 'fwd other seed')
    assert rel_rms(o2, o_ref) > 0.1
    report('attention_dropout_cabi', S=(sq, sk), d_head=dh, p=p, causal=causal, self_attn=self_attn, fwd=rel_rms(oattn, o_ref), **errs)


def test_dropout_refuses_what_it_cannot_do():
    from genie import _hip
    lib = _hip.load_library()
    P = _hip.ptr
    u = torch.zeros(1, 8, 32, dtype=torch.bfloat16, device='cuda')
    o = torch.empty_like(u)
    m = _hip.i64((1, 8 * 32, 0, 32))
    assert lib.genie_attention_fwd_dropout(P(u), P(u), P(u), None, P(o), None, None, 1, 1, 32, 8, 8, m, m, m, 0.25, 0, 32, 1.0, 1, _hip.stream_ptr()) != 0   # p = 1
    assert lib.genie_attention_fwd_dropout(P(u), P(u), P(u), None, P(o), None, None, 1, 1, 32, 8, 8, m, m, m, 0.25, 0, 32, -0.1, 1, _hip.stream_ptr()) != 0


@pytest.mark.parametrize('kind,n_head,d_head,thw,p', [('space', 2, 64, (3, 12, 12), 0.2), ('time', 4, 32, (12, 4, 5), 0.3), ('space', 2, 32, (2, 6, 7), 0.5),
                                                      ('space', 4, 16, (2, 8, 8), 0.3), ('time', 4, 16, (9, 3, 4), 0.2)])
def test_attention_module_with_dropout_matches_oracle(kind, n_head, d_head, thw, p):
    """SpatialAttention / TemporalAttention(dropout=p) against the oracle's spatial_attention / temporal_attention with the module's own mask (rebuilt from
    `last_dropout_seed`): output, input gradient, LayerNorm gradients.  Eval mode drops as well (the reference hands `dropout_p` to the FUNCTIONAL sdpa)."""
    from oracle import genie_oracle as O
    from genie.module.attention import SpatialAttention, TemporalAttention
    torch.manual_seed(11)
    cls, ofn = (SpatialAttention, O.spatial_attention) if kind == 'space' else (TemporalAttention, O.temporal_attention)
    # (TemporalAttention is causal only where SpaceTimeAttention builds it, attention.py:409-424; the oracle's temporal_attention IS that use)
    m = cls(n_head=n_head, d_head=d_head, transpose=True, dropout=p, **({'causal': True} if kind == 'time' else {}))
    with torch.no_grad():
        # a small LayerNorm gain: with gamma ~ 1 the self-score |u|^2 * scale (~ 20) dwarfs the cross scores and softmax is the identity -- a test
        # through which a wrong mask (or a missing causal flag: the first version of this test) passes; here the weights are spread over the keys
        m.norm.weight.copy_(torch.randn_like(m.norm.weight) * 0.1 + 0.45)
        m.norm.bias.copy_(torch.randn_like(m.norm.bias) * 0.2)
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    m = m.cuda()
    c = n_head * d_head
    t, h, w = thw
    x = bf16_round(torch.randn(2, c, t, h, w))
    xc = x.cuda().requires_grad_(True)
    torch.manual_seed(5)
    out = m(xc)
    seed = m.last_dropout_seed
    assert seed is not None
    nseq, s = (2 * t, h * w) if kind == 'space' else (2 * h * w, t)
    keep = export_mask(nseq, n_head, s, s, p, seed)
    sd_req = {k: (v.clone().requires_grad_(True) if 'freq' not in k else v) for k, v in sd.items()}
    xr = x.clone().requires_grad_(True)
    ref = ofn(xr, sd_req, '', n_head, d_head, True, drop_keep=keep, drop_p=p)
    assert rel_rms(out, ref) < 1.5e-2, rel_rms(out, ref)
    dy = bf16_round(torch.randn_like(ref))
    ref.backward(dy)
    out.backward(dy.cuda())
    assert rel_rms(xc.grad, xr.grad) < 5e-2, rel_rms(xc.grad, xr.grad)
    for name in ('norm.weight', 'norm.bias'):
        assert rel_rms(dict(m.named_parameters())[name].grad, sd_req[name].grad) < 6e-2, name
    # the same torch seed reproduces the run; another one draws another mask; eval mode still drops (QUIRK)
    torch.manual_seed(5)
    again = m(x.cuda())
    assert m.last_dropout_seed == seed and torch.equal(again, out.detach())
    other = m(x.cuda())
    assert m.last_dropout_seed != seed and rel_rms(other, out) > 0.05
    plain = cls(n_head=n_head, d_head=d_head, transpose=True, **({'causal': True} if kind == 'time' else {}))
    plain.load_state_dict(sd)
    plain = plain.cuda()
    m.eval()
    assert rel_rms(m(x.cuda()), plain(x.cuda())) > 0.05
    report('attention_dropout_module', kind=kind, d_head=d_head, p=p, fwd=rel_rms(out, ref), dx=rel_rms(xc.grad, xr.grad))


def test_space_time_block_with_dropout_trains_and_refuses_capture(monkeypatch):
    """SpaceTimeAttention(dropout=...) hands the rate to both attentions (attention.py:409-424): a step runs, gradients are finite, and a hipGraph capture of
    it is refused (the seed is a launch argument: a replay would reuse one mask)."""
    from genie.module.attention import SpaceTimeAttention
    torch.manual_seed(0)
    m = SpaceTimeAttention(n_head=2, d_head=32, transpose=True, dropout=0.1).cuda()
    assert m.space_attn.dropout == 0.1 and m.temp_attn.dropout == 0.1
    x = torch.randn(2, 64, 4, 6, 6, device='cuda', requires_grad=True)
    y = m(x)
    y.float().pow(2).mean().backward()
    assert torch.isfinite(x.grad).all() and all(torch.isfinite(p.grad).all() for n, p in m.named_parameters() if p.grad is not None)
    monkeypatch.setattr(torch.cuda, 'is_current_stream_capturing', lambda: True)        # (no real capture: an exception inside one poisons the stream)
    with pytest.raises(RuntimeError, match='hipGraph'):
        m(x.detach())
