"""Fused vocabulary head + masked cross-entropy (csrc/linear_ce.hip, genie_linear_ce_fwd / _bwd) against the fp32 arithmetic of the reference:
``logits = self.head(x)`` (genie/dynamics.py:62) and ``cross_entropy(logits[mask], tokens[mask])`` (dynamics.py:89-97), restated in
oracle/genie_oracle.py::linear_cross_entropy.  The HIP side never forms the logits; the oracle forms them in fp32 on the CPU.

Tolerances (inputs are bf16-representable on both sides, so the only differences are fp32 summation order, the bf16 rounding of the
softmax weights inside the two gradient products and of dh on the way out): loss 1e-4 relative, lse 2e-4 absolute + 1e-5 relative,
dh / dW / db 0.5 % of the tensor RMS."""
import pytest
import torch

from util import bf16_round, report

pytestmark = pytest.mark.gpu


def rel_rms(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return ((a - b).pow(2).mean().sqrt() / (b.pow(2).mean().sqrt() + 1e-30)).item()


def _case(m, d, v, bias=True, valid_frac=None, seed=0, scale=1.0, targets='random'):
    g = torch.Generator().manual_seed(seed)
    h = bf16_round(torch.randn(m, d, generator=g) * scale)
    w = bf16_round(torch.randn(v, d, generator=g) * (d ** -0.5))
    b = torch.randn(v, generator=g) * 0.1 if bias else None
    if targets == 'zero':                              # what compute_loss produces: every target equals the fill value (dynamics.py:84-93)
        t = torch.zeros(m, dtype=torch.int64)
    else:
        t = torch.randint(0, v, (m,), generator=g)
    valid = None if valid_frac is None else (torch.rand(m, generator=g) < valid_frac)
    return h, w, b, t, valid


def _run_hip(h, w, b, t, valid, grad_out=1.0):
    from genie import functional as GF
    hc = h.to(torch.bfloat16).cuda().requires_grad_(True)
    wc = w.cuda().requires_grad_(True)
    bc = None if b is None else b.cuda().requires_grad_(True)
    wpack = wc.detach().to(torch.bfloat16).contiguous()
    assert GF.linear_ce_supported(hc, wc)
    loss = GF.linear_cross_entropy(hc, wc, bc, wpack, t.cuda(), None if valid is None else valid.cuda())
    (loss * grad_out).backward()
    torch.cuda.synchronize()
    return loss.detach().cpu(), hc.grad.float().cpu(), wc.grad.cpu(), None if bc is None else bc.grad.cpu()


def _run_oracle(h, w, b, t, valid, grad_out=1.0):
    from oracle import genie_oracle as O
    hr, wr = h.clone().requires_grad_(True), w.clone().requires_grad_(True)
    br = None if b is None else b.clone().requires_grad_(True)
    loss = O.linear_cross_entropy(hr, wr, br, t, valid)
    (loss * grad_out).backward()
    return loss.detach(), hr.grad, wr.grad, None if br is None else br.grad


CASES = [
    # (rows, D, V, bias, valid fraction, targets, logit scale)     what the case is there for
    (200, 64, 1000, True, None, 'random', 1.),            # ragged everything: last row tile, last vocabulary tile, one split
    (513, 128, 4099, True, 0.7, 'random', 3.),            # rows switched off; V not a multiple of 8; peaked softmax
    (300, 256, 777, False, None, 'random', 2.),           # no bias
    (1000, 512, 5000, True, 0.5, 'zero', 4.),             # the shipped width; all-equal targets (the one-hot kernel's run merging)
    (77, 512, 50, True, None, 'random', 1.),              # V smaller than one tile pair
    (4100, 512, 40000, True, 0.9, 'random', 4.),          # several row tiles x several vocabulary splits
]


def _softmax_parts(h, w, b, t, valid, grad_out):
    """The two gradient products WITHOUT their one-hot terms, in float64: (softmax * g / count) W and (softmax * g / count)^T h.  With
    near-uniform logits the one-hot rows (W[target] / h[m], exact copies) are hundreds of times larger than these sums and would hide
    any error of the MFMA path in a relative RMS over the whole tensor."""
    on = torch.ones(h.shape[0], dtype=torch.bool) if valid is None else valid
    logits = h.double() @ w.double().t() + (0 if b is None else b.double())
    p = torch.softmax(logits, dim=1) * on[:, None] * (grad_out / on.sum())
    onehot_w = torch.zeros_like(w, dtype=torch.float64).index_add_(0, t[on], h.double()[on] * (grad_out / on.sum()))
    return p @ w.double(), p.t() @ h.double(), p.sum(0), onehot_w


@pytest.mark.parametrize('m,d,v,bias,vf,targets,scale', CASES)
def test_linear_ce_forward_backward(m, d, v, bias, vf, targets, scale):
    h, w, b, t, valid = _case(m, d, v, bias, vf, seed=m + v, targets=targets, scale=scale)
    lh, dh, dw, db = _run_hip(h, w, b, t, valid, grad_out=1.7)
    lo, dho, dwo, dbo = _run_oracle(h, w, b, t, valid, grad_out=1.7)
    soft_dh, soft_dw, soft_db, onehot_w = _softmax_parts(h, w, b, t, valid, 1.7)
    r = dict(loss=abs(lh.item() - lo.item()) / abs(lo.item()), dh=rel_rms(dh, dho), dw=rel_rms(dw, dwo), db=0. if db is None else rel_rms(db, dbo),
             dw_softmax_part=rel_rms(dw.double() + onehot_w, soft_dw), soft_share_of_dh=(soft_dh.pow(2).mean().sqrt() / dho.pow(2).mean().sqrt()).item())
    report('linear_ce', m=m, d=d, v=v, scale=scale, **r)
    assert r['loss'] < 1e-4, r
    assert r['dh'] < 5e-3 and r['dw'] < 5e-3 and r['db'] < 5e-3 and r['dw_softmax_part'] < 5e-3, r
    # the softmax-weighted sum alone (bf16 weights, fp32 accumulation): 0.5 % of ITS OWN RMS
    assert r['dw_softmax_part'] < 5e-3, r
    if valid is not None:
        assert (dh[~valid] == 0).all()                    # rows that are switched off carry exactly no gradient


def test_linear_ce_row_lse_and_large_logits():
    """lse per row against the fp32 oracle, with logits large enough (|h W^T| up to ~60) that the running maximum moves late in the sweep:
    row i's largest logit is planted at vocabulary row (i * 37) % V, so the deferred-maximum branch (rescale of O and l) fires at a
    different tile for different rows of one wave, also after many tiles."""
    from genie import functional as GF
    m, d, v = 256, 512, 9000
    h, w, b, t, _ = _case(m, d, v, seed=5)
    for i in range(m):
        w[(i * 37) % v] = bf16_round(h[i] * (60. / h[i].pow(2).sum()))        # logit of row i at that column ~ 60
    hc, wc, bc = h.to(torch.bfloat16).cuda().requires_grad_(True), w.cuda().requires_grad_(True), b.cuda()
    fn = GF._LinearCEFn
    loss = fn.apply(hc, wc, bc, wc.detach().to(torch.bfloat16), t.cuda(), None)
    loss.backward()
    logits = h.double() @ w.double().t() + b.double()
    lse = torch.logsumexp(logits, dim=1)
    ref = (lse - logits.gather(1, t[:, None])[:, 0]).mean()
    assert abs(loss.item() - ref.item()) < 1e-4 * abs(ref.item()), (loss.item(), ref.item())
    p = torch.softmax(logits, dim=1)
    p[torch.arange(m), t] -= 1
    dh_ref = (p @ w.double()) / m
    assert rel_rms(hc.grad.float().cpu(), dh_ref.float()) < 5e-3
    dw_ref = (p.t() @ h.double()) / m
    assert rel_rms(wc.grad.cpu(), dw_ref.float()) < 5e-3


def test_linear_ce_bad_target_poisons_the_loss_only_when_the_row_counts():
    """F.cross_entropy raises on a target outside [0, V); the kernel poisons the loss with NaN instead (same contract as
    genie_masked_ce_fwd) -- unless the row is switched off."""
    from genie import functional as GF
    h, w, b, t, _ = _case(130, 64, 300, seed=9)
    t[5] = 300
    hc, wc, bc = h.to(torch.bfloat16).cuda(), w.cuda(), b.cuda()
    wp = wc.to(torch.bfloat16)
    assert torch.isnan(GF.linear_cross_entropy(hc, wc, bc, wp, t.cuda(), None)).item()
    valid = torch.ones(130, dtype=torch.bool)
    valid[5] = False
    assert torch.isfinite(GF.linear_cross_entropy(hc, wc, bc, wp, t.cuda(), valid.cuda())).item()


def test_linear_ce_matches_the_materialising_path():
    """Same rows through the 1x1x1 gather-GEMM + genie_masked_ce (logits rounded to bf16 in HBM) and through the fused operator (fp32
    logits on chip): the two HIP paths agree to the bf16 rounding of the logits, and no-grad evaluation takes the lse-only sweep."""
    from genie import functional as GF
    from genie.dynamics import DynamicsModel
    torch.manual_seed(3)
    desc = (('space-time_attn', {'n_rep': 1, 'n_head': 2, 'd_head': 32}),)
    m = DynamicsModel(desc, tok_vocab=1000, act_vocab=5, embed_dim=64).cuda().train()
    tok, act = torch.randint(0, 1000, (2, 4, 8, 8)).cuda(), torch.randint(0, 5, (2, 4)).cuda()
    mask = (torch.rand(2, 4, 8, 8) < 0.75).cuda()
    fused = m.compute_loss(tok, act, mask=mask)
    fused.backward()
    g_f = {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}
    m.zero_grad(set_to_none=True)
    old, GF.FUSED_LINEAR_CE = GF.FUSED_LINEAR_CE, False
    try:
        plain = m.compute_loss(tok, act, mask=mask)
        plain.backward()
    finally:
        GF.FUSED_LINEAR_CE = old
    assert abs(fused.item() - plain.item()) < 2e-3 * abs(plain.item()), (fused.item(), plain.item())
    for n, p in m.named_parameters():
        if p.grad is not None and p.grad.abs().max() > 0:
            assert rel_rms(g_f[n], p.grad) < 3e-2, (n, rel_rms(g_f[n], p.grad))
    ws_train = GF._LinearCEFn.last_ws_floats
    with torch.no_grad():
        ev = m.compute_loss(tok, act, mask=mask)
    assert abs(ev.item() - fused.item()) < 1e-5 * abs(fused.item())
    # needs_input_grad mirrors requires_grad, not the grad mode: the evaluation pass must have asked for the lse-only workspace
    assert GF._LinearCEFn.last_ws_floats < ws_train, (GF._LinearCEFn.last_ws_floats, ws_train)


def test_linear_ce_frozen_weight_trainable_bias_and_deterministic_mode():
    """ A frozen head weight next to a trainable bias must not get a .grad; deterministic mode must route around the fused
    operator (its loss / dW reductions are fp32 atomics) and give bit-identical loss and head gradients run to run."""
    from genie import functional as GF, conv as GC
    from genie.dynamics import DynamicsModel
    torch.manual_seed(5)
    desc = (('space-time_attn', {'n_rep': 1, 'n_head': 2, 'd_head': 32}),)
    m = DynamicsModel(desc, tok_vocab=1000, act_vocab=5, embed_dim=64).cuda().train()
    tok, act = torch.randint(0, 1000, (2, 4, 8, 8)).cuda(), torch.randint(0, 5, (2, 4)).cuda()
    mask = (torch.rand(2, 4, 8, 8) < 0.75).cuda()
    m.head.weight.requires_grad_(False)
    m.compute_loss(tok, act, mask=mask).backward()
    assert m.head.weight.grad is None and m.head.bias.grad is not None and m.head.bias.grad.abs().max() > 0
    gb_frozen = m.head.bias.grad.clone()
    m.head.weight.requires_grad_(True)
    m.zero_grad(set_to_none=True)
    m.compute_loss(tok, act, mask=mask).backward()
    assert rel_rms(gb_frozen, m.head.bias.grad) < 1e-5
    m.zero_grad(set_to_none=True)
    old = GC.set_deterministic(True)
    try:
        xr = torch.zeros(128, 64, dtype=torch.bfloat16, device='cuda')
        assert not GF.linear_ce_supported(xr, m.head.weight)
        runs = []
        for _ in range(3):
            m.zero_grad(set_to_none=True)
            loss = m.compute_loss(tok, act, mask=mask)
            loss.backward()
            runs.append((loss.detach().clone(), {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}))
        for l, g in runs[1:]:
            assert torch.equal(l, runs[0][0])
            for n in g:
                # what deterministic mode promises (genie/conv.py): the loss and every conv / linear WEIGHT gradient (one K split per tile, a
                # single owner per element).  The LayerNorm gains / shifts and the embedding tables accumulate per-block partial sums with fp32
                # atomics in every mode (csrc/attention.hip rotary_ln_bwd_kernel, F.embedding's scatter): reproducible to ~1e-6, not bit for bit
                if n.startswith('head.') or '.ffn.' in n and n.endswith('.0.weight'):
                    assert torch.equal(g[n], runs[0][1][n]), n
                else:
                    assert rel_rms(g[n], runs[0][1][n]) < 1e-5, n
    finally:
        GC.set_deterministic(old)


@pytest.mark.parametrize('rows', [1024, 3000])
def test_linear_ce_full_vocabulary(rows):
    """BASELINE configs[3] head shape: D = 512, V = 2^18.  The oracle forms rows x 2^18 fp32 logits on the CPU (1 - 3 GB)."""
    m, d, v = rows, 512, 1 << 18
    h, w, b, t, valid = _case(m, d, v, True, 0.8, seed=21, targets='zero', scale=3.)
    lh, dh, dw, db = _run_hip(h, w, b, t, valid)
    lo, dho, dwo, dbo = _run_oracle(h, w, b, t, valid)
    on = valid
    onehot_w = torch.zeros_like(w).index_add_(0, t[on], h[on] / on.sum())
    r = dict(loss=abs(lh.item() - lo.item()) / abs(lo.item()), dh=rel_rms(dh, dho), dw=rel_rms(dw, dwo), db=rel_rms(db, dbo),
             dw_softmax_part=rel_rms(dw + onehot_w, dwo + onehot_w))
    report('linear_ce_full_vocabulary', m=m, d=d, v=v, loss_hip=lh.item(), loss_oracle=lo.item(), **r)
    assert r['loss'] < 1e-4, r
    assert r['dh'] < 5e-3 and r['dw'] < 5e-3 and r['db'] < 5e-3 and r['dw_softmax_part'] < 5e-3, r
