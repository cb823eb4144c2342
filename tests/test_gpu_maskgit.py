"""MaskGIT token ids on the HIP path (-m gpu): genie_maskgit_sample / genie_maskgit_paint and DynamicsModel.generate against the
oracle's restatement of reference genie/dynamics.py:101-165 (itself pinned to the real reference's generate() with injected noise,
tests/test_oracle_golden.py::test_dynamics_generate_token_ids).

Bit-exactness is defined at the operator boundary, as for the LFQ indices: given the SAME logits the sampled ids, the painted
positions and the resulting codes must equal the oracle's exactly.  End to end the HIP model's logits are bf16 results of bf16
activations while the oracle's are fp32, so a draw whose threshold u * total sits within the logits' noise of a CDF boundary can
legitimately differ; those tests require every differing draw to be PROVEN to sit that close to a boundary (EPS_CDF below), and
report the match rate.
"""
import os

import pytest
import torch

from util import bf16_round, report

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
EPS_ULP = 2e-6       # operator level: fp32 exp / sum-order noise between the device and torch-CPU softmax, relative to the row total
EPS_CDF = 0.1        # end to end: cap on max_j |cdf_oracle - cdf_hip| of any row (bf16 logits of bf16 activations vs the fp32 oracle; measured 0.07)


def _explain_draws(prob_ref, u, pred_ref, pred_hip, eps, prob_hip=None):
    """Every row whose draw differs must have its threshold within eps (relative to the row total) of the oracle CDF at one of the
    boundaries between the two ids.  With `prob_hip` (the probabilities the device path actually sampled from) the bound per row is
    the MEASURED deviation max_j |cdf_ref - cdf_hip| of that row -- a draw can only move if the threshold lies between the two
    CDFs -- and that deviation itself must stay below eps.  Returns the number of differing rows."""
    bad = (pred_ref != pred_hip).nonzero().flatten().tolist()
    cdf = prob_ref.double().cumsum(-1)
    cdf = cdf / cdf[:, -1:]
    dev = None
    if prob_hip is not None:
        ch = prob_hip.double().cumsum(-1)
        dev = (ch / ch[:, -1:] - cdf).abs().amax(-1)
        assert dev.max().item() < eps, f'CDF of the device logits deviates by {dev.max().item():.3g} from the oracle CDF (cap {eps})'
    for r in bad:
        lo, hi = sorted((int(pred_ref[r]), int(pred_hip[r])))
        gap = (cdf[r, lo:hi] - float(u[r])).abs().min().item()
        bound = eps if dev is None else dev[r].item() * (1 + 1e-6) + 1e-12
        assert gap <= bound, (f'row {r}: ids {int(pred_ref[r])} vs {int(pred_hip[r])} but the threshold is {gap:.3g} away from every CDF boundary '
                              f'between them (bound {bound:.3g})')
    # the probability that a draw moves = the measure of u between the two CDFs = their L1 distance
    return len(bad) if dev is None else (len(bad), (ch / ch[:, -1:] - cdf).abs().sum().item())


@pytest.mark.parametrize('rows,v,dtype,temp,spread', [
    (64, 256, torch.bfloat16, 1.0, 3.0), (37, 250, torch.float32, 1.0, 2.0), (50, 1000, torch.bfloat16, 0.7, 4.0), (16, 4096, torch.bfloat16, 1.3, 1.0),
    (8, 1 << 18, torch.bfloat16, 1.0, 3.0), (3, (1 << 18) - 5, torch.float32, 0.5, 2.0), (40, 16, torch.bfloat16, 1.0, 2.0), (5, 1, torch.float32, 1.0, 1.0),
    (33, 7, torch.float32, 2.0, 5.0),
])
def test_maskgit_sample_operator(rows, v, dtype, temp, spread):
    """Same logits in, same ids out: genie_maskgit_sample == softmax -> cumsum(double) -> count of the oracle (dynamics.py:138-143
    with injected uniforms), including u = 0, u -> 1, ragged V and -inf logits."""
    from genie import functional as GF
    from oracle import genie_oracle as O
    torch.manual_seed(rows * 7 + v % 1000)
    logits = (torch.randn(1, rows, v) * spread).to(dtype)
    if v > 4:
        logits[0, 0, : v // 2] = float('-inf')             # zero-probability prefix
    u = torch.rand(rows)
    u[0] = 0.
    if rows > 2:
        u[1] = 1. - 2 ** -24
        u[2] = 0.5
    pred, conf = GF.maskgit_sample(logits.cuda(), u, temp)
    torch.cuda.synchronize()
    lf = logits.float()[0]
    pred_ref, conf_ref = O.maskgit_sample_step(lf, u, temp)
    prob = torch.softmax(lf / temp, -1)
    assert pred.dtype == torch.int64 and tuple(pred.shape) == (1, rows)
    n_diff = _explain_draws(prob, u, pred_ref, pred.cpu()[0], EPS_ULP)
    report('maskgit_sample_operator', rows=rows, v=v, dtype=str(dtype).split('.')[-1], temp=temp, n_diff=n_diff)
    # "bit-exact up to proven CDF ties": a draw may differ only if its threshold sits within 2e-6 (relative) of a CDF boundary -- the fp32
    # exp / summation-order noise between the device's and torch-CPU's softmax -- which a uniform threshold does with probability
    # ~ 2 * 2e-6 * (boundaries near u) per row: at most ONE such row is tolerated per call (measured: 0 in every configuration)
    assert n_diff <= 1, f'{n_diff} of {rows} draws differ at the operator level'
    same = pred_ref == pred.cpu()[0]
    torch.testing.assert_close(conf.cpu()[0][same], conf_ref[same], rtol=2e-4, atol=1e-30)       # torch's CPU softmax sums 2^18 terms in fp32 lanes (measured 2.3e-5 off the exact sum)
    assert int(pred.min()) >= 0 and int(pred.max()) < v


def test_maskgit_sample_strided_last_frame():
    """Rows addressed in place inside a (B, T, h, w, Vp) logits tensor: the logits[:, -1] slice, channel pitch > V."""
    from genie import functional as GF
    from oracle import genie_oracle as O
    torch.manual_seed(3)
    b, t, h, w, v, vp = 3, 4, 5, 6, 100, 104
    full = torch.randn(b, t, h, w, vp).to(torch.bfloat16)
    u = torch.rand(b * h * w)
    pred, conf = GF.maskgit_sample(full.cuda()[:, -1, :, :, :v], u, 1.0)
    pred_ref, conf_ref = O.maskgit_sample_step(full[:, -1, :, :, :v].float(), u, 1.0)
    prob = torch.softmax(full[:, -1, :, :, :v].float(), -1).reshape(-1, v)
    assert _explain_draws(prob, u, pred_ref, pred.cpu().reshape(-1), EPS_ULP) <= 1
    # a dtype the kernel does not read (fp16) on the same padded-pitch view: the conversion re-packs it, pitches must follow
    half = full.to(torch.float16)
    pred16, _ = GF.maskgit_sample(half.cuda()[:, -1, :, :, :v], u, 1.0)
    pred16_ref, _ = O.maskgit_sample_step(half[:, -1, :, :, :v].float(), u, 1.0)
    prob16 = torch.softmax(half[:, -1, :, :, :v].float(), -1).reshape(-1, v)
    assert _explain_draws(prob16, u, pred16_ref, pred16.cpu().reshape(-1), EPS_ULP) <= 1


@pytest.mark.parametrize('b,n,ks', [(2, 16, (1, 5, 10)), (3, 256, (1, 6, 11, 17, 23, 28, 34, 40, 46, 50)), (1, 1000, (999, 1)), (2, 4096, (1000, 3000, 96)), (4, 1, (1,))])
def test_maskgit_paint_operator(b, n, ks):
    """genie_maskgit_paint == conf[~mask] = -inf; topk; gather; scatter_ of dynamics.py:146-158, step after step on the same state."""
    from genie import functional as GF
    from oracle import genie_oracle as O
    torch.manual_seed(n + b)
    mask_ref = torch.ones(b, n, dtype=torch.bool)
    code_ref = torch.zeros(b, n, dtype=torch.int64)
    mask = torch.ones(b, n, dtype=torch.uint8, device='cuda')
    code = torch.zeros(b, n, dtype=torch.int64, device='cuda')
    for k in ks:
        conf = torch.rand(b, n)                               # distinct with probability 1 (no ties: torch.topk's tie order is unspecified)
        pred = torch.randint(0, 1 << 18, (b, n))
        O.maskgit_paint_step(conf, pred, mask_ref, code_ref, k)
        GF.maskgit_paint(conf.cuda(), pred.cuda(), k, code, mask)
        assert torch.equal(code.cpu(), code_ref) and torch.equal(mask.cpu().bool(), mask_ref)
    assert int(mask.sum()) == n * b - sum(ks) * b


def _build(desc, tok_vocab, act_vocab, embed_dim, sd=None, seed=0, head_gain=2.):
    from genie.dynamics import DynamicsModel
    torch.manual_seed(seed)
    m = DynamicsModel(desc, tok_vocab=tok_vocab, act_vocab=act_vocab, embed_dim=embed_dim)
    if sd is not None:
        m.load_state_dict(sd)
    else:
        with torch.no_grad():
            for n_, p in m.named_parameters():
                if 'freq' not in n_ and p.dim() >= 2:
                    p.copy_(bf16_round(p * (head_gain if 'head' in n_ else 1.)))
                elif n_.endswith('attn.norm.weight'):
                    p.fill_(0.45)        # (round 6) spread softmax: with gamma = 1 the self-score makes q = k = v attention the identity and masks go untested
        sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    return m.cuda().eval(), sd


def _check_generate(m, sd, desc, tok, act, u, steps, which, temp, gen_ref=None):
    """(1) operator level on the REAL logits: replaying the oracle's sampler on the logits the HIP model produced reproduces the
    HIP ids bit for bit;  (2) end to end against the fp32 oracle: every differing draw / pick is explained by the logits' noise."""
    from genie import functional as GF
    from oracle import genie_oracle as O
    b, t, h, w = tok.shape
    n = h * w
    trace = []
    gen = m.generate(tok.cuda(), act.cuda(), steps=steps, which=which, temp=temp, uniforms=u, trace=trace).cpu()
    assert tuple(gen.shape) == (b, t + 1, h, w) and gen.dtype == tok.dtype and torch.equal(gen[:, :t], tok)

    # (1) oracle sampler replayed on the HIP logits
    mask = torch.ones(b, n, dtype=torch.bool)
    code = torch.zeros(b, n, dtype=torch.int64)
    explained = 0
    for step, tr in enumerate(trace):
        lg = tr['logits'].float().cpu().reshape(b * n, -1)
        pred_o, conf_o = O.maskgit_sample_step(lg, u[step], temp)
        explained += _explain_draws(torch.softmax(lg / temp, -1), u[step], pred_o, tr['pred'].cpu().reshape(-1), EPS_ULP)
        assert torch.equal(tr['mask_before'].cpu().bool(), mask), f'step {step}: mask state diverged'
        cg = tr['conf'].cpu().reshape(b, n).masked_fill(~mask, -1.)
        if tr['k'] < int(mask.sum(-1).min()):      # (with more candidates than picks) torch.topk's order among equal confidences is unspecified; the kernel's is (lower index first)
            top = cg.topk(tr['k'] + 1, -1).values
            assert (top[:, -2] > top[:, -1]).all(), f'step {step}: tied confidences at the top-k boundary (test input too peaked)'
        O.maskgit_paint_step(tr['conf'].cpu().reshape(b, n), tr['pred'].cpu().reshape(b, n), mask, code, tr['k'])
    assert explained <= 2, f'{explained} draws differ between the device sampler and the oracle sampler on identical logits'
    assert torch.equal(gen[:, -1].reshape(b, n), code), 'painted codes differ from the oracle sampler replayed on the same logits'
    assert not mask.any()

    # (2) end to end vs the fp32 oracle (or the committed output of the real reference)
    tr_ref = []
    gen_o = O.dynamics_generate(tok, act, sd, desc, u, steps=steps, which=which, temp=temp, trace=tr_ref)
    if gen_ref is not None:
        assert torch.equal(gen_o, gen_ref)
    n_draw = n_diff = 0
    budget = 0.
    for step, (tg, to) in enumerate(zip(trace, tr_ref)):
        prob_o = torch.softmax(to['logits'].reshape(b * n, -1) / temp, -1)
        prob_g = torch.softmax(tg['logits'].float().cpu().reshape(b * n, -1) / temp, -1)
        pred_g, pred_r = tg['pred'].cpu().reshape(-1), to['pred'].reshape(-1)
        nd, dsum = _explain_draws(prob_o, u[step], pred_r, pred_g, EPS_CDF, prob_hip=prob_g)
        n_diff += nd
        budget += dsum          # expected number of moved draws: sum over rows of the L1 distance between the two CDFs
        n_draw += b * n
        # teacher-forced pick: painting the oracle's state with the HIP confidences must choose the oracle's positions, except where
        # the k-th and (k+1)-th confidences are closer than the logits' noise (samples with a differing draw are skipped: a different
        # id carries a different confidence)
        conf_g = tg['conf'].cpu().reshape(b, n).masked_fill(~to['mask_before'], -1.)
        conf_r = to['conf'].reshape(b, n).masked_fill(~to['mask_before'], -1.)
        k = to['k']
        pick_g, pick_r = conf_g.topk(k, -1).indices, conf_r.topk(k, -1).indices
        same_draws = (pred_g == pred_r).reshape(b, n).all(-1)
        for bi in range(b):
            sg, sr = set(pick_g[bi].tolist()), set(pick_r[bi].tolist())
            if sg != sr and bool(same_draws[bi]):
                kth = conf_r[bi].topk(min(k + 1, n)).values[-2:].mean().item()
                for pos in sg ^ sr:
                    assert abs(conf_r[bi, pos].item() - kth) < EPS_CDF * 2, f'step {step}: position {pos} picked differently with margin {abs(conf_r[bi, pos].item() - kth):.3g}'
    match = (gen == gen_o).float().mean().item()
    print(f'MaskGIT end-to-end: {n_diff}/{n_draw} draws differ from the fp32 oracle (each explained by the measured CDF deviation of its row, cap {EPS_CDF}); final id match rate {match:.4f}; expected number of moved draws <= {budget:.1f}')
    report('maskgit_generate_vs_oracle', shape=list(tok.shape), steps=steps, which=which, temp=temp, vocab=int(trace[0]['logits'].shape[-1]), draws=n_draw,
           draws_moved=n_diff, expected_moved=budget, final_id_match_rate=match, sampler_bit_exact_on_same_logits=True)
    assert n_diff <= 1.5 * budget + 10, (n_diff, budget)
    assert match > 0.75
    return match


@pytest.mark.parametrize('seed,shape,steps,which,temp,cfg', [
    (0, (2, 5, 4, 4), 6, 'linear', 1.0, (2, 2, 32, 256, 5, 64)),
    (1, (3, 3, 8, 8), 10, 'cosine', 0.8, (2, 4, 16, 64, 4, 64)),
    (2, (2, 4, 6, 10), 7, 'arccos', 1.2, (1, 2, 64, 1000, 8, 128)),
    (3, (2, 2, 16, 16), 5, 'linear', 1.0, (2, 4, 8, 512, 3, 32)),
])
def test_generate_token_ids_vs_oracle(seed, shape, steps, which, temp, cfg):
    n_rep, n_head, d_head, vocab, n_act, dim = cfg
    desc = (('space-time_attn', {'n_rep': n_rep, 'n_head': n_head, 'd_head': d_head}),)
    m, sd = _build(desc, vocab, n_act, dim, seed=seed)
    torch.manual_seed(100 + seed)
    b, t, h, w = shape
    tok, act = torch.randint(0, vocab, shape), torch.randint(0, n_act, (b, t))
    u = torch.rand(steps, b * h * w)
    _check_generate(m, sd, desc, tok, act, u, steps, which, temp)


def test_generate_vs_reference_golden():
    """The HIP generate() against the committed output of the REAL reference's generate() on its own test configuration
    (test/test_dynamics.py: 4 x ST(4 x 16), 16 tokens, (2, 10, 16, 16) context; tests/golden/make_golden_generate.py)."""
    g = torch.load(os.path.join(GOLD, 'dynamics_generate.pt'), weights_only=False)
    for name, e in g.items():
        m, sd = _build(e['desc'], e['tok_vocab'], e['act_vocab'], e['embed_dim'], sd=e['sd'])
        _check_generate(m, sd, e['desc'], e['tokens'], e['act'], e['uniforms'], e['steps'], e['which'], e['temp'], gen_ref=e['gen'])


def test_generate_device_rng_and_feedback():
    """Without injected noise the uniforms come from the device RNG (same distribution as the reference's torch.multinomial);
    feedback=True (opt-in repair of dynamics.py:128,136) re-runs the model on the painted frame each step."""
    desc = (('space-time_attn', {'n_rep': 1, 'n_head': 2, 'd_head': 32}),)
    m, sd = _build(desc, 64, 4, 64, seed=5)
    tok, act = torch.randint(0, 64, (2, 3, 4, 4)).cuda(), torch.randint(0, 4, (2, 3)).cuda()
    torch.manual_seed(0)
    g1 = m.generate(tok, act, steps=4)
    torch.manual_seed(0)
    g2 = m.generate(tok, act, steps=4)
    assert torch.equal(g1, g2) and tuple(g1.shape) == (2, 4, 4, 4) and int(g1.max()) < 64
    g3 = m.generate(tok, act, steps=4, feedback=True, uniforms=torch.rand(4, 32))
    assert tuple(g3.shape) == (2, 4, 4, 4) and torch.equal(g3[:, :3], tok)
