"""oracle/genie_oracle.py against the committed outputs of the REAL reference (tests/golden/*.pt, produced by
tests/golden/make_golden.py in the build container).  Runs everywhere (CPU); this is what pins the oracle."""
import os

import pytest
import torch

from oracle import genie_oracle as O

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def load(name):
    return torch.load(os.path.join(GOLD, name), weights_only=False)


def close(a, b, tol=2e-5):
    assert a.shape == b.shape, (a.shape, b.shape)
    err = (a - b).abs().max().item()
    assert err <= tol * max(1.0, b.abs().max().item()), f'max err {err}'


def test_ops():
    g = load('ops.pt')
    for i in range(4):
        e = g[f'causal_conv3d_{i}']
        close(O.causal_conv3d(e['x'], e['weight'], e['bias'], stride=e['stride']), e['out'])
    for i in range(4):
        e = g[f'video_residual_{i}']
        close(O.video_residual_block(e['x'], e['sd'], '', **e['kw']), e['out'])
    e = g['depth2spacetime']
    close(O.depth2spacetime_upsample(e['x'], e['sd'], '', time_factor=2, space_factor=2), e['out'])
    e = g['adagn']
    close(O.adaptive_group_norm(e['x'], e['cond'], e['sd'], '', 4), e['out'])
    e = g['groupnorm_silu']
    close(O.silu(O.group_norm(e['x'], 8, e['sd']['weight'], e['sd']['bias'])), e['out'])


def test_lfq():
    g = load('lfq.pt')
    for name, e in g.items():
        if 'eval_idx' in e:
            (o, i), l = O.lfq_forward(e['x'], e['sd'], '', e['d'], e['n'], training=False, transpose=True)
            assert l is None
            assert torch.equal(i, e['eval_idx']) and i.dtype == e['eval_idx'].dtype
            close(o, e['eval_out'])
        if 'train_loss' in e:
            (o, i), l = O.lfq_forward(e['x'], e['sd'], '', e['d'], e['n'], training=True, transpose=True)
            close(l, e['train_loss'], 1e-5)
            if 'train_idx' in e:
                assert torch.equal(i, e['train_idx'])
    # the factorised entropy (what the HIP kernel computes) against the reference's d=18 loss
    e = g['d18_train']
    z = e['x'].movedim(1, -1).reshape(-1, 18)
    inp_ent, avg_ent = O.lfq_entropy_terms_factored(z, 100.)
    commit = (z - z.sign()).pow(2).mean()
    loss = (inp_ent + avg_ent) * 0.1 + commit * 0.25
    assert abs(loss.item() - e['train_loss'].item()) < 1e-5


def test_tokenizer_small():
    g = load('tokenizer_small.pt')
    sd, x = g['sd'], g['x']
    close(O.tokenizer_encode(x, sd, g['enc_desc']), g['enc'], 1e-4)
    q, idx = O.tokenizer_tokenize(x, sd, g['enc_desc'], g['d_codebook'])
    assert torch.equal(idx, g['idx'])
    close(q, g['quant'])
    close(O.tokenizer_decode(g['quant'], sd, g['dec_desc']), g['rec'], 1e-4)
    loss, (rec_loss, q_loss), _, _ = O.tokenizer_forward_hotpath(x, sd, g['enc_desc'], g['dec_desc'], g['d_codebook'])
    close(loss, g['rfwd_loss'], 1e-4)
    close(q_loss, g['quant_loss'], 1e-4)


def test_st_block():
    g = load('st_block.pt')
    for name, e in g.items():
        cond = (None, e['cond']) if e['cond'] is not None else None
        close(O.space_time_block(e['x'], e['sd'], '', 4, 16, transpose=e['transpose'], cond=cond), e['out'], 1e-4)


def test_dynamics():
    g = load('dynamics_small.pt')
    logits, last = O.dynamics_forward(g['tokens'], g['act'], g['sd'], g['desc'])
    close(logits, g['logits'], 1e-4)
    close(O.dynamics_loss(g['tokens'], g['act'], g['mask'], g['sd'], g['desc']), g['loss'], 1e-4)
    assert O.maskgit_schedule(10, (16, 16)).tolist() == g['schedule_linear_10_16x16'] == [1, 6, 11, 17, 23, 28, 34, 40, 46, 50]
    assert O.maskgit_schedule(7, (8, 8), 'cosine').tolist() == g['schedule_cosine_7_8x8']
    assert O.maskgit_schedule(7, (8, 8), 'arccos').tolist() == g['schedule_arccos_7_8x8']
    with pytest.raises(ValueError):
        O.maskgit_schedule(5, (4, 4), 'nope')


def test_dynamics_generate_token_ids():
    """MaskGIT token ids: the oracle's loop == the real reference's generate() (torch.multinomial replaced by the injected-noise
    draw, tests/golden/make_golden_generate.py), bit for bit, on the reference's own test configuration."""
    g = load('dynamics_generate.pt')
    for name, e in g.items():
        gen = O.dynamics_generate(e['tokens'], e['act'], e['sd'], e['desc'], e['uniforms'], steps=e['steps'], which=e['which'], temp=e['temp'])
        assert gen.dtype == e['gen'].dtype and torch.equal(gen, e['gen']), name
        assert torch.equal(gen[:, :-1], e['tokens'])


def test_gan_critic_path():
    g = load('gan.pt')
    for i in range(3):
        e = g[f'image_residual_{i}']
        close(O.image_residual_block(e['x'], e['sd'], '', **e['kw']), e['out'], 1e-4)
    e = g['frame_discriminator']
    close(O.frame_discriminator(e['x'], e['sd'], '', **e['kw']), e['out'], 1e-4)
    e = g['gan_loss']
    sd = {'gan_crit.' + k: v for k, v in e['sd'].items()}
    close(O.gan_loss(e['rec'], e['video'], True, e['gen']['frame_idxs'], sd, **e['kw']), e['gen']['loss'], 1e-4)
    close(O.gan_loss(e['rec'], e['video'], False, e['dis']['frame_idxs'], sd, **e['kw']), e['dis']['loss'], 1e-4)


def test_latent_action_composition_vs_reference_pieces():
    """R-lam pinned to the reference's OWN pieces: tests/golden/lam_pieces.pt was produced by executing action.py:111-176
    on modules built from the reference's classes with the repaired blueprints (tests/golden/make_golden_lam.py) -- encoder stack,
    `to_act`, LFQ, decoder with the quantised action as temporal condition, `proj_out`, both losses and every parameter gradient."""
    g = load('lam_pieces.pt')
    sd = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and 'freq' not in k and 'codebook' not in k and 'bit_mask' not in k else v)
          for k, v in g['sd'].items()}
    trace = {}
    idxs, loss, (rec_loss, q_loss), recon = O.latent_action_forward(g['x'], sd, g['enc_desc'], g['dec_desc'], g['d_codebook'], training=True, trace=trace)
    o = g['out']
    close(trace['enc_video'], o['enc_video'])
    close(trace['act'], o['act_pre'])
    assert torch.equal(idxs, o['idxs'])
    close(recon, o['recon'])
    close(rec_loss, o['rec_loss'], 1e-5)
    close(q_loss, o['q_loss'], 1e-5)
    close(loss, o['loss'], 1e-5)
    loss.backward()
    checked = 0
    for k, gr in g['grads'].items():
        if gr.abs().max() == 0:
            continue
        assert sd[k].grad is not None, k
        close(sd[k].grad, gr, 2e-4)
        checked += 1
    assert checked >= 40, checked
