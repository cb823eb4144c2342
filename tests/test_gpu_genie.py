"""Genie and the command-line entry points on the MI355X (-m gpu)."""
import json
import os
import subprocess
import sys

import pytest
import torch

from util import ROOT, bf16_round, report

pytestmark = pytest.mark.gpu

TOK_ENC = (('spacetime_downsample', {'in_channels': 3, 'kernel_size': 3, 'out_channels': 64, 'time_factor': 2, 'space_factor': 4}),
           ('space-time_attn', {'n_rep': 1, 'n_head': 2, 'd_head': 32, 'transpose': True}))
TOK_DEC = (('space-time_attn', {'n_rep': 1, 'n_head': 2, 'd_head': 32, 'transpose': True}),
           ('depth2spacetime_upsample', {'in_channels': 64, 'kernel_size': 3, 'out_channels': 3, 'time_factor': 2, 'space_factor': 4}))
LAM_ENC = (('space-time_attn', {'n_head': 2, 'd_head': 32, 'transpose': True}),
           ('spacetime_downsample', {'in_channels': 64, 'kernel_size': 3, 'time_factor': 1, 'space_factor': 2}))
LAM_DEC = (('depth2spacetime_upsample', {'in_channels': 64, 'kernel_size': 3, 'time_factor': 1, 'space_factor': 2}),
           ('space-time_attn', {'n_head': 2, 'd_head': 32, 'transpose': True, 'has_ext': True, 'time_attn_kw': {'key_dim': 4}}))
DYN = (('space-time_attn', {'n_rep': 2, 'n_head': 2, 'd_head': 32}),)


def _genie():
    from genie import Genie, VideoTokenizer
    torch.manual_seed(0)
    tok = VideoTokenizer(TOK_ENC, TOK_DEC, d_codebook=6, gan_loss_weight=0., perc_loss_weight=0.)
    g = Genie(tok, enc_desc=LAM_ENC, dec_desc=LAM_DEC, d_codebook=4, inp_shape=(16, 16), n_embd=64, dyn_desc=DYN, embed_dim=64)
    with torch.no_grad():
        for p in g.parameters():
            if p.dim() >= 2:
                p.copy_(bf16_round(p))
    return g


def test_genie_compute_loss_matches_its_parts_and_oracle():
    """compute_loss = LatentAction loss + MaskGIT loss on the frozen tokenizer's index grid with the actions sub-sampled to the
    latent frame rate (R-genie repairs 2 and 3); both parts against the oracle on the same tokens / mask."""
    from oracle import genie_oracle as O
    g = _genie()
    sd_lam = {k: v.detach().clone() for k, v in g.latent_action.state_dict().items()}
    sd_dyn = {k: v.detach().clone() for k, v in g.dynamics_model.state_dict().items()}
    g = g.cuda().train()
    assert g.tok_codebook == 64 and g.act_codebook == 16
    x = bf16_round(torch.rand(2, 3, 8, 16, 16))
    torch.manual_seed(11)
    loss, aux = g.compute_loss(x.cuda())
    aux = dict(aux)
    loss.backward()
    tokens = g._token_grid(x.cuda()).cpu()
    assert tuple(tokens.shape) == (2, 4, 4, 4)                                         # time / 2, space / 4
    idx_ref, act_loss_ref, _, _ = O.latent_action_forward(x, sd_lam, LAM_ENC, LAM_DEC, 4, training=True)
    assert abs(aux['act_loss'].item() - act_loss_ref.item()) < 4e-2 * abs(act_loss_ref.item()), (aux['act_loss'].item(), act_loss_ref.item())
    torch.manual_seed(11)                                                                # the Bernoulli mask of compute_loss (dynamics.py:77-80)
    mask = torch.distributions.Bernoulli(torch.empty(1).uniform_(0.5, 1).item()).sample((2, 4, 4, 4)).bool()
    act_hip, _, _ = g.latent_action(x.cuda())
    dyn_ref = O.dynamics_loss(tokens, act_hip.cpu()[:, 1::2], mask, sd_dyn, DYN)
    assert abs(aux['dyn_loss'].item() - dyn_ref.item()) < 3e-2 * abs(dyn_ref.item()) + 1e-3, (aux['dyn_loss'].item(), dyn_ref.item())
    assert abs(loss.item() - (aux['act_loss'] + aux['dyn_loss']).item()) < 1e-5
    # tensor-level checks (a scalar forgives a wrong layer):
    # (a) the token grid: bit-exact against the oracle's LFQ on the HIP encoder output; against the fp32 oracle end to end wherever no latent
    #     component is within eps of zero
    sd_tok = {k: v.detach().clone().cpu() for k, v in g.tokenizer.state_dict().items()}
    with torch.no_grad():
        g.tokenizer.eval()
        e_hip = g.tokenizer.encode(x.cuda())
        g.tokenizer.train()
    (_, idx_op), _ = O.lfq_forward(e_hip.float().cpu(), sd_tok, 'quant.', 6, 1, training=False, transpose=True)
    assert torch.equal(tokens, idx_op.reshape(tokens.shape))
    e_ref = O.tokenizer_encode(x, sd_tok, TOK_ENC)
    _, idx_ref_tok = O.tokenizer_tokenize(x, sd_tok, TOK_ENC, 6)
    z_ref = torch.nn.functional.linear(e_ref.movedim(1, -1), sd_tok['quant.proj_inp.weight'], sd_tok['quant.proj_inp.bias'])    # the 6 pre-sign values per token
    safe = (z_ref.abs() >= 6e-2 * z_ref.pow(2).mean().sqrt()).all(-1)                   # (B, t, h, w): every bit decided by a margin (6 % of the RMS: ~4x the
    #                                                                                     # bf16-vs-fp32 error of the encoder output)
    assert safe.float().mean() > 0.5, safe.float().mean()
    assert torch.equal(tokens[safe], idx_ref_tok.reshape(tokens.shape)[safe])
    # (b) the action ids LatentAction hands to the dynamics model, same rule
    tr = {}
    with torch.no_grad():
        O.latent_action_forward(x, sd_lam, LAM_ENC, LAM_DEC, 4, training=True, trace=tr)
    safe_a = (tr['act'].abs() >= 6e-2 * tr['act'].pow(2).mean().sqrt()).all(-1)
    assert torch.equal(act_hip.cpu().reshape(safe_a.shape)[safe_a], idx_ref.reshape(safe_a.shape)[safe_a])
    # (c) the dynamics logits on the shared token grid and actions, and the gradients of compute_loss's MaskGIT term
    acts = act_hip.cpu().reshape(2, 8)[:, 1::2]
    with torch.no_grad():
        logits_hip, _ = g.dynamics_model(tokens.cuda(), acts.cuda())
    sd_req = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and 'freq' not in k else v) for k, v in sd_dyn.items()}
    logits_ref, _ = O.dynamics_forward(tokens, acts, sd_req, DYN)
    r_logits = ((logits_hip.float().cpu() - logits_ref).pow(2).mean().sqrt() / logits_ref.pow(2).mean().sqrt()).item()
    assert r_logits < 2e-2, r_logits
    O.dynamics_loss(tokens, acts, mask, sd_req, DYN).backward()
    worst = 0.
    for n, p in g.dynamics_model.named_parameters():
        gr = sd_req[n].grad if sd_req[n].requires_grad else None
        if gr is None or gr.abs().max() == 0:
            continue
        r = ((p.grad.float().cpu() - gr).pow(2).mean().sqrt() / gr.pow(2).mean().sqrt()).item()
        worst = max(worst, r)
        assert r < 6e-2, (n, r)
    report('genie_compute_loss_parts', token_grid_bit_exact=True, safe_token_fraction=safe.float().mean().item(), logits_rel_rms=r_logits,
           dyn_grad_worst_rel_rms=worst, act_loss_hip=aux['act_loss'].item(), act_loss_oracle=act_loss_ref.item(), dyn_loss_hip=aux['dyn_loss'].item(),
           dyn_loss_oracle=dyn_ref.item())
    assert all(p.grad is None for p in g.tokenizer.parameters())
    missing = [n for n, p in g.named_parameters() if p.requires_grad and p.grad is None and 'freq' not in n]
    assert not missing, missing
    assert sum(p.numel() for p in g.configure_optimizers().param_groups[0]['params']) == sum(p.numel() for p in g.parameters() if p.requires_grad)


def test_genie_generates_video_from_prompt():
    g = _genie().cuda().eval()
    prompt = torch.rand(2, 3, 4, 16, 16, device='cuda')                                # 4 frames -> 2 latent frames
    actions = torch.randint(0, 16, (2, 6), device='cuda')
    video = g(prompt, actions, num_frames=2, steps_per_frame=3)
    assert tuple(video.shape) == (2, 3, 8, 16, 16) and torch.isfinite(video.float()).all()     # (2 + 2) latent frames x 2
    with pytest.raises(ValueError):
        g(torch.rand(2, 16, device='cuda'), actions)


@pytest.mark.parametrize('batch', [1, 2])
def test_genie_image_prompt(batch):
    """A single-frame prompt (reference genie.py:78-87, the on_validation_end usage): the tokenizer's ``idxs.squeeze()`` drops the frame
    axis (and the batch axis at B = 1); the grid must come back as (B, 1, h, w), not as B context frames of one clip."""
    from genie import Genie, VideoTokenizer
    torch.manual_seed(0)
    enc = (('spacetime_downsample', {'in_channels': 3, 'kernel_size': 3, 'out_channels': 64, 'time_factor': 1, 'space_factor': 4}),
           ('space-time_attn', {'n_rep': 1, 'n_head': 2, 'd_head': 32, 'transpose': True}))
    dec = (('space-time_attn', {'n_rep': 1, 'n_head': 2, 'd_head': 32, 'transpose': True}),
           ('depth2spacetime_upsample', {'in_channels': 64, 'kernel_size': 3, 'out_channels': 3, 'time_factor': 1, 'space_factor': 4}))
    tok = VideoTokenizer(enc, dec, d_codebook=6, gan_loss_weight=0., perc_loss_weight=0.)
    g = Genie(tok, enc_desc=LAM_ENC, dec_desc=LAM_DEC, d_codebook=4, inp_shape=(16, 16), n_embd=64, dyn_desc=DYN, embed_dim=64).cuda().eval()
    img = torch.rand(batch, 3, 16, 16, device='cuda')
    grid = g._token_grid(img[:, :, None])
    assert tuple(grid.shape) == (batch, 1, 4, 4)
    actions = torch.randint(0, 16, (batch, 4), device='cuda')
    video = g(img, actions, num_frames=3, steps_per_frame=2)
    assert tuple(video.shape) == (batch, 3, 4, 16, 16) and torch.isfinite(video.float()).all()
    # each sample's first generated context is ITS OWN prompt frame: sample 0 of a batch equals the same prompt run alone
    if batch == 2:
        torch.manual_seed(5)
        u = torch.rand(2, 2 * 16)
        a = g.dynamics_model.generate(grid, actions[:, :1], steps=2, uniforms=u)
        b1 = g.dynamics_model.generate(grid[:1], actions[:1, :1], steps=2, uniforms=u[:, :16])
        assert torch.equal(a[:1], b1)


def _run(args, timeout=900):
    env = dict(os.environ, GENIE_USE_LIGHTNING='0')
    r = subprocess.run([sys.executable, *args], cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    return [json.loads(l) for l in r.stdout.splitlines() if l.startswith('{')]


def test_tokenizer_entry_point_fits_from_yaml(tmp_path):
    """`python tokenizer.py fit --config config/tokenize_repaired.yaml` (reference tokenizer.py:6-19 / README: LightningCLI surface):
    three steps on synthetic clips, one JSON log line per step, a loadable last.ckpt."""
    out = str(tmp_path / 'run')
    logs = _run(['tokenizer.py', 'fit', '--config', 'config/tokenize_repaired.yaml', '--trainer.max_steps', '3', '--trainer.log_every_n_steps', '1',
                 '--trainer.default_root_dir', out, '--data.synthetic', 'true', '--data.batch_size', '1', '--data.num_frames', '4'])
    train = [l for l in logs if l['split'] == 'train']
    assert len(train) >= 3 and all('train_loss' in l and l['train_loss'] == l['train_loss'] for l in train)
    ck = torch.load(os.path.join(out, 'last.ckpt'), map_location='cpu')
    assert ck['global_step'] == 3
    from genie.cli import build_tokenizer, load_config
    m = build_tokenizer(load_config(os.path.join(ROOT, 'config', 'tokenize_repaired.yaml')))
    m.load_state_dict(ck['state_dict'])
    # the shipped (unrepaired) config fails the way the reference does
    env = dict(os.environ, GENIE_USE_LIGHTNING='0')
    r = subprocess.run([sys.executable, 'tokenizer.py', 'fit', '--config', 'config/tokenize.yaml', '--model.gan_loss_weight', '0.', '--model.perc_loss_weight', '0.',
                        '--trainer.max_steps', '1', '--data.synthetic', 'true', '--data.batch_size', '1', '--data.num_frames', '4'],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and 'is not of sufficient size to rotate in all the positions 512' in r.stderr


def test_genie_entry_point_fits_from_yaml(tmp_path):
    out = str(tmp_path / 'run')
    logs = _run(['genie.py', 'fit', '--config', 'config/genie.yaml', '--trainer.max_steps', '2', '--trainer.log_every_n_steps', '1',
                 '--trainer.default_root_dir', out, '--data.synthetic', 'true', '--data.batch_size', '2', '--data.num_frames', '4'])
    train = [l for l in logs if l['split'] == 'train']
    assert len(train) >= 2 and all(k in train[0] for k in ('train_loss', 'train/act_loss', 'train/dyn_loss'))
    assert os.path.exists(os.path.join(out, 'last.ckpt'))
