"""BASELINE config 0 (SURVEY.md 8d, R-yaml): the YAML configuration surface of the reference's LightningCLI entry point.
CPU part: both configs load and build; the shipped config is the reference's (same parsed content); the live reference fails on it
exactly the way this package does; the repaired config gives the reference's state_dict layout.  GPU part (-m gpu): the shipped
config raises the reference's AssertionError at encode, the repaired one trains and matches the oracle."""
import os

import pytest
import torch

from util import ROOT, bf16_round

CFG = os.path.join(ROOT, 'config', 'tokenize.yaml')
CFG_FIXED = os.path.join(ROOT, 'config', 'tokenize_repaired.yaml')
MSG = 'feature dimension 32 is not of sufficient size to rotate in all the positions 512'


def test_configs_load_and_build():
    from genie.cli import build_tokenizer, load_config, optimizer_factory
    cfg, fixed = load_config(CFG), load_config(CFG_FIXED)
    assert cfg['seed_everything'] == 31415 and cfg['model']['d_codebook'] == 10 and cfg['trainer']['strategy'] == 'ddp_find_unused_parameters_false'
    m = build_tokenizer(cfg, gan_loss_weight=0., perc_loss_weight=0.)          # builds (as in the reference); encode is what fails
    assert sum(p.numel() for p in m.parameters()) == 113_340_310
    m = build_tokenizer(fixed)
    assert sum(p.numel() for p in m.parameters()) == 113_522_198                # 113.5 M
    opt = m.configure_optimizers()
    assert type(opt).__name__ == 'AdamW' and opt.defaults['lr'] == 1e-3 and opt.defaults['weight_decay'] == 0.01
    assert optimizer_factory(None) is torch.optim.AdamW
    with pytest.raises(ValueError):
        load_config(os.path.join(ROOT, 'BASELINE.json'))


def test_shipped_config_is_the_references_and_fails_the_same_way():
    from oracle.ref_import import import_reference, reference_available
    if not reference_available():
        pytest.skip('reference not present')
    import copy

    import yaml
    from genie.cli import build_tokenizer, load_config
    ref_cfg = yaml.safe_load(open('/root/reference/config/tokenize.yaml'))
    assert load_config(CFG) == ref_cfg
    ref = import_reference()
    kw = copy.deepcopy(ref_cfg['model'])
    kw.pop('optimizer')
    kw['enc_desc'] = [tuple(d) for d in kw['enc_desc']]
    kw['dec_desc'] = [tuple(d) for d in kw['dec_desc']]
    rm = ref.VideoTokenizer(**{**kw, 'gan_loss_weight': 0., 'perc_loss_weight': 0.})
    with pytest.raises(AssertionError, match=MSG):
        rm.encode(torch.randn(1, 3, 2, 64, 64))
    # repaired: same state_dict layout as the reference built from the same repaired description
    fixed = load_config(CFG_FIXED)
    kw = copy.deepcopy(fixed['model'])
    kw.pop('optimizer')
    kw['enc_desc'] = [tuple(d) for d in kw['enc_desc']]
    kw['dec_desc'] = [tuple(d) for d in kw['dec_desc']]
    rm = ref.VideoTokenizer(**kw)
    ours = build_tokenizer(fixed)
    a = {k: tuple(v.shape) for k, v in rm.state_dict().items()}
    b = {k: tuple(v.shape) for k, v in ours.state_dict().items()}
    assert a == b


@pytest.mark.gpu
def test_config0_shipped_yaml_raises_repaired_yaml_runs():
    from genie.cli import build_tokenizer, load_config
    from oracle import genie_oracle as O
    cfg = load_config(CFG)
    torch.manual_seed(cfg['seed_everything'])
    m = build_tokenizer(cfg, gan_loss_weight=0., perc_loss_weight=0.).cuda()
    with pytest.raises(AssertionError, match=MSG):
        m.encode(torch.randn(4, 3, 16, 64, 64, device='cuda'))
    del m
    fixed = load_config(CFG_FIXED)
    torch.manual_seed(fixed['seed_everything'])
    m = build_tokenizer(fixed)
    with torch.no_grad():
        for p in m.parameters():
            if p.dim() >= 2:
                p.copy_(bf16_round(p))
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    m = m.cuda().train()
    video = torch.randn(4, 3, 16, 64, 64)
    enc = m.encode(video.cuda())
    assert tuple(enc.shape) == (4, 512, 16, 32, 32)                              # latent (B, 512, 16, 32, 32)
    q, idx = m.tokenize(video.cuda())
    assert tuple(idx.shape) == (4, 16, 32, 32) and idx.dtype == torch.int64 and int(idx.max()) < 1024
    loss, aux = m(video.cuda())
    assert torch.isfinite(loss).item()
    loss.backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for n, p in m.named_parameters() if p.requires_grad)
    # parity with the oracle on a clip it finishes in seconds
    x = bf16_round(torch.randn(1, 3, 4, 64, 64))
    enc_desc, dec_desc = fixed['model']['enc_desc'], fixed['model']['dec_desc']
    e_ref = O.tokenizer_encode(x, sd, enc_desc)
    e_hip = m.encode(x.cuda())
    rr = ((e_hip.float().cpu() - e_ref).pow(2).mean().sqrt() / e_ref.pow(2).mean().sqrt()).item()
    assert rr < 3e-2, rr
    m.zero_grad()
    loss, aux = m(x.cuda())
    ref, _, _, _ = O.tokenizer_forward_hotpath(x, sd, enc_desc, dec_desc, 10, entropy_weight=0.01, commit_weight=0.25, diversity_weight=1.)
    assert abs(loss.item() - ref.item()) < 5e-2 * abs(ref.item()), (loss.item(), ref.item())
