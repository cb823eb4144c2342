"""Generate the golden fixtures under tests/golden/ by RUNNING THE REAL REFERENCE (needs /root/reference; build
container only).  The reference holds no golden vectors of its own, so these seeded input/output
pairs are what pins oracle/genie_oracle.py -- and through it the HIP path -- to the reference's behaviour.

    python tests/golden/make_golden.py

Every fixture is a small ``torch.save`` dict: inputs, the reference's state_dict, and the reference's outputs.
"""
import copy
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

from oracle.ref_import import import_reference, ref_module  # noqa: E402

SMALL_ENC = (
    ('causal-conv3d', {'in_channels': 3, 'out_channels': 16, 'kernel_size': 3}),
    ('video-residual', {'n_rep': 2, 'in_channels': 16}),
    ('spacetime_downsample', {'in_channels': 16, 'out_channels': 16, 'kernel_size': 3, 'time_factor': 2, 'space_factor': 2}),
    ('video-residual', {'in_channels': 16, 'out_channels': 32}),
    ('group_norm', {'num_groups': 8, 'num_channels': 32}),
    ('silu', {}),
    ('causal-conv3d', {'in_channels': 32, 'out_channels': 6, 'kernel_size': 1}),
)
SMALL_DEC = (
    ('causal-conv3d', {'in_channels': 6, 'out_channels': 32, 'kernel_size': 3}),
    ('video-residual', {'n_rep': 2, 'in_channels': 32}),
    ('adaptive_group_norm', {'dim_cond': 6, 'num_groups': 8, 'num_channels': 32, 'has_ext': True}),
    ('depth2spacetime_upsample', {'in_channels': 32, 'kernel_size': 3, 'time_factor': 2, 'space_factor': 2}),
    ('video-residual', {'in_channels': 32, 'out_channels': 16}),
    ('group_norm', {'num_groups': 8, 'num_channels': 16}),
    ('silu', {}),
    ('causal-conv3d', {'in_channels': 16, 'out_channels': 3, 'kernel_size': 3}),
)
DYN_DESC = (('space-time_attn', {'n_rep': 2, 'n_head': 4, 'd_head': 8}),)


def bf16r(t):
    return t.to(torch.bfloat16).float()


def sd_of(m):
    return {k: v.detach().clone() for k, v in m.state_dict().items()}


def round_weights(m):
    """Make every matrix-like parameter bf16-representable, so the bf16 HIP path and the fp32 reference multiply
    IDENTICAL numbers and differ only by accumulation order / activation rounding."""
    with torch.no_grad():
        for n, p in m.named_parameters():
            if p.dim() >= 2:
                p.copy_(bf16r(p))


def save(name, obj):
    path = os.path.join(HERE, name)
    torch.save(obj, path)
    print(f'{name}: {os.path.getsize(path) / 1024:.0f} KiB')


def main():
    ref = import_reference()
    V, A, Q, N = ref_module('module.video'), ref_module('module.attention'), ref_module('module.quantization'), ref_module('module.norm')

    # ---- operators ----------------------------------------------------------------------------------------
    ops = {}
    torch.manual_seed(100)
    for i, (ci, co, k, s) in enumerate([(3, 16, 3, (1, 1, 1)), (16, 8, 3, (2, 2, 2)), (8, 24, 1, (1, 1, 1)), (16, 16, 3, (1, 2, 2))]):
        m = V.CausalConv3d(ci, co, k, stride=s); round_weights(m)
        x = bf16r(torch.randn(2, ci, 5, 8, 8))
        ops[f'causal_conv3d_{i}'] = dict(cin=ci, cout=co, kernel=k, stride=s, x=x, weight=m.conv3d.weight.detach().clone(),
                                        bias=m.conv3d.bias.detach().clone(), out=m(x).detach())
    for i, kw in enumerate([dict(in_channels=16), dict(in_channels=16, out_channels=32), dict(in_channels=16, use_causal=True),
                            dict(in_channels=16, out_channels=32, downsample=(2, 2))]):
        m = V.VideoResidualBlock(**kw); round_weights(m)
        x = bf16r(torch.randn(2, 16, 4, 8, 8))
        ops[f'video_residual_{i}'] = dict(kw=kw, x=x, sd=sd_of(m), out=m(x).detach())
    m = V.DepthToSpaceTimeUpsample(16, out_channels=8, time_factor=2, space_factor=2, kernel_size=3); round_weights(m)
    x = bf16r(torch.randn(2, 16, 3, 4, 4))
    ops['depth2spacetime'] = dict(x=x, sd=sd_of(m), out=m(x).detach())
    m = N.AdaptiveGroupNorm(6, 4, 16)
    for p in m.parameters():
        torch.nn.init.normal_(p)
    x, c = bf16r(torch.randn(2, 16, 3, 4, 4)), bf16r(torch.randn(2, 6, 2, 2, 2))
    ops['adagn'] = dict(x=x, cond=c, sd=sd_of(m), out=m(x, c).detach())
    gn = torch.nn.GroupNorm(8, 32)
    for p in gn.parameters():
        torch.nn.init.normal_(p)
    x = bf16r(torch.randn(2, 32, 3, 4, 4) * 2 + 0.5)
    ops['groupnorm_silu'] = dict(x=x, sd=sd_of(gn), out=torch.nn.functional.silu(gn(x)).detach())
    save('ops.pt', ops)

    # ---- LFQ --------------------------------------------------------------------------------------------
    lfq = {}
    torch.manual_seed(101)
    for name, (d, n, inp) in {'d18': (18, 1, 18), 'd8': (8, 1, 8), 'd6x3_proj': (6, 3, 32), 'd10': (10, 1, 10)}.items():
        m = Q.LookupFreeQuantization(d, n, input_dim=inp)
        x = bf16r(torch.randn(2, inp, 2, 4, 4) * (0.05 if d == 18 else 0.5))
        x[0, :, 0, 0, 0] = 0.
        m.eval()
        (oe, ie), _ = m(x, transpose=True)
        entry = dict(d=d, n=n, inp=inp, x=x, sd=sd_of(m), eval_out=oe.detach(), eval_idx=ie.detach())
        if d <= 10:                  # the reference's training path materialises N x 2^d probabilities
            m.train()
            (ot, it), loss = m(x, transpose=True)
            entry.update(train_out=ot.detach(), train_idx=it.detach(), train_loss=loss.detach())
        lfq[name] = entry
    # d = 18 training loss on a handful of tokens (64 x 2^18 fp32 = 64 MiB in the reference)
    m = Q.LookupFreeQuantization(18, 1, input_dim=18).train()
    x = bf16r(torch.randn(1, 18, 1, 8, 8) * 0.03)
    (_, _), loss = m(x, transpose=True)
    lfq['d18_train'] = dict(d=18, n=1, inp=18, x=x, sd=sd_of(m), train_loss=loss.detach())
    save('lfq.pt', lfq)

    # ---- VideoTokenizer (small blueprint) ------------------------------------------------------------------
    torch.manual_seed(102)
    m = ref.VideoTokenizer(copy.deepcopy(SMALL_ENC), copy.deepcopy(SMALL_DEC), d_codebook=6, gan_loss_weight=0., perc_loss_weight=0.)
    for n_, p in m.named_parameters():
        if '.std.' in n_ or '.avg.' in n_:
            torch.nn.init.normal_(p, std=0.3)
    round_weights(m)
    x = bf16r(torch.randn(2, 3, 4, 16, 16))
    enc = m.encode(x)
    q, idx = m.tokenize(x)
    rec = m.decode(q)
    m.train()
    (qt, _), ql = m.quant(enc, transpose=True)
    loss = torch.nn.functional.mse_loss(m.decode(qt), x) + ql      # R-fwd: forward() itself needs VGG16 weights
    save('tokenizer_small.pt', dict(enc_desc=SMALL_ENC, dec_desc=SMALL_DEC, d_codebook=6, x=x, sd=sd_of(m), enc=enc.detach(), quant=q.detach(),
                                    idx=idx.detach(), rec=rec.detach(), rfwd_loss=loss.detach(), quant_loss=ql.detach()))

    # ---- state_dict layout of the real MAGVIT2 tokenizer (keys + shapes only) ----------------------------------
    m = ref.VideoTokenizer(copy.deepcopy(ref.MAGVIT2_ENC_DESC), copy.deepcopy(ref.MAGVIT2_DEC_DESC), d_codebook=18, gan_loss_weight=0., perc_loss_weight=0.)
    save('magvit2_state_dict_layout.pt', {k: tuple(v.shape) for k, v in m.state_dict().items()})

    # ---- space-time block -----------------------------------------------------------------------------------
    st = {}
    torch.manual_seed(103)
    for name, (tr, kw) in {'cl': (False, {}), 'cf': (True, {}), 'cf_cond': (True, {'time_attn_kw': {'key_dim': 6}})}.items():
        m = A.SpaceTimeAttention(n_head=4, d_head=16, transpose=tr, **kw)
        for n_, p in m.named_parameters():
            if 'freq' not in n_:
                torch.nn.init.normal_(p, std=0.5 if p.dim() < 2 else 0.05)
        round_weights(m)
        x = bf16r(torch.randn(2, 64, 5, 4, 6) if tr else torch.randn(2, 5, 4, 6, 64))
        cond = bf16r(torch.randn(2, 5, 6)) if kw else None
        out = m(x, cond=(None, cond)) if kw else m(x)
        st[name] = dict(transpose=tr, kw=kw, x=x, cond=cond, sd=sd_of(m), out=out.detach())
    save('st_block.pt', st)

    # ---- DynamicsModel ----------------------------------------------------------------------------------------
    torch.manual_seed(104)
    m = ref.DynamicsModel(copy.deepcopy(DYN_DESC), tok_vocab=64, act_vocab=5, embed_dim=32)
    round_weights(m)
    tok, act = torch.randint(0, 64, (2, 5, 4, 4)), torch.randint(0, 5, (2, 5))
    logits, last = m(tok, act)
    mask = torch.rand(2, 5, 4, 4) < 0.7
    loss = m.compute_loss(tok, act, mask=mask)
    save('dynamics_small.pt', dict(desc=DYN_DESC, tok_vocab=64, act_vocab=5, embed_dim=32, tokens=tok, act=act, mask=mask, sd=sd_of(m),
                                   logits=logits.detach(), loss=loss.detach(),
                                   schedule_linear_10_16x16=m.get_schedule(10, (16, 16)).tolist(),
                                   schedule_cosine_7_8x8=m.get_schedule(7, (8, 8), 'cosine').tolist(),
                                   schedule_arccos_7_8x8=m.get_schedule(7, (8, 8), 'arccos').tolist()))


if __name__ == '__main__':
    main()
