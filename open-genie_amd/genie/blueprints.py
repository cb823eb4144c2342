"""Shipped model descriptions that are not tied to one model file.

The reference's LATENT_ACT_ENC / LATENT_ACT_DEC (genie/__init__.py:10-54) cannot build: the space-time blocks get an `n_embd` keyword
their constructor rejects, 4 heads x 16 do not make the 256-wide stream LayerNorm expects, and 'spacetime_upsample' is not in the
registry.  These are the R-lam repaired forms: same structure and sizes, runnable -- n_head * d_head == 256,
channels-first blocks (transpose=True), 'depth2spacetime_upsample'."""

LATENT_ACT_ENC = (
    ('space-time_attn', {'n_rep': 2, 'n_head': 4, 'd_head': 64, 'transpose': True}),
    ('spacetime_downsample', {'in_channels': 256, 'kernel_size': 3, 'time_factor': 1, 'space_factor': 2}),
    ('space-time_attn', {'n_rep': 2, 'n_head': 4, 'd_head': 64, 'transpose': True}),
)

LATENT_ACT_DEC = (
    ('space-time_attn', {'n_rep': 2, 'n_head': 4, 'd_head': 64, 'transpose': True, 'has_ext': True, 'time_attn_kw': {'key_dim': 8}}),
    ('depth2spacetime_upsample', {'in_channels': 256, 'kernel_size': 3, 'time_factor': 1, 'space_factor': 2}),
    ('space-time_attn', {'n_rep': 2, 'n_head': 4, 'd_head': 64, 'transpose': True, 'has_ext': True, 'time_attn_kw': {'key_dim': 8}}),
)

# MaskGIT dynamics trunk of BASELINE configs[3]: 8 space-time blocks of 8 heads x 64 on channels-last token grids
DYNAMICS_DESC = (
    ('space-time_attn', {'n_rep': 8, 'n_head': 8, 'd_head': 64, 'transpose': False}),
)
