"""genie -- MI355X-native drop-in for the hot path of myscience/open-genie.

Public names mirror the reference package (reference genie/__init__.py:1-54).  All arithmetic of the hot path runs in
``lib/libgenie_hip.so`` (hand-written HIP for gfx950); importing this package on a machine without a GPU works (so
configs, registries and state_dicts can be inspected), but any forward pass raises -- there is no CPU fallback.
"""
from .action import LatentAction
from .dynamics import DynamicsModel
from .genie import Genie
from .tokenizer import MAGVIT2_DEC_DESC, MAGVIT2_ENC_DESC, REPR_TOK_DEC, REPR_TOK_ENC, VideoTokenizer

# The reference ships LATENT_ACT_ENC/DEC blueprints that cannot build (genie/__init__.py:10-54: `n_embd` keyword,
# 4 x 16 heads for a 256-wide stream, unregistered 'spacetime_upsample').  These are the R-lam repaired forms
# (SURVEY.md section 8c): same structure, runnable: n_head * d_head == 256, transpose=True, depth2spacetime_upsample.
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

__all__ = ['VideoTokenizer', 'LatentAction', 'DynamicsModel', 'Genie', 'MAGVIT2_ENC_DESC', 'MAGVIT2_DEC_DESC', 'REPR_TOK_ENC',
           'REPR_TOK_DEC', 'LATENT_ACT_ENC', 'LATENT_ACT_DEC']
