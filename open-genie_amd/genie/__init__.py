"""genie -- MI355X-native drop-in for the hot path of myscience/open-genie.

Public names mirror the reference package (reference genie/__init__.py:1-54).  All arithmetic of the hot path runs in
``lib/libgenie_hip.so`` (hand-written HIP for gfx950); importing this package on a machine without a GPU works (so
configs, registries and state_dicts can be inspected), but any forward pass raises -- there is no CPU fallback.
"""
from .action import LatentAction
from .blueprints import DYNAMICS_DESC, LATENT_ACT_DEC, LATENT_ACT_ENC
from .dynamics import DynamicsModel
from .genie import Genie
from .tokenizer import MAGVIT2_DEC_DESC, MAGVIT2_ENC_DESC, REPR_TOK_DEC, REPR_TOK_ENC, VideoTokenizer

__all__ = ['VideoTokenizer', 'LatentAction', 'DynamicsModel', 'Genie', 'MAGVIT2_ENC_DESC', 'MAGVIT2_DEC_DESC', 'REPR_TOK_ENC',
           'REPR_TOK_DEC', 'LATENT_ACT_ENC', 'LATENT_ACT_DEC', 'DYNAMICS_DESC']
