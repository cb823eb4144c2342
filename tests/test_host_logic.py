"""Host-side logic that needs no GPU: registry/blueprint contract, conv geometry, state_dict layout, the
parameter arena, clip sharding, and the loud failure of the product path on CPU tensors."""
import copy
import os

import pytest
import torch

from util import ROOT

GOLD = os.path.join(ROOT, 'tests', 'golden')


def test_registry_contract():
    from genie.module import get_module, parse_blueprint
    from genie.module.video import CausalConv3d, VideoResidualBlock
    assert get_module('causal-conv3d') is CausalConv3d and get_module('video-residual') is VideoResidualBlock
    for name in ('space_attn', 'time_attn', 'space-time_attn', 'depth2space_upsample', 'depth2time_upsample',
                 'depth2spacetime_upsample', 'spacetime_downsample', 'group_norm', 'adaptive_group_norm', 'gelu', 'relu',
                 'leaky_relu', 'silu', 'causal-conv3d-transpose', 'blur_pool', 'space_downsample', 'image-residual'):
        assert get_module(name) is not None
    with pytest.raises(ValueError, match='Unknown module name'):       # reference genie/module/__init__.py:69
        get_module('nope')
    desc = (('video-residual', {'n_rep': 3, 'in_channels': 8, 'has_ext': False}), 'silu',
            ('adaptive_group_norm', {'dim_cond': 4, 'num_groups': 2, 'num_channels': 8, 'has_ext': True}))
    layers, ext = parse_blueprint(desc)
    assert len(layers) == 5 and ext == [False, False, False, False, True]
    assert 'n_rep' not in desc[0][1] and 'has_ext' not in desc[0][1]   # the reference mutates the caller's dicts too (:82-90)


def test_causal_geometry_matches_reference_formula():
    from genie.conv import causal_spec, same_spec
    from oracle import genie_oracle as O
    for k, s, d in [((3, 3, 3), (1, 1, 1), (1, 1, 1)), ((3, 3, 3), (2, 2, 2), (1, 1, 1)), ((1, 1, 1), (1, 1, 1), (1, 1, 1)),
                    ((3, 5, 5), (1, 2, 2), (2, 1, 1)), ((2, 3, 3), (2, 1, 1), (1, 1, 1))]:
        spec = causal_spec(4, 4, k, s, d)
        assert spec.pad_front == O.causal_pad_amounts(k, s, d) and spec.pad_back[0] == 0
        x = torch.zeros(1, 4, 9, 12, 12)
        ref = O.causal_conv3d(x, torch.zeros(4, 4, *k), None, stride=s, dilation=d)
        assert spec.out_size((9, 12, 12)) == tuple(ref.shape[2:])
    assert same_spec(4, 4, (3, 3, 3)).pad_front == (1, 1, 1) == same_spec(4, 4, (3, 3, 3)).pad_back
    with pytest.raises(ValueError):
        causal_spec(4, 4, (1, 3, 3), (2, 1, 1))                       # negative causal pad


def test_magvit2_state_dict_layout_matches_reference():
    from genie import MAGVIT2_DEC_DESC, MAGVIT2_ENC_DESC, VideoTokenizer
    layout = torch.load(os.path.join(GOLD, 'magvit2_state_dict_layout.pt'), weights_only=False)
    m = VideoTokenizer(MAGVIT2_ENC_DESC, MAGVIT2_DEC_DESC, d_codebook=18, gan_loss_weight=0., perc_loss_weight=0.)
    mine = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    assert mine == layout and len(mine) == 449
    assert sum(p.numel() for p in m.parameters()) == 375_554_837
    # module-level descs are not consumed by construction (they are shared objects)
    assert dict(MAGVIT2_ENC_DESC[1][1]) == {'n_rep': 4, 'in_channels': 128}


def test_reference_checkpoint_loads():
    from genie import VideoTokenizer
    g = torch.load(os.path.join(GOLD, 'tokenizer_small.pt'), weights_only=False)
    m = VideoTokenizer(g['enc_desc'], g['dec_desc'], d_codebook=g['d_codebook'], gan_loss_weight=0., perc_loss_weight=0.)
    missing, unexpected = m.load_state_dict(g['sd'], strict=True)
    assert not missing and not unexpected
    assert torch.equal(m.enc_layers[0].conv3d.weight.detach(), g['sd']['enc_layers.0.conv3d.weight'])


def test_product_path_fails_loudly_without_gpu():
    from genie import VideoTokenizer
    g = torch.load(os.path.join(GOLD, 'tokenizer_small.pt'), weights_only=False)
    m = VideoTokenizer(g['enc_desc'], g['dec_desc'], d_codebook=g['d_codebook'], gan_loss_weight=0., perc_loss_weight=0.)
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        m.encode(g['x'])
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        m.quant(torch.randn(2, 6, 2, 4, 4), transpose=True)


def test_default_tokenizer_critics_are_out_of_scope():
    from genie import VideoTokenizer
    g = torch.load(os.path.join(GOLD, 'tokenizer_small.pt'), weights_only=False)
    m = VideoTokenizer(g['enc_desc'], g['dec_desc'], d_codebook=6)        # default gan/perc weights = 1
    with pytest.raises(NotImplementedError, match='outside the implemented hot path'):
        m.gan_crit(None, None, train_gen=True)


def test_param_arena_views_and_offsets():
    from genie import VideoTokenizer
    from genie.trainer import ParamArena
    g = torch.load(os.path.join(GOLD, 'tokenizer_small.pt'), weights_only=False)
    m = VideoTokenizer(g['enc_desc'], g['dec_desc'], d_codebook=6, gan_loss_weight=0., perc_loss_weight=0.)
    m.load_state_dict(g['sd'])
    before = {k: v.clone() for k, v in m.state_dict().items()}
    arena = ParamArena(m)
    for k, v in m.state_dict().items():
        assert torch.equal(v, before[k])
    n = sum(p.numel() for p in m.parameters())
    assert arena.numel >= n and arena.numel % 64 == 0
    for name, p in m.named_parameters():
        off, cnt = arena.slots[name]
        assert p.data_ptr() == arena.params.data_ptr() + 4 * off and p.grad.data_ptr() == arena.grads.data_ptr() + 4 * off
        assert p.grad.stride() == p.stride()
    w = m.enc_layers[1].main[2].weight
    assert w.stride()[1] == 1                                         # conv weights stay channels_last_3d inside the arena
    p0 = next(m.parameters())
    p0.grad.fill_(3.)
    assert arena.grads[:p0.numel()].eq(3.).all()
    assert arena.offset_of(m.dec_layers[0], m) == arena.slots['dec_layers.0.conv3d.weight'][0]


def test_shard_clips():
    from genie.trainer import shard_clips
    shards = [list(shard_clips(10, r, 4)) for r in range(4)]
    assert shards == [[0, 4, 8], [1, 5, 9], [2, 6], [3, 7]]
    assert sorted(sum(shards, [])) == list(range(10))
