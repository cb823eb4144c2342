"""Host-side logic that needs no GPU: registry/blueprint contract, conv geometry, state_dict layout, the
parameter arena, clip sharding, and the loud failure of the product path on CPU tensors."""
import copy
import os
import sys

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
    # negative causal pad (kt = 1, time stride 2): the reference crops the first frame (F.pad with -1, video.py:154-164); here the module
    # slices `causal_time_crop` frames off and the conv runs without time padding -- same output size
    from genie.conv import causal_time_crop
    spec = causal_spec(4, 4, (1, 3, 3), (2, 1, 1))
    assert spec.pad_front[0] == 0 and causal_time_crop((1, 3, 3), (2, 1, 1)) == 1 and causal_time_crop((3, 3, 3), (2, 2, 2)) == 0
    ref = O.causal_conv3d(torch.zeros(1, 4, 9, 12, 12), torch.zeros(4, 4, 1, 3, 3), None, stride=(2, 1, 1))
    assert spec.out_size((9 - 1, 12, 12)) == tuple(ref.shape[2:])


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


def test_tokenizer_critics():
    """GAN critic: built from disc_kwargs like the reference (tokenizer.py:294-299; with the default empty disc_kwargs the reference's
    FrameDiscriminator() call fails for want of inp_size, and so does this one).  Perceptual critic: needs VGG16 weights that do not
    exist offline -> calling it raises with a pointer to perc_loss_weight=0."""
    from genie import VideoTokenizer
    from genie.module.loss import GANLoss
    g = torch.load(os.path.join(GOLD, 'tokenizer_small.pt'), weights_only=False)
    with pytest.raises(TypeError, match='inp_size'):
        VideoTokenizer(g['enc_desc'], g['dec_desc'], d_codebook=6)        # default gan/perc weights = 1, disc_kwargs = {}
    m = VideoTokenizer(g['enc_desc'], g['dec_desc'], d_codebook=6, disc_kwargs={'inp_size': 16, 'model_dim': 8})
    assert isinstance(m.gan_crit, GANLoss) and sum(p.numel() for p in m.gan_crit.parameters()) > 0
    with pytest.raises(NotImplementedError, match='outside the implemented hot path'):
        m.perc_crit(None, None)


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


def test_kw_triple_schedule_host_logic():
    """conv.tri_rows (the schedule conv_igemm3.hip consumes): grouping, channel blocks, weight offsets, rejections; and the
    forward / backward-data tap lists it is built from."""
    from genie import conv as gconv
    spec = gconv.same_spec(128, 64, (3, 3, 3))
    taps = gconv._fwd_tap_list(spec)
    assert len(taps) == 27 and taps[0] == (-1, -1, -1, 0, 0, 128) and taps[13] == (0, 0, 0, 13 * 128, 0, 128)
    rows = gconv.tri_rows(taps, hs=8, ws=16, cs=128)
    assert len(rows) == 9 * 2                                        # 9 (dt, dh) pairs x 2 channel blocks
    a_delta, dt, dh, w0, w1, w2, rows_per_dt, dt_min = rows[0]
    assert (dt, dh) == (-1, -1) and a_delta == ((-1 * 8 - 1) * 16) * 128 and (w0, w1, w2) == (0, 128, 256)
    # inside a dt: channel block outer, dh inner (the three row-shifted images of one channel block back to back: L2 reuse, TRI_DH_INNER)
    assert [(r[1], r[2]) for r in rows[:6]] == [(-1, -1), (-1, 0), (-1, 1)] * 2
    assert rows[1][0] == rows[0][0] + 16 * 128 and rows[3][0] == rows[0][0] + 64 and rows[3][3:6] == [64, 192, 320]   # row 3: second 64-channel block of the first triple
    old_order = __import__('unittest.mock').mock.patch.object(gconv, 'TRI_DH_INNER', False)
    with old_order:
        r_old = gconv.tri_rows(taps, hs=8, ws=16, cs=128)
    assert r_old[1][0] == r_old[0][0] + 64 and sorted(map(tuple, r_old)) == sorted(map(tuple, rows))                 # same rows, (dh, channel block) order
    # sorted by dt, every dt owning rows_per_dt consecutive rows (what lets a row tile inside one frame skip the padding frames)
    assert (rows_per_dt, dt_min) == (6, -1) and all(r[6:8] == [6, -1] for r in rows)
    assert [r[1] for r in rows] == [-1] * 6 + [0] * 6 + [1] * 6
    # causal conv: all time padding in front -> dt in {-2, -1, 0}
    ctaps = gconv._fwd_tap_list(gconv.causal_spec(64, 64, (3, 3, 3)))
    assert sorted({t[0] for t in ctaps}) == [-2, -1, 0]
    crows = gconv.tri_rows(ctaps, 4, 8, 64)
    assert len(crows) == 9 and crows[0][6:8] == [3, -2] and [r[1] for r in crows] == [-2] * 3 + [-1] * 3 + [0] * 3
    # backward-data taps of a stride-1 conv: offsets flipped, still complete triples (dw = +1, 0, -1 order)
    dtaps = gconv.dgrad_taps(spec, (0, 0, 0), 64, want_list=True)
    drows = gconv.tri_rows(dtaps, 8, 16, 64)
    assert len(drows) == 9 and drows[-1][1] == 1 and drows[0][1] == -1 and drows[0][3] > drows[0][5]   # dw = -1 reads the LAST kw slice
    assert sorted((r[1], r[2]) for r in drows) == [(a, b) for a in (-1, 0, 1) for b in (-1, 0, 1)]
    # not eligible: 1-wide kernels, channel counts that are not whole 64-blocks, dilated w taps
    assert gconv.tri_rows(gconv._fwd_tap_list(gconv.same_spec(64, 64, (1, 1, 1))), 8, 8, 64) is None
    assert gconv.tri_rows(gconv._fwd_tap_list(gconv.same_spec(72, 64, (3, 3, 3))), 8, 8, 72) is None
    dil = gconv.ConvSpec(64, 64, (1, 1, 3), (1, 1, 1), (1, 1, 2), (0, 0, 2), (0, 0, 2), None)
    assert gconv.tri_rows(gconv._fwd_tap_list(dil), 8, 8, 64) is None


def test_conv_variant_table_matches_header():
    """Every GENIE_VARIANT_* id of include/genie_hip.h has a name in genie.conv.VARIANT_NAMES (bench.py keys its roofline on them)."""
    import re
    from genie import conv
    hdr = open(os.path.join(ROOT, 'include', 'genie_hip.h')).read()
    ids = {int(v): k for k, v in re.findall(r'#define (GENIE_VARIANT_\w+) (\d+)', hdr)}
    assert ids and set(ids) == set(conv.VARIANT_NAMES), (sorted(ids.items()), sorted(conv.VARIANT_NAMES))
    assert conv.VARIANT_NAMES[6] == 'gemm_pw_kernel' and 'splitk' in conv.VARIANT_NAMES[7]


def test_pointwise_spec_detection():
    """Only 1x1x1, stride-1, unpadded, unshuffled convolutions are declared as plain GEMMs (GenieConvDesc.pointwise)."""
    from genie import conv
    assert conv._is_pointwise(conv.same_spec(128, 256, (1, 1, 1)))
    assert not conv._is_pointwise(conv.same_spec(128, 256, (3, 3, 3)))
    assert not conv._is_pointwise(conv.causal_spec(128, 256, (1, 1, 1), stride=(1, 2, 2)))
    assert not conv._is_pointwise(conv.causal_spec(64, 64 * 8, (1, 1, 1), shuffle=(2, 2, 2)))
    assert not conv._is_pointwise(conv.causal_spec(64, 64, (2, 1, 1)))          # causal front padding in time


def test_dynamics_compact_row_grid():
    """DynamicsModel.compute_loss lays the gathered masked rows out as a (t, h, 256) grid for the vocabulary head: the grid must hold
    every row, keep each axis under the kernels' 1024 limit, and pad little."""
    from genie.dynamics import DynamicsModel
    for r in (1, 255, 256, 257, 3072, 4096, 131072, 131073, 1 << 20, 50_000_000):
        gt, gh, rp = DynamicsModel._compact_grid(r)
        assert rp == gt * gh * 256 and rp >= r
        assert 1 <= gt < 1024 and 1 <= gh < 1024
        assert rp - r < (256 if r <= 131072 else 512 * 256)


def test_graph_capture_declarations():
    """Which training steps Trainer(graph=True) may record once and replay: the tokenizer without its GAN critic (the critic draws random
    frames per step); Genie only with device-drawn masks (its default draws the MaskGIT mask on the host and gathers a data-dependent number
    of rows).  And the step wrapper refuses host tensors -- there is no CPU path to capture."""
    from genie import Genie, VideoTokenizer
    from genie.graph import GraphedTrainStep
    from genie.trainer import ParamArena
    g = torch.load(os.path.join(GOLD, 'tokenizer_small.pt'), weights_only=False)
    tok = VideoTokenizer(g['enc_desc'], g['dec_desc'], d_codebook=g['d_codebook'], gan_loss_weight=0., perc_loss_weight=0.)
    assert tok.graph_capture_safe
    tok_gan = VideoTokenizer(g['enc_desc'], g['dec_desc'], d_codebook=6, perc_loss_weight=0., disc_kwargs={'inp_size': 16, 'model_dim': 8})
    assert not tok_gan.graph_capture_safe
    lam = (('space-time_attn', {'n_head': 2, 'd_head': 32, 'transpose': True}),)
    dyn = (('space-time_attn', {'n_rep': 1, 'n_head': 2, 'd_head': 32}),)
    for flag in (False, True):
        gen = Genie(tok, enc_desc=lam, dec_desc=lam, d_codebook=4, inp_shape=(16, 16), n_embd=64, dyn_desc=dyn, embed_dim=64, device_masks=flag)
        assert gen.graph_capture_safe is flag
    fresh = VideoTokenizer(g['enc_desc'], g['dec_desc'], d_codebook=g['d_codebook'], gan_loss_weight=0., perc_loss_weight=0.)   # (Genie froze `tok`)
    with pytest.raises(ValueError, match='CUDA'):
        GraphedTrainStep(fresh, ParamArena(fresh), torch.zeros(1, 3, 4, 16, 16))


def test_tri_executed_fraction_matches_the_kernels_trim_rule():
    """Host restatement of tri_trim_range (csrc/conv_igemm3.hip): a 256-row tile inside one frame skips the step-table rows whose source
    frame is time padding.  'same' 3x3x3 conv: 2 of 3 T (frame, dt) pairs; causal: 3; tiles that straddle frames skip nothing."""
    from genie.conv import tri_executed_fraction
    # symmetric padding: dt in {-1, 0, 1} -> dt_min -1; 3 dt x 3 dh x 2 channel blocks = 18 rows, 6 per dt
    assert abs(tri_executed_fraction((6, -1, 18), 16, 32 * 32, 64 * 16 * 1024, 256) - 46 / 48) < 1e-12
    assert abs(tri_executed_fraction((6, -1, 18), 8, 16 * 16, 64 * 8 * 256, 256) - 22 / 24) < 1e-12
    # causal: dt in {-2, -1, 0}
    assert abs(tri_executed_fraction((6, -2, 18), 16, 32 * 32, 4 * 16 * 1024, 256) - 45 / 48) < 1e-12
    # 4 x 8 x 8 frames: a 256-row tile spans four frames -> nothing is trimmed; no trim table -> 1
    assert tri_executed_fraction((6, -1, 18), 4, 64, 64 * 4 * 64, 256) == 1.0
    assert tri_executed_fraction((0, 0, 18), 16, 1024, 16384, 256) == 1.0 and tri_executed_fraction(None, 16, 1024, 16384, 256) == 1.0
    # 128-row tiles on a 16 x 16 frame: two tiles per frame, both inside it
    assert abs(tri_executed_fraction((3, -1, 9), 8, 256, 2 * 8 * 256, 128) - 22 / 24) < 1e-12


def test_bench_launcher_branch(monkeypatch):
    """`python bench.py --gpus N` started without a launcher (this branch had never executed anywhere): it must exec
    `torch.distributed.run` with one rank per GPU on 127.0.0.1 and the caller's own flags, keep the dmabuf-IPC switch in the environment, and
    refuse to run when fewer GPUs are visible than the label says."""
    import bench
    seen = {}

    class Exec(Exception):
        pass

    def fake_execv(path, argv):
        seen['path'], seen['argv'] = path, list(argv)
        raise Exec()
    monkeypatch.setattr(bench.os, 'execv', fake_execv)
    monkeypatch.delenv('WORLD_SIZE', raising=False)
    monkeypatch.delenv('HSA_ENABLE_IPC_MODE_LEGACY', raising=False)
    monkeypatch.setattr(bench.torch.cuda, 'device_count', lambda: 8)
    monkeypatch.setattr(sys, 'argv', ['bench.py', '--gpus', '4', '--steps', '3', '--warmup', '1', '--allreduce', 'rs_ag'])
    with pytest.raises(Exec):
        bench.main()
    a = seen['argv']
    assert seen['path'] == sys.executable and a[:3] == [sys.executable, '-m', 'torch.distributed.run']
    assert '--nnodes=1' in a and '--nproc-per-node=4' in a
    assert a[a.index('--master-addr') + 1] == '127.0.0.1' and 1024 <= int(a[a.index('--master-port') + 1]) < 65536
    script = a.index(os.path.join(ROOT, 'bench.py'))
    assert a[script + 1:] == ['--gpus', '4', '--steps', '3', '--warmup', '1', '--allreduce', 'rs_ag']      # the ranks see the caller's flags
    assert os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY') == '0'
    assert bench.launcher_command(2, ['--gpus', '2'], 1234)[-5:] == ['--master-port', '1234', os.path.join(ROOT, 'bench.py'), '--gpus', '2']
    # fewer GPUs than the label: refuse
    monkeypatch.setattr(bench.torch.cuda, 'device_count', lambda: 1)
    with pytest.raises(SystemExit) as ei:
        bench.main()
    assert 'only 1 GPU' in str(ei.value)
    # a rank started by the launcher with the wrong world size is refused as well (and no GPU here: the product path has no CPU fallback)
    monkeypatch.setenv('WORLD_SIZE', '4')
    with pytest.raises(SystemExit):
        bench.main()


def test_linear_ce_plan_fills_whole_rounds():
    """The fused head + CE runs one workgroup per CU, i.e. in rounds of (number of CUs) workgroups: the split count must not leave a nearly empty
    last round.  Round 5's plan always asked for ~768 workgroups: 192 row tiles x 4 = three rounds, but the 193 row tiles a Bernoulli mask actually
    leaves x 4 = 772 = FOUR (14.2 ms in the step vs 10.97 ms stand-alone).  genie_linear_ce_ws_floats is host arithmetic (256 CUs are
    assumed where no device answers), so the plan is visible here: splits = workspace / (padded rows x (2 + D))."""
    from genie import _hip
    lib = _hip.load_library()
    d, v, cus = 512, 1 << 18, 256

    def splits(m):
        tiles = (m + 127) // 128
        ws = lib.genie_linear_ce_ws_floats(m, d, v, 1)
        assert ws % (tiles * 128 * (2 + d)) == 0
        return tiles, ws // (tiles * 128 * (2 + d))

    for m in (24576, 24616, 24000, 25000, 20000, 12345, 8192, 3072, 1024, 128):
        tiles, ns = splits(m)
        assert 1 <= ns <= 16
        rounds = -(-tiles * ns // cus)
        cost = rounds / ns                      # time in units of one un-split sweep
        ideal = tiles / cus
        # within a quarter of a round-unit of the ideal, and never the 4 / 3 of round 5
        assert cost <= max(ideal * 1.12, ideal + 0.13), (m, tiles, ns, rounds, cost, ideal)
    assert splits(24576) == (192, 4)            # the stand-alone bench's shape keeps its three full rounds
    assert lib.genie_linear_ce_ws_floats(24616, d, v, 0) * (2 + d) == lib.genie_linear_ce_ws_floats(24616, d, v, 1) * 2      # lse-only sweep: no O partials


def test_attention_dropout_argument_contract():
    """Attention(dropout=...) (reference attention.py:166-198): any rate in [0, 1) constructs and is handed down by SpaceTimeAttention, for every head
    width the path supports; a rate outside [0, 1) is refused at construction."""
    from genie.module.attention import SpaceTimeAttention, SpatialAttention
    m = SpaceTimeAttention(n_head=2, d_head=32, dropout=0.25)
    assert m.space_attn.dropout == 0.25 and m.temp_attn.dropout == 0.25 and m.space_attn.last_dropout_seed is None
    assert SpatialAttention(n_head=2, d_head=16).dropout == 0.0
    with pytest.raises(ValueError):
        SpatialAttention(n_head=2, d_head=32, dropout=1.0)
    with pytest.raises(ValueError):
        SpatialAttention(n_head=2, d_head=32, dropout=-0.1)
    assert SpatialAttention(n_head=2, d_head=16, dropout=0.1).dropout == 0.1          # narrow heads too (attention_narrow.hip)
