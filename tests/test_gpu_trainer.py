"""Training runtime on the MI355X: fused AdamW over the parameter arena against torch.optim.AdamW, and the bf16 weight packs
that ride on the optimiser kernel (mirror + one batched transpose) against per-conv packing."""
import pytest
import torch

from util import ROOT  # noqa: F401  (puts the package on sys.path)

pytestmark = pytest.mark.gpu

ENC = (('causal-conv3d', {'in_channels': 3, 'out_channels': 64, 'kernel_size': 3}),
       ('video-residual', {'in_channels': 64}),
       ('spacetime_downsample', {'in_channels': 64, 'out_channels': 64, 'kernel_size': 3, 'time_factor': 2, 'space_factor': 2}),
       ('video-residual', {'in_channels': 64, 'out_channels': 128}),
       ('group_norm', {'num_groups': 8, 'num_channels': 128}), ('silu', {}),
       ('causal-conv3d', {'in_channels': 128, 'out_channels': 10, 'kernel_size': 1}))
DEC = (('causal-conv3d', {'in_channels': 10, 'out_channels': 128, 'kernel_size': 3}),
       ('video-residual', {'in_channels': 128}),
       ('adaptive_group_norm', {'dim_cond': 10, 'num_groups': 8, 'num_channels': 128, 'has_ext': True}),
       ('depth2spacetime_upsample', {'in_channels': 128, 'kernel_size': 3, 'time_factor': 2, 'space_factor': 2}),
       ('video-residual', {'in_channels': 128, 'out_channels': 72}),
       ('group_norm', {'num_groups': 8, 'num_channels': 72}), ('silu', {}),
       ('causal-conv3d', {'in_channels': 72, 'out_channels': 3, 'kernel_size': 3}))


def _model():
    from genie import VideoTokenizer
    torch.manual_seed(0)
    return VideoTokenizer(ENC, DEC, d_codebook=10, gan_loss_weight=0., perc_loss_weight=0.).cuda().train()


def test_arena_adamw_matches_torch():
    """genie_adamw_step over the flat arena == torch.optim.AdamW (reference tokenizer.py:437-442, config lr / weight decay) on
    the same gradients, two steps, including the consume-and-clear of the gradient arena."""
    from genie.trainer import ParamArena
    m = _model()
    arena = ParamArena(m)
    ref_params = [torch.nn.Parameter(p.detach().clone()) for p in m.parameters()]
    opt = torch.optim.AdamW(ref_params, lr=1e-3, weight_decay=0.01)
    x = torch.randn(2, 3, 4, 16, 16, device='cuda')
    for _ in range(2):
        loss, _ = m(x)
        loss.backward()
        for rp, p in zip(ref_params, m.parameters()):
            rp.grad = p.grad.detach().clone()
        arena.adamw_step(lr=1e-3, weight_decay=0.01)
        opt.step()
        assert arena.grads.abs().max().item() == 0.
        for (name, p), rp in zip(m.named_parameters(), ref_params):
            torch.testing.assert_close(p.detach(), rp.detach(), rtol=2e-6, atol=2e-7, msg=name)


def test_weight_packs_follow_the_optimiser():
    """attach_weight_packs: after every optimiser step the managed forward / backward-data packs are bit-identical to what
    per-convolution packing of the fp32 weights gives, the loss trajectory is unchanged (up to atomic-add order), and unmanaged convs still repack."""
    from genie import conv as gconv
    from genie.module.video import Conv3d
    from genie.trainer import ParamArena
    x = torch.randn(2, 3, 4, 16, 16, device='cuda')
    losses = []
    for attach in (False, True):
        m = _model()
        arena = ParamArena(m)
        n = arena.attach_weight_packs(m) if attach else 0
        convs = [c for c in m.modules() if isinstance(c, Conv3d)]
        if attach:
            assert 0 < n < len(convs)                       # stem (3 ch) and the 10-channel latent convs stay unmanaged
        traj = []
        for _ in range(3):
            loss, _ = m(x)
            loss.backward()
            arena.adamw_step(lr=1e-3, weight_decay=0.01)
            traj.append(loss.item())
            if attach:
                managed = {id(op) for op, *_ in arena._packs['managed']}
                for c in convs:
                    if id(c.op) not in managed:
                        continue
                    key = (c.weight._version, c.weight.data_ptr())
                    assert c.op._fwd[0] == key and c.op._bwd[0] == key
                    assert torch.equal(c.op._fwd[1].reshape(-1), gconv.pack_weight_fwd(c.weight, c.spec).reshape(-1)), c
                    assert torch.equal(c.op._bwd[1].reshape(-1), gconv.pack_weight_bwd(c.weight, c.spec).reshape(-1)), c
        losses.append(traj)
    # identical packs -> identical arithmetic up to the order of the wgrad kernels' fp32 atomics; after two optimiser steps that
    # ulp-level noise has gone through the LFQ entropy term (slope ~4 beta = 400 at the decision boundary), hence 1e-3
    for a, b in zip(*losses):
        assert abs(a - b) <= 1e-3 * abs(a), losses
    assert losses[0][0] == losses[1][0] and abs(losses[0][1] - losses[1][1]) <= 1e-5 * abs(losses[0][1]), losses


PROJ_ENC = ENC[:-1] + (('causal-conv3d', {'in_channels': 128, 'out_channels': 16, 'kernel_size': 1}),)
PROJ_DEC = (('causal-conv3d', {'in_channels': 16, 'out_channels': 128, 'kernel_size': 3}),) + DEC[1:2] + \
           (('adaptive_group_norm', {'dim_cond': 16, 'num_groups': 8, 'num_channels': 128, 'has_ext': True}),) + DEC[3:]


def _step_grads(build, run, dp_cuts, loopback):
    """One forward/backward of a freshly built model; returns (gradient arena copy, DataParallel or None, arena)."""
    from genie.trainer import DataParallel, ParamArena
    m = build()
    arena = ParamArena(m)
    m._arena_for_cuts = arena
    dp = None
    if loopback:
        dp = DataParallel(arena.grads, compress='bf16', loopback=True)
        dp.install_overlap_hooks(arena, m, dp_cuts(m))
        dp.trace = True                                      # HIP events around every bucket's all-reduce and finish()'s wait (comm_report)
    loss = run(m)
    loss.backward()
    if dp is not None:
        fired_in_backward = list(dp.fired)
        dp.finish()
        dp.fired_in_backward = fired_in_backward
    torch.cuda.synchronize()
    return arena.grads.clone(), dp, arena


def test_data_parallel_loopback_on_real_models():
    """The RCCL side-stream path on ONE GPU (a single-rank nccl group, loopback=True): arena laid out in execution order, overlap
    hooks on real models whose registration order differs from execution order (a PROJECTING LFQ -- quant.proj_inp / proj_out --
    and a DynamicsModel with embeddings + head), buckets reduced on the comm stream during backward.  With bf16 compression a
    reduced bucket is rounded to bf16 in place, so a gradient contribution that arrived AFTER its bucket had been reduced (the
    failure ADVICE.md round 1 describes) would leave values that are not bf16-representable: every gradient must be, and must
    match the run without data parallelism."""
    import torch.distributed as dist
    from genie import VideoTokenizer
    from genie.dynamics import DynamicsModel
    if not dist.is_initialized():
        dist.init_process_group('nccl', init_method='tcp://127.0.0.1:29611', rank=0, world_size=1, device_id=torch.device('cuda', 0))
    try:
        x = torch.randn(2, 3, 4, 16, 16, device='cuda')

        def build_tok():
            torch.manual_seed(0)
            m = VideoTokenizer(PROJ_ENC, PROJ_DEC, d_codebook=10, gan_loss_weight=0., perc_loss_weight=0.).cuda().train()
            assert isinstance(m.quant.proj_inp, torch.nn.Linear)
            return m

        desc = (('space-time_attn', {'n_rep': 3, 'n_head': 2, 'd_head': 32}),)
        tok, act = torch.randint(0, 256, (2, 4, 4, 4), device='cuda'), torch.randint(0, 5, (2, 4), device='cuda')
        mask = (torch.rand(2, 4, 4, 4, device='cuda') < 0.7)

        def build_dyn():
            torch.manual_seed(1)
            return DynamicsModel(desc, tok_vocab=256, act_vocab=5, embed_dim=64).cuda().train()

        cases = [
            (build_tok, lambda m: m(x)[0], lambda m: [m.enc_layers[2], m.quant, m.dec_layers[1], m.dec_layers[4]]),
            (build_dyn, lambda m: m.compute_loss(tok, act, mask=mask), lambda m: [m.dec_layers[1], m.dec_layers[2], m.head]),
        ]
        # equal-byte cuts (what bench.py uses): chosen among the layers, in forward order, close to the k / n points of the arena
        from genie.trainer import DataParallel, ParamArena
        m0 = build_tok()
        a0 = ParamArena(m0)
        layers = [l for l in list(m0.enc_layers) + list(m0.dec_layers) if any(p.requires_grad for p in l.parameters())]
        picks = DataParallel.equal_byte_cuts(a0, m0, layers, 4)
        offs = [a0.offset_of(l, m0) for l in picks]
        assert offs == sorted(offs) and 1 <= len(picks) <= 4 and all(0 < o < a0.numel for o in offs)
        assert offs[0] == min(o for o in (a0.offset_of(l, m0) for l in layers) if o)       # the small first bucket: whatever precedes the second layer
        assert all(min(abs(o - a0.numel * k / 4) for k in (1, 2, 3)) < a0.numel / 4 for o in offs[1:])
        del m0, a0
        cases.append((build_tok, lambda m: m(x)[0],
                      lambda m: DataParallel.equal_byte_cuts(m._arena_for_cuts, m, [l for l in list(m.enc_layers) + list(m.dec_layers)
                                                                                   if any(p.requires_grad for p in l.parameters())], 4)))
        for build, run, cuts in cases:
            g_ref, _, _ = _step_grads(build, run, cuts, loopback=False)
            g_dp, dp, arena = _step_grads(build, run, cuts, loopback=True)
            nb = len(dp.buckets)
            assert dp.last_fired == list(range(nb - 1, -1, -1)), dp.last_fired
            assert len(dp.fired_in_backward) >= nb - 1, (dp.fired_in_backward, nb)        # all but the first bucket start during backward
            assert dp.bytes_reduced == arena.numel * 2
            # the overlap diagnostics bench.py --gpus N prints (`comm`): one entry per bucket, payloads add up, times are sane
            rep = dp.comm_report()
            assert rep['steps_traced'] == 1 and len(rep['buckets']) == nb
            assert sum(b_['elements'] for b_ in rep['buckets']) == arena.numel
            assert all(b_['allreduce_ms'] > 0 and b_['issued_before_backward_end_ms'] >= 0 for b_ in rep['buckets'])
            assert rep['exposed_ms_per_step'] >= 0 and rep['allreduce_ms_per_step'] > 0
            assert torch.equal(g_dp, g_dp.to(torch.bfloat16).float()), 'a gradient was written after its bucket had been reduced'
            for name, (off, n) in arena.slots.items():
                a, b = g_dp[off:off + n], g_ref[off:off + n]
                assert b.abs().max() > 0 or 'bias' in name or 'freq' in name, f'{name}: no gradient'
                err = (a - b).abs().max().item()
                assert err <= 2 ** -7 * b.abs().max().item() + 1e-6, (name, err, b.abs().max().item())
    finally:
        dist.destroy_process_group()


def test_async_wgrad_side_stream_matches_in_order_execution():
    """functional.ASYNC_WGRAD: weight-gradient kernels issued on a side stream (concurrent with the rest of backward) give the same
    gradients and the same parameters after the optimiser step as in-order execution -- the joins in ParamArena / DataParallel hold."""
    from genie import functional as GF
    from genie.trainer import ParamArena
    x = torch.randn(2, 3, 4, 16, 16, device='cuda')
    res = []
    for flag in (0, 1, 2):
        GF.ASYNC_WGRAD = flag
        try:
            m = _model()
            arena = ParamArena(m)
            arena.attach_weight_packs(m)
            g = None
            for it in range(2):
                loss, _ = m(x)
                loss.backward()
                GF.join_wgrad()
                if it == 0:
                    g, l_first = arena.grads.clone(), loss.item()      # same parameters in every mode: the gradients must agree
                arena.adamw_step(lr=1e-3, weight_decay=0.01)
            torch.cuda.synchronize()
            res.append((g, arena.params.clone(), l_first))
        finally:
            GF.ASYNC_WGRAD = 0
    (g0, p0, l0) = res[0]
    for g1, p1, l1 in res[1:]:
        assert abs(l0 - l1) <= 1e-3 * abs(l0)
        assert (g0 - g1).abs().max().item() <= 2e-3 * g0.abs().max().item() + 1e-6          # fp32 atomics order only
        # AdamW's first steps move every parameter by ~lr * sign(grad): where the gradient is rounding noise the atomics order can flip
        # the sign, so the bound on a single parameter is 2 * lr per step; the bulk must agree far better than that
        dp = (p0 - p1).abs()
        assert dp.max().item() <= 2 * 2 * 1e-3 + 5e-4, dp.max().item()
        assert dp.mean().item() <= 2e-5, dp.mean().item()


def test_checkpoint_resume_continues_the_run(tmp_path):
    """last.ckpt carries the weights, AdamW's moments / step counter and the position in the epoch (the reference's
    ModelCheckpoint(save_last) is resumable): 2 steps + resume + 2 steps == 4 steps straight, up to the atomic-add order of the weight
    gradients."""
    from genie.dataset import LightningSynthetic
    from genie.trainer import Trainer

    def data():
        return LightningSynthetic(num_clips=16, shape=(3, 4, 16, 16), seed=3, batch_size=2, num_workers=0, train_shuffle=False)

    straight = _model()
    Trainer(max_steps=4, default_root_dir=str(tmp_path / 'a'), log_every_n_steps=100).fit(straight, data())
    first = _model()
    Trainer(max_steps=2, default_root_dir=str(tmp_path / 'b'), log_every_n_steps=100).fit(first, data())
    ck = torch.load(str(tmp_path / 'b' / 'last.ckpt'), map_location='cpu')
    assert ck['global_step'] == 2 and ck['loops']['batches_done_in_epoch'] == 2
    st = ck['optimizer_states'][0]['state']
    assert set(st) == {n for n, p in first.named_parameters() if p.requires_grad}
    assert all(v['step'] == 2 and v['exp_avg'].abs().sum() >= 0 for v in st.values())
    resumed = _model()
    with torch.no_grad():
        for p in resumed.parameters():
            p.add_(1.0)                                       # whatever the fresh model holds must be overwritten by the checkpoint
    tr = Trainer(max_steps=4, default_root_dir=str(tmp_path / 'c'), log_every_n_steps=100).fit(resumed, data(), ckpt_path=str(tmp_path / 'b' / 'last.ckpt'))
    assert tr.global_step == 4 and tr.arena.step_count == 4
    for (n, a), (_, b) in zip(straight.named_parameters(), resumed.named_parameters()):
        torch.testing.assert_close(a.detach(), b.detach(), rtol=1e-4, atol=1e-5, msg=n)
    # the checkpoint's moments matter: zero them and the continuation leaves the straight run
    ck2 = torch.load(str(tmp_path / 'b' / 'last.ckpt'), map_location='cpu')
    for v in ck2['optimizer_states'][0]['state'].values():
        v['exp_avg'].zero_(); v['exp_avg_sq'].zero_()
    torch.save(ck2, str(tmp_path / 'b' / 'no_moments.ckpt'))
    cold = _model()
    Trainer(max_steps=4, default_root_dir=str(tmp_path / 'd'), log_every_n_steps=100).fit(cold, data(), ckpt_path=str(tmp_path / 'b' / 'no_moments.ckpt'))
    diff = max((a.detach() - b.detach()).abs().max().item() for a, b in zip(straight.parameters(), cold.parameters()))
    assert diff > 1e-4, diff
