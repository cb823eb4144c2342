"""A training step captured in a hipGraph (genie/graph.py): replays must do what the eager step does -- same losses, same parameters --
with the step count and learning rate of AdamW living in device memory (genie_adamw_step_graph)."""
import pytest
import torch

from util import ROOT, report  # noqa: F401  (puts the package on sys.path)
from test_gpu_trainer import _model

pytestmark = pytest.mark.gpu


def _batches(n):
    g = torch.Generator(device='cuda').manual_seed(7)
    return [torch.randn(2, 3, 4, 16, 16, device='cuda', generator=g) for _ in range(n)]


def test_graph_safe_adamw_equals_the_scalar_argument_form():
    """genie_adamw_step_graph (coefficients computed on the device from a device-side step counter) == genie_adamw_step_mirror
    (coefficients computed on the host) on IDENTICAL gradients, three steps.  (Through a model the comparison is meaningless: Adam's
    first updates are lr * sign(g), so a 1-ulp difference in the step size flips the sign of the next step's near-zero gradients.)"""
    from genie.trainer import ParamArena
    gen = torch.Generator(device='cuda').manual_seed(3)
    grads = None
    params, mirrors = [], []
    for graph_safe in (False, True):
        m = _model()
        arena = ParamArena(m)
        arena.attach_weight_packs(m)
        if grads is None:
            grads = [torch.randn(arena.numel, device='cuda', generator=gen) * (10.0 ** -i) for i in range(3)]
        if graph_safe:
            arena.set_graph_hyperparameters(2e-3, 0.05)
        for g in grads:
            arena.grads.copy_(g)
            arena.adamw_step(lr=2e-3, weight_decay=0.05, graph_safe=graph_safe)
            assert arena.grads.abs().max().item() == 0.
        params.append(arena.params.clone())
        mirrors.append(arena.mirror.clone())
        if graph_safe:
            assert int(arena._opt_state[:1].view(torch.int32).item()) == 3 == arena.step_count
    diff = (params[0] - params[1]).abs().max().item()
    scale = params[0].abs().max().item()
    report('graph_safe_adamw', max_abs_diff=diff, max_abs_param=scale)
    assert diff <= 1e-6 * max(1.0, scale), diff          # the step size is rounded once on the host in one form, once on the device in the other
    assert (mirrors[0].float() - mirrors[1].float()).abs().max().item() <= 2 ** -7 * scale


def test_graphed_train_step_replays_the_eager_step():
    """GraphedTrainStep (2 eager warm-up steps + 1 captured step on the example, then replays on new batches) against the eager loop
    over the same batch sequence: loss of every replayed step and the final parameters."""
    from genie import conv as gconv
    from genie.graph import GraphedTrainStep
    from genie.trainer import ParamArena
    xs = _batches(4)
    seq = [xs[0], xs[0], xs[0], xs[1], xs[2], xs[3], xs[1]]          # what the graphed run sees: 3 x example, then 4 replays
    old = gconv.set_deterministic(True)                               # single-owner weight gradients: the two runs are comparable bit for bit
    try:
        m = _model()
        arena = ParamArena(m)
        arena.attach_weight_packs(m)
        arena.set_graph_hyperparameters(1e-3, 0.01)
        eager_losses = []
        for x in seq:
            loss, _ = m(x)
            loss.backward()
            arena.adamw_step(graph_safe=True)                          # the same kernel form as the captured step
            eager_losses.append(loss.item())
        eager_params = arena.params.clone()

        m2 = _model()
        arena2 = ParamArena(m2)
        arena2.attach_weight_packs(m2)
        gs = GraphedTrainStep(m2, arena2, xs[0], lr=1e-3, weight_decay=0.01, warmup=2)
        assert gs.steps_done == 3 and arena2.step_count == 3
        graph_losses = [gs.loss.item()]
        for x in seq[3:]:
            graph_losses.append(gs(x).item())
        assert arena2.step_count == len(seq)
        # eager use after replays sees current weights (on-demand packs rebuilt, managed packs re-keyed)
        with torch.no_grad():
            la, lb = m(xs[2])[0].item(), m2(xs[2])[0].item()
    finally:
        gconv.set_deterministic(old)
    dl = max(abs(a - b) for a, b in zip(eager_losses[2:], graph_losses))
    dp = (eager_params - arena2.params).abs().max().item()
    rel = ((eager_params - arena2.params).norm() / eager_params.norm()).item()
    report('graphed_train_step', max_loss_diff=dl, max_param_diff=dp, rel_l2_param_diff=rel, eval_loss_diff=abs(la - lb), losses=len(graph_losses))
    # identical kernels in identical order on identical data: expected bit-identical; the bounds leave room for one atomic-order
    # difference in a near-zero gradient (Adam turns that into an O(lr) step of that one element)
    assert dl <= 1e-5 * max(1.0, abs(eager_losses[-1])), (eager_losses, graph_losses)
    assert rel <= 1e-5 and dp <= 1e-2, (rel, dp)
    assert abs(la - lb) <= 1e-4 * max(1.0, abs(la))


def test_graphed_step_learning_rate_is_a_device_value():
    """set_lr between replays changes the update without a re-capture: lr = 0 leaves the parameters untouched (weight decay 0)."""
    from genie.graph import GraphedTrainStep
    from genie.trainer import ParamArena
    xs = _batches(2)
    m = _model()
    arena = ParamArena(m)
    arena.attach_weight_packs(m)
    gs = GraphedTrainStep(m, arena, xs[0], lr=1e-3, weight_decay=0.0, warmup=1)
    before = arena.params.clone()
    gs.set_lr(0.0)
    gs(xs[1])
    assert torch.equal(before, arena.params)
    gs.set_lr(1e-3)
    gs(xs[1])
    assert not torch.equal(before, arena.params)


def test_trainer_fit_with_graph_matches_eager_fit(tmp_path):
    """Trainer(graph=True): two eager steps, the third captured, the rest replayed -- the same run as Trainer(graph=False) on the same
    data (logged losses, final parameters, step counters, a resumable last.ckpt)."""
    from genie import conv as gconv
    from genie.dataset import LightningSynthetic
    from genie.trainer import Trainer

    def data():
        return LightningSynthetic(num_clips=16, shape=(3, 4, 16, 16), seed=3, batch_size=2, num_workers=0, train_shuffle=False)

    old = gconv.set_deterministic(True)
    try:
        runs = []
        for graph in (False, True):
            m = _model()
            tr = Trainer(max_steps=6, default_root_dir=str(tmp_path / f'g{int(graph)}'), log_every_n_steps=1, graph=graph, device_state_adamw=True).fit(m, data())
            assert tr.global_step == 6 and tr.arena.step_count == 6
            runs.append((tr, m))
    finally:
        gconv.set_deterministic(old)
    (ta, ma), (tb, mb) = runs
    la = [r['train_loss'] for r in ta.history if r['split'] == 'train']
    lb = [r['train_loss'] for r in tb.history if r['split'] == 'train']
    assert len(la) == len(lb) >= 6
    dl = max(abs(a - b) for a, b in zip(la, lb))
    rel = ((ta.arena.params - tb.arena.params).norm() / ta.arena.params.norm()).item()
    report('trainer_graph_fit', max_loss_diff=dl, rel_l2_param_diff=rel, steps=6)
    # Both runs use the SAME AdamW form (device_state_adamw: hyper-parameters read from device memory) and deterministic weight gradients,
    # so a replayed step is the eager step's arithmetic: all six steps are checked tightly (the 5e-3 bound of round 5 -- needed
    # when the eager run used the scalar-argument AdamW and an LFQ sign flipped at step 4 -- would have let a diverging replay pass).
    assert dl <= 1e-6 * max(1.0, abs(la[-1])), (la, lb)
    assert rel <= 1e-6, rel
    ck = torch.load(str(tmp_path / 'g1' / 'last.ckpt'), map_location='cpu')
    assert ck['global_step'] == 6 and all(v['step'] == 6 for v in ck['optimizer_states'][0]['state'].values())
    # a validation pass between replays overwrites model._last_logged and a replay never re-enters Python to refresh it: 'train' lines must
    # still report TRAINING metrics (the captured tensors), the final one included
    m = _model()
    tv = Trainer(max_steps=6, default_root_dir=str(tmp_path / 'gv'), log_every_n_steps=1, graph=True, val_check_interval=4,
                 limit_val_batches=1).fit(m, data())
    recs = [r for r in tv.history if r['split'] == 'train']
    assert len(recs) >= 7 and all('train_loss' in r and not any(k.startswith('val') for k in r if k not in ('split',)) for r in recs), tv.history
    assert any(r['split'] == 'val' and 'val_loss' in r for r in tv.history)
    # a model whose step draws per-step randomness outside the graph's reach is refused, not silently replayed with one draw
    m = _model()
    m.gan_loss_weight = 1.0
    assert not m.graph_capture_safe
    with pytest.raises(ValueError, match='graph'):
        Trainer(max_steps=2, default_root_dir=str(tmp_path / 'g2'), graph=True).fit(m, data())


def test_dynamics_fixed_rows_loss_and_graph_replay():
    """DynamicsModel.compute_loss(fixed_rows=True) -- masked rows sorted to the front, the head over all rows, the others switched off in
    the cross-entropy -- gives the loss and the gradients of the gathered-rows form on the same mask (also for an all-False and an all-True
    mask), and with it the Dynamics step replays as a hipGraph: token / action / mask buffers as inputs, losses equal to the eager loop."""
    from genie.dynamics import DynamicsModel
    from genie.graph import GraphedTrainStep
    from genie.trainer import ParamArena
    desc = (('space-time_attn', {'n_rep': 2, 'n_head': 2, 'd_head': 32}),)

    def build():
        torch.manual_seed(11)
        return DynamicsModel(desc, tok_vocab=256, act_vocab=5, embed_dim=64).cuda().train()

    g = torch.Generator(device='cuda').manual_seed(3)
    data = [(torch.randint(0, 256, (2, 5, 4, 4), device='cuda', generator=g), torch.randint(0, 5, (2, 5), device='cuda', generator=g),
             torch.rand(2, 5, 4, 4, device='cuda', generator=g) < 0.7) for _ in range(5)]
    # the two forms of the loss on one mask
    m = build()
    tok, act, mask = data[0]
    la = m.compute_loss(tok, act, mask=mask)
    la.backward()
    ga = {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None}
    for p in m.parameters():
        p.grad = None
    lb = m.compute_loss(tok, act, mask=mask, fixed_rows=True)
    lb.backward()
    assert abs(la.item() - lb.item()) <= 1e-6 * abs(la.item())
    for n, p in m.named_parameters():
        if n in ga:
            err = (p.grad - ga[n]).norm().item() / (ga[n].norm().item() + 1e-12)
            assert err < 1e-3, (n, err)
    full = m.compute_loss(tok, act, mask=torch.ones_like(mask), fixed_rows=True)
    assert abs(full.item() - m.compute_loss(tok, act, mask=torch.ones_like(mask)).item()) <= 1e-6 * abs(full.item())
    assert torch.isnan(m.compute_loss(tok, act, mask=torch.zeros_like(mask), fixed_rows=True))
    # graph replay against the eager loop (both with the shape-stable loss and the device-state AdamW)
    loss_fn = lambda mod, b: mod.compute_loss(b[0], b[1], mask=b[2], fixed_rows=True)
    seq = [data[0]] * 3 + data[1:]
    me = build()
    ae = ParamArena(me); ae.attach_weight_packs(me); ae.set_graph_hyperparameters(1e-3, 0.01)
    eager = []
    for b in seq:
        l = loss_fn(me, b); l.backward(); ae.adamw_step(graph_safe=True); eager.append(l.item())
    mg = build()
    ag = ParamArena(mg); ag.attach_weight_packs(mg)
    gs = GraphedTrainStep(mg, ag, data[0], loss_fn=loss_fn, lr=1e-3, weight_decay=0.01, warmup=2)
    replay = [gs.loss.item()] + [gs(*b).item() for b in data[1:]]
    dl = max(abs(a - b) for a, b in zip(eager[2:], replay))
    rel = ((ae.params - ag.params).norm() / ae.params.norm()).item()
    report('dynamics_graph_replay', max_loss_diff=dl, rel_l2_param_diff=rel, steps=len(replay))
    assert dl <= 1e-4 * abs(eager[-1]) and rel <= 1e-4, (eager, replay, rel)


def test_genie_step_with_device_masks_replays_and_redraws():
    """Genie(device_masks=True): the training step (frozen tokenizer -> latent actions -> MaskGIT dynamics loss on a device-drawn mask) is
    shape-stable and replays as a hipGraph; every replay draws a NEW mask (with the learning rate at 0 the loss still changes from replay
    to replay), and Trainer(graph=True) accepts the model."""
    from genie.graph import GraphedTrainStep
    from genie.trainer import ParamArena
    from test_gpu_genie import _genie
    g = _genie()
    g.device_masks = True
    g = g.cuda().train()
    g.tokenizer.eval()
    assert g.graph_capture_safe
    video = torch.rand(2, 3, 8, 16, 16, device='cuda')
    arena = ParamArena(g)
    arena.attach_weight_packs(g)
    gs = GraphedTrainStep(g, arena, video, loss_fn=lambda m, b: m.compute_loss(b)[0], lr=1e-3, weight_decay=0.0, warmup=2)
    assert torch.isfinite(gs.loss).item()
    gs.set_lr(0.0)
    before = arena.params.clone()
    losses = [gs(video).item() for _ in range(4)]
    assert torch.equal(before, arena.params)
    assert all(l == l for l in losses) and len(set(losses)) > 1, losses          # same weights, same clip: only the mask can move the loss
    report('genie_graph_device_masks', losses_at_lr0=[round(l, 5) for l in losses])
