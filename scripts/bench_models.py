#!/usr/bin/env python
"""Training-step timing of the other BASELINE configs on one MI355X (configs[2] LatentAction, configs[3] DynamicsModel, configs[4] full
Genie on 32x128x128 clips, and the repaired yaml / REPR tokenizer): ms per step and clips (or latent frames) per second, synthetic inputs
resident in HBM.  Each line also carries

  kernels      -- every conv / GEMM / attention kernel family of the step (HIP events around each launch): launches, ms per step, share,
                  achieved TFLOP/s (or GB/s for the traffic-bound attention shapes) and fraction of the peak
  roofline     -- the family with the largest share of the step (attention included), priced like bench.py's
  cpu_baseline -- (`--cpu-baseline`, configs[2] / [3]) the same training step on the host cores, ONE step after one warm-up: the reference's own
                  modules where /root/reference exists and the model can be built from them (kind "reference"), else the oracle (kind "port")

  python scripts/bench_models.py [lam] [dyn] [repr] [genie4] [smallbatch] [--cpu-baseline] [--lam-batch=16] [--dyn-batch=32] [--genie-batch=4]
Batches (round 5): LatentAction 16 clips, DynamicsModel 32 token grids per step -- chip-filling; rounds 1-4 ran 2 / 4 (the vocabulary head's
logits were materialised then: 0.5 GB per grid and 2^31 elements at most, i.e. 8 grids).
(`smallbatch`: the tokenizer step at 4 / 8 clips and the LatentAction step, eager launches vs one hipGraph replay)
The per-kernel rocprofv3 tables of the same runs: scripts/profile_models.sh -> profiles/rNN_models_*_kernel_stats.csv."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'open-genie_amd')]
import torch

from genie import (LATENT_ACT_DEC, LATENT_ACT_ENC, MAGVIT2_DEC_DESC, MAGVIT2_ENC_DESC, REPR_TOK_DEC, REPR_TOK_ENC, DynamicsModel, Genie, LatentAction,
                   VideoTokenizer)
from genie import conv as gconv
from genie.trainer import ParamArena

PEAK, PEAK_HBM = 2500.0, 8000.0


def run(name, model, step_fn, units, steps=4, warm=2, cpu=None, quiet=False):
    arena = ParamArena(model)
    arena.attach_weight_packs(model)
    for _ in range(warm):
        step_fn().backward(); arena.adamw_step()
    torch.cuda.synchronize()
    # every step timed on its own (one host synchronisation per step: these steps are 40-800 ms): `ms_per_step` is the MEDIAN, so that one
    # allocator / first-touch hiccup inside a three-step window does not become the number; the mean rides along
    per = []
    for _ in range(steps):
        t0 = time.perf_counter()
        step_fn().backward(); arena.adamw_step()
        torch.cuda.synchronize()
        per.append((time.perf_counter() - t0) * 1e3)
    ms = sorted(per)[len(per) // 2]
    out = {'model': name, 'ms_per_step': round(ms, 2), 'ms_per_step_mean': round(sum(per) / len(per), 2), 'ms_per_step_all': [round(v, 2) for v in per],
           'units_per_s': round(units / ms * 1e3, 1), 'params': sum(p.numel() for p in model.parameters()),
           'peak_mem_GB': round(torch.cuda.max_memory_allocated() / 2 ** 30, 2)}
    # second pass with HIP events around every conv / GEMM launch (costs a few % of the step: kept out of the timing above)
    prof = gconv.PROFILER = gconv.LaunchProfiler(only_triple=False)
    for _ in range(steps):
        step_fn().backward(); arena.adamw_step()
    gconv.PROFILER = None
    summ = prof.summary()
    if summ:
        def price(v):
            """(bound, achieved, peak, unit) of one kernel family: attention on short sequences is traffic (its `[hbm]` name), the rest MFMA."""
            if v['ms'] <= 0:
                return ('mfma', None, PEAK, 'TFLOP/s')
            if v.get('hbm'):
                return ('hbm', v['bytes'] / (v['ms'] * 1e-3) / 1e9, PEAK_HBM, 'GB/s')
            return ('mfma', v['flops'] / (v['ms'] * 1e-3) / 1e12, PEAK, 'TFLOP/s')
        for k, v in summ.items():
            v['hbm'] = k.endswith('[hbm]')
        out['kernels'] = {}
        for k, v in sorted(summ.items(), key=lambda kv: -kv[1]['ms']):
            bound, ach, peak, unit = price(v)
            out['kernels'][k] = {'launches_per_step': v['launches'] // steps, 'ms_per_step': round(v['ms'] / steps, 3), 'share_of_step_time': round(v['ms'] / steps / ms, 4),
                                 'bound': bound, 'achieved': round(ach, 1) if ach is not None else None, 'unit': unit, 'frac': round(ach / peak, 4) if ach is not None else None}
            if bound == 'mfma' and v['ms'] > 0 and abs(v.get('flops_exec', v['flops']) - v['flops']) > 1e-6 * v['flops']:
                # executed FLOPs differ from the algorithmic count: zero-frame skipping in the kw-triple convs (fewer), score recomputation in the fused head (more)
                out['kernels'][k]['frac_executed'] = round(v['flops_exec'] / (v['ms'] * 1e-3) / 1e12 / peak, 4)
        # `roofline` = the kernel family with the largest share of the step, attention included (configs[2] / [4] are
        # attention-dominated; round 3 reported a conv variant with 8-12 % of the step there)
        dom = max(summ, key=lambda k: summ[k]['ms'])
        d = summ[dom]
        bound, ach, peak, unit = price(d)
        timed = sum(v['ms'] for v in summ.values()) / steps
        out['roofline'] = {'kernel': dom, 'bound': bound, 'achieved': round(ach, 1), 'peak': peak, 'unit': unit, 'frac': round(ach / peak, 4),
                           **({'frac_executed': out['kernels'][dom]['frac_executed']} if 'frac_executed' in out['kernels'][dom] else {}),
                           'launches_per_step': d['launches'] // steps, 'share_of_step_time': round(d['ms'] / steps / ms, 4),
                           'timed_kernels_share_of_step_time': round(timed / ms, 4),
                           'note': 'HIP events around every conv / GEMM / attention launch of the step (attention: dense 4 S Sk C forward, 2.5 x that backward; '
                                   '[hbm] families priced by their tensor passes); GroupNorm, LFQ and element-wise kernels are in the rocprofv3 table of the same run'}
    if cpu is not None:
        out['cpu_baseline'] = cpu()
    if not quiet:
        print(json.dumps(out), flush=True)
    del arena
    torch.cuda.empty_cache(); torch.cuda.reset_peak_memory_stats()
    return out


def cpu_step(build, units, what, build_ref=None):
    """One training step on the host cores (fp32 torch CPU, every thread this process may use) after one warm-up step: the REAL reference
    modules when /root/reference exists and the caller knows how to build the step from them (`build_ref`; kind = "reference"), else the
    oracle restatement (kind = "port": the GPU boxes have no copy of the reference)."""
    def go():
        import bench as B_
        threads = min(B_.effective_cpus(), 128)
        torch.set_num_threads(threads)
        kind, step = 'port', None
        if build_ref is not None and os.path.isdir('/root/reference'):
            try:
                step, kind = build_ref(), 'reference'
            except Exception as ex:                       # e.g. a reference model that cannot be constructed at HEAD
                print(f'cpu_baseline: reference step unavailable ({type(ex).__name__}: {ex}); using the oracle restatement', flush=True)
        if step is None:
            step = build()
        step()
        t0 = time.perf_counter()
        step()
        dt = time.perf_counter() - t0
        return {'value': round(units / dt, 3), 'unit': what, 'cores': threads, 'kind': kind,
                'sample': f'one training step (fwd + bwd, no optimiser) of the {"reference modules" if kind == "reference" else "oracle restatement"} after one warm-up step, {dt:.1f} s'}
    return go


def bench_lam(B, want_cpu=False, quiet=False, steps=4):
    """BASELINE configs[2]: LatentAction (R-lam, n_embd 256, 8-action codebook) training on B clips of 16x64x64."""
    lam = LatentAction(LATENT_ACT_ENC, LATENT_ACT_DEC, d_codebook=8, inp_channels=3, inp_shape=(64, 64), n_embd=256).cuda().train()
    v = torch.randn(B, 3, 16, 64, 64, device='cuda')
    def lam_cpu():
        from oracle import genie_oracle as O
        sd = {k: (t.detach().float().cpu().clone().requires_grad_(t.is_floating_point() and 'freq' not in k)) for k, t in lam.state_dict().items()}
        x = torch.randn(1, 3, 16, 64, 64)
        return lambda: O.latent_action_forward(x, sd, LATENT_ACT_ENC, LATENT_ACT_DEC, 8, training=True)[1].backward()
    return run(f'LatentAction (configs[2]: R-lam, n_embd 256, 16x64x64, B={B}) [frames/s]', lam, lambda: lam(v)[1], B * 16, steps=steps,
               cpu=cpu_step(lam_cpu, 16, 'video-frames/sec (one 16x64x64 clip)') if want_cpu else None, quiet=quiet)


def bench_dyn(B, want_cpu=False, quiet=False, steps=4):
    """BASELINE configs[3]: DynamicsModel (8 x ST(8x64), V = 2^18) MaskGIT training step on B (16, 8, 8) token grids, mask rate 0.75."""
    desc = (('space-time_attn', {'n_rep': 8, 'n_head': 8, 'd_head': 64}),)
    dyn = DynamicsModel(desc, tok_vocab=2 ** 18, act_vocab=8, embed_dim=512).cuda().train()
    tok = torch.randint(0, 2 ** 18, (B, 16, 8, 8), device='cuda'); act = torch.randint(0, 8, (B, 16), device='cuda')
    g = torch.Generator().manual_seed(1)
    mask = (torch.rand(B, 16, 8, 8, generator=g) < 0.75)
    def dyn_cpu():
        from oracle import genie_oracle as O
        sd = {k: (t.detach().float().cpu().clone().requires_grad_(t.is_floating_point() and 'freq' not in k)) for k, t in dyn.state_dict().items()}
        tk, ac, mk = tok[:2].cpu(), act[:2].cpu(), mask[:2]
        return lambda: O.dynamics_loss(tk, ac, mk, sd, desc).backward()
    def dyn_ref():
        # the reference's own DynamicsModel (genie/dynamics.py:14-99) with our weights; its compute_loss draws its own mask (same rate on average)
        from oracle.ref_import import import_reference
        ref = import_reference()
        rm = ref.DynamicsModel(desc, tok_vocab=2 ** 18, act_vocab=8, embed_dim=512)
        rm.load_state_dict({k: t.detach().float().cpu() for k, t in dyn.state_dict().items()}, strict=False)
        tk, ac = tok[:2].cpu(), act[:2].cpu()
        return lambda: rm.compute_loss(tk, ac).backward()
    return run(f'DynamicsModel (configs[3]: 8 x ST(8x64), V=2^18, (16,8,8) tokens, B={B}) [latent frames/s]', dyn,
               lambda: dyn.compute_loss(tok, act, mask=mask), B * 16, steps=steps,
               cpu=cpu_step(dyn_cpu, 2 * 16, 'latent frames/sec (two (16,8,8) token grids)', build_ref=dyn_ref) if want_cpu else None, quiet=quiet)


def bench_genie4(B, want_cpu=False, quiet=False, steps=3):
    """BASELINE configs[4]: full Genie on 32x128x128 clips -- frozen MAGVIT2 tokenizer (8x16x16 tokens of 2^18 codes), latent-action model
    over 32 frames of 128x128 pixels (spatial attention over S = 16384 positions), MaskGIT dynamics.  B >= 2 (a batch of one dies in the
    reference's mask.squeeze() indexing, dynamics.py:89-97, and so does it here); 16 GB of HBM per clip: B = 4 is 64 GB, as much as the headline's step."""
    tokz = VideoTokenizer(MAGVIT2_ENC_DESC, MAGVIT2_DEC_DESC, d_codebook=18, gan_loss_weight=0., perc_loss_weight=0.).cuda().eval()
    gen = Genie(tokz, inp_shape=(128, 128)).cuda().train()
    gen.tokenizer.eval()
    v = torch.randn(B, 3, 32, 128, 128, device='cuda')
    out = run(f'Genie (configs[4]: frozen MAGVIT2 tokenizer + R-lam + dynamics, 32x128x128, B={B}) [frames/s]', gen,
              lambda: gen.compute_loss(v)[0], B * 32, steps=steps, warm=1, quiet=quiet)
    del gen, tokz
    torch.cuda.empty_cache()
    return out


def run_quiet(fn, batch):
    """bench.py's entry: one of bench_lam / bench_dyn without the JSON print, fewer steps."""
    torch.manual_seed(0)
    return fn(batch, False, quiet=True, steps=3)


def main():
    torch.manual_seed(0)
    res = []
    argv = [a for a in sys.argv[1:] if not a.startswith('--')]
    want_cpu = '--cpu-baseline' in sys.argv
    opt = {a.split('=')[0]: int(a.split('=')[1]) for a in sys.argv[1:] if a.startswith('--') and '=' in a}
    lam_batch, dyn_batch = opt.get('--lam-batch', 16), opt.get('--dyn-batch', 32)
    which = argv or ['lam', 'dyn', 'repr']
    if 'lam' in which:
        res.append(bench_lam(lam_batch, want_cpu))
    if 'dyn' in which:
        res.append(bench_dyn(dyn_batch, want_cpu))
    if 'repr' in which:
        B = 2
        tokz = VideoTokenizer(REPR_TOK_ENC, REPR_TOK_DEC, d_codebook=10, gan_loss_weight=0., perc_loss_weight=0.).cuda().train()
        v = torch.randn(B, 3, 16, 64, 64, device='cuda')
        res.append(run(f'VideoTokenizer REPR_TOK (8+8 ST blocks, C=512, 16x16x16 latent, B={B}) [frames/s]', tokz, lambda: tokz(v)[0], B * 16))
    if 'genie4' in which:
        res.append(bench_genie4(opt.get('--genie-batch', 4)))
    if 'smallbatch' in which:
        # what an 8-GPU STRONG-scaling run sees per GPU: the MAGVIT2 tokenizer step at 4 / 8 clips and the LatentAction
        # step, issued from Python launch by launch vs one replay of the captured hipGraph (genie/graph.py); `host_issue_ms` = how long
        # Python needs to issue one eager step when nothing synchronises
        from genie.graph import GraphedTrainStep

        def eager_vs_graph(name, build, batch, units, loss_fn, steps=6):
            out = {'model': name}
            model = build()
            arena = ParamArena(model); arena.attach_weight_packs(model)
            def estep():
                o = loss_fn(model, batch); (o[0] if isinstance(o, (tuple, list)) else o).backward(); arena.adamw_step()
            for _ in range(3): estep()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(steps): estep()
            t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
            out['eager'] = {'ms_per_step': round((t2 - t0) / steps * 1e3, 2), 'host_issue_ms': round((t1 - t0) / steps * 1e3, 2)}
            del arena, model
            torch.cuda.empty_cache()
            model = build()
            arena = ParamArena(model); arena.attach_weight_packs(model)
            gs = GraphedTrainStep(model, arena, batch, loss_fn=loss_fn, warmup=2)
            for _ in range(2): gs(batch)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(steps): gs(batch)
            t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
            out['graph'] = {'ms_per_step': round((t2 - t0) / steps * 1e3, 2), 'host_issue_ms': round((t1 - t0) / steps * 1e3, 2),
                            'loss': round(gs.loss.item(), 5)}
            out['units_per_s'] = {'eager': round(units / out['eager']['ms_per_step'] * 1e3, 1), 'graph': round(units / out['graph']['ms_per_step'] * 1e3, 1)}
            print(json.dumps(out), flush=True)
            del gs, arena, model
            torch.cuda.empty_cache()
            return out

        def build_tok():
            torch.manual_seed(0)
            return VideoTokenizer(MAGVIT2_ENC_DESC, MAGVIT2_DEC_DESC, d_codebook=18, gan_loss_weight=0., perc_loss_weight=0.).cuda().train()

        def build_lam():
            torch.manual_seed(0)
            return LatentAction(LATENT_ACT_ENC, LATENT_ACT_DEC, d_codebook=8, inp_channels=3, inp_shape=(64, 64), n_embd=256).cuda().train()
        for B in (4, 8):
            v = torch.randn(B, 3, 16, 64, 64, device='cuda')
            res.append(eager_vs_graph(f'VideoTokenizer MAGVIT2 (configs[1]) at {B} clips per GPU [frames/s]', build_tok, v, B * 16, lambda m, b: m(b)[0]))
        v = torch.randn(2, 3, 16, 64, 64, device='cuda')
        res.append(eager_vs_graph('LatentAction (configs[2]) B=2 [frames/s]', build_lam, v, 2 * 16, lambda m, b: m(b)[1]))

        # DynamicsModel (configs[3]): the default loss gathers the masked rows (a data-dependent row count: eager only); fixed_rows=True is
        # its shape-stable form -- eager vs replayed.  The eager gathered-rows step is the `dyn` entry of this script.
        def build_dyn():
            torch.manual_seed(0)
            return DynamicsModel((('space-time_attn', {'n_rep': 8, 'n_head': 8, 'd_head': 64}),), tok_vocab=2 ** 18, act_vocab=8, embed_dim=512).cuda().train()
        B = 4
        gm = torch.Generator(device='cuda').manual_seed(1)
        batch = (torch.randint(0, 2 ** 18, (B, 16, 8, 8), device='cuda', generator=gm), torch.randint(0, 8, (B, 16), device='cuda', generator=gm),
                 torch.rand(B, 16, 8, 8, device='cuda', generator=gm) < 0.75)
        res.append(eager_vs_graph(f'DynamicsModel (configs[3]) B={B}, shape-stable loss (fixed_rows) [latent frames/s]', build_dyn, batch, B * 16,
                                  lambda m, b: m.compute_loss(b[0], b[1], mask=b[2], fixed_rows=True)))
    os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
    tag = '_'.join(which)
    json.dump(res, open(os.path.join(ROOT, 'gpurun_out', f'bench_models_{tag}.json'), 'w'), indent=1)


if __name__ == '__main__':
    main()
