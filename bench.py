#!/usr/bin/env python
"""Headline benchmark: VideoTokenizer (MAGVIT2 blueprint, d_codebook=18) TRAINING on 16x64x64 clips.

  python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run, one rank per GPU)

One step = one full pass of the hot path over one batch of synthetic clips per GPU, inputs resident in HBM:
encode -> LFQ (training mode: quantise + entropy/commit loss over 2^18 codes) -> decode -> MSE + quant loss
-> backward through everything -> gradient all-reduce (N > 1) -> fused AdamW on all 375.6 M parameters.
(R-fwd loss, SURVEY.md 8c: the GAN and VGG16-perceptual critics cannot run offline and are outside the hot path.)

Prints ONE JSON line (rank 0): metric video-frames/sec over ALL GPUs, plus
  roofline     -- the dominant kernel (the conv kernel variant with the largest share of the step; normally the kw-triple
                  gather-GEMM behind Conv3d fwd/dgrad) priced against the dense bf16 MFMA peak: algorithmic FLOPs of every
                  launch / HIP-event time of every launch, timed region only
  cpu_baseline -- the oracle (a port of the reference's algorithm, oracle/genie_oracle.py) doing the same training step
                  on this box's host cores, bounded to one B=1 step.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, 'open-genie_amd')):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch
import torch.distributed as dist

BF16_MFMA_PEAK_TFLOPS = 2500.0      # dense bf16, MI355X_MICROARCH.md
CLIP = (3, 16, 64, 64)
TRAIN_GFLOP_PER_CLIP = 7518.0       # 3 x 2506.1 GFLOP forward


def effective_cpus() -> int:
    """CPUs this process may actually use: affinity mask, capped by the cgroup CPU quota (os.cpu_count() reports the host)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, n)


def _cpu_baseline_worker(threads: int, frames: int, q) -> None:
    import copy
    import statistics

    import torch as T
    from genie import MAGVIT2_DEC_DESC, MAGVIT2_ENC_DESC, VideoTokenizer
    from oracle import genie_oracle as O
    from oracle import ref_import
    T.manual_seed(0)
    T.set_num_threads(threads)
    x = T.randn(1, CLIP[0], frames, CLIP[2], CLIP[3])
    kind = 'port'
    if ref_import.reference_available():
        # the REAL reference modules ("reference modules via the 8c stubs"), whenever /root/reference is present (the build
        # container; the GPU box has no copy of it): reference VideoTokenizer.encode / .quant / .decode composed into the R-fwd step
        # (its own forward() needs the VGG16 critic), reference-default AdamW
        try:
            ref = ref_import.import_reference()
            rm = ref.VideoTokenizer(copy.deepcopy(MAGVIT2_ENC_DESC), copy.deepcopy(MAGVIT2_DEC_DESC), d_codebook=18, gan_loss_weight=0., perc_loss_weight=0.).train()
            ropt = T.optim.AdamW(rm.parameters(), lr=1e-3, weight_decay=0.01)

            def step():
                t0 = time.perf_counter()
                ropt.zero_grad(set_to_none=True)
                enc = rm.encode(x)
                (qt, _), ql = rm.quant(enc, transpose=True)
                loss = T.nn.functional.mse_loss(rm.decode(qt), x) + ql
                loss.backward()
                ropt.step()
                return time.perf_counter() - t0
            kind = 'reference'
        except Exception:
            kind = 'port'
    if kind == 'port':
        m = VideoTokenizer(MAGVIT2_ENC_DESC, MAGVIT2_DEC_DESC, d_codebook=18, gan_loss_weight=0., perc_loss_weight=0.)
        sd = {k: (v.detach().contiguous().clone().requires_grad_(v.is_floating_point())) for k, v in m.state_dict().items()}
        del m
        opt = T.optim.AdamW([v for v in sd.values() if v.requires_grad], lr=1e-3, weight_decay=0.01)

        def step():
            t0 = time.perf_counter()
            opt.zero_grad(set_to_none=True)
            loss, _, _, _ = O.tokenizer_forward_hotpath(x, sd, MAGVIT2_ENC_DESC, MAGVIT2_DEC_DESC, 18)
            loss.backward()
            opt.step()
            return time.perf_counter() - t0

    first = step()                                          # warm-up 1 (also tells how many more steps the budget allows)
    warm, timed = (2, 3) if first < 14.0 else ((1, 1) if first < 40.0 else (1, 0))
    timed = max(timed, int(os.environ.get('GENIE_CPU_BASELINE_MIN_TIMED', '0')))      # scripts/cpu_baseline_reference.py: a warm step even on a slow host
    for _ in range(warm - 1):
        step()
    times = [step() for _ in range(timed)] or [first]
    q.put((statistics.median(times), warm if timed else 0, len(times), T.get_num_threads(), kind))


def cpu_baseline(budget_s: float = 170.0):
    """The CPU path doing the SAME training step -- fwd + bwd + AdamW of the MAGVIT2 tokenizer on one 16x64x64 clip -- on the host cores:
    the REAL reference modules when /root/reference is present (`kind: "reference"`; the build container), otherwise the oracle's
    restatement of them (`kind: "port"`; the GPU box, which has no copy of the reference).  BASELINE.md section 3 protocol, 2 warm-up +
    3 timed steps, median (fewer when the first (cold) step takes longer than 14 s; 4-frame clips if nothing finishes inside the
    budget).  Runs in a child process with a hard time budget; threads = every CPU this process may use (affinity mask capped by the
    cgroup quota; the GPU boxes of the pool grant 16 of their 256), at most 128."""
    import multiprocessing as mp
    threads = min(effective_cpus(), 128)
    ctx = mp.get_context('spawn')
    for frames in (CLIP[1], 4):
        q = ctx.Queue()
        p = ctx.Process(target=_cpu_baseline_worker, args=(threads, frames, q))
        p.start()
        p.join(budget_s)
        if p.is_alive():
            p.kill()
            p.join()
            continue
        if p.exitcode == 0:
            dt, warm, timed, nthr, kind = q.get()
            return {'value': round(frames / dt, 4), 'unit': 'video-frames/sec', 'cores': nthr, 'kind': kind,
                    'sample': f'training step (fwd+bwd+AdamW) of the MAGVIT2 tokenizer ({"reference modules" if kind == "reference" else "oracle restatement of the reference"}) on one {frames}x64x64 clip, fp32, torch CPU, {nthr} threads of '
                              f'{os.cpu_count()} host CPUs: {warm} warm-up + {timed} timed steps, median {dt:.2f} s/step'}
    return {'value': None, 'unit': 'video-frames/sec', 'cores': threads, 'kind': 'port', 'sample': f'did not finish within {budget_s:.0f} s'}


def csrc_sha16() -> str:
    """Fingerprint of the kernel sources (the GPU box has no .git): profiles/rNN_summary.json records it, so that PMC figures measured
    on other kernels are never attached to a run."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, 'open-genie_amd', 'csrc')
    for f in sorted(os.listdir(d)):
        if f.endswith(('.hip', '.h', '.cpp')):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), 'rb').read())
    return h.hexdigest()[:16]


def socket_power_w():
    """Average socket power in watts as `rocm-smi --showpower --json` reports it for device 0, or None (tool missing / no permission)."""
    import shutil
    import subprocess
    exe = shutil.which('rocm-smi') or '/opt/rocm/bin/rocm-smi'
    try:
        out = subprocess.run([exe, '-d', '0', '--showpower', '--json'], capture_output=True, text=True, timeout=20).stdout
        card = next(iter(json.loads(out).values()))
        for k, v in card.items():
            if 'power' in k.lower() and 'w' in k.lower():
                return float(str(v).split()[0])
    except Exception:
        return None
    return None


def power_under_load(fn, seconds: float = 2.0, launch_ms: float = 3.0):
    """Socket power while `fn` (one asynchronous kernel launch) runs back to back for ~`seconds`: the launches are enqueued first, the
    host then samples rocm-smi while the GPU works through them.  Returns (mean watts or None, samples)."""
    n = max(50, int(seconds * 1e3 / max(launch_ms, 0.05)))
    for _ in range(n):
        fn()
    samples = []
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds * 0.8 and len(samples) < 4:
        w = socket_power_w()
        if w is None:
            break
        samples.append(w)
    torch.cuda.synchronize()
    return (round(sum(samples) / len(samples), 1) if samples else None), samples


def power_cap_probe(batch: int = 64):
    """How much of the distance between `roofline.achieved` and the 2.5 PFLOP/s peak is the chip's POWER cap rather than the kernel's
    schedule: one launch shape of the dominant kernel (256 -> 256 channels, 3x3x3, 16x32x32, the step's batch) timed on random bf16 operands
    and on all-zero operands -- the identical instruction stream and cycle count, a fraction of the switching power, so the clock stays
    near its 2.4 GHz ceiling (rocprofv3 GRBM_GUI_ACTIVE of the two: 1.66 vs 2.30 GHz, profiles/r03_power_cap_ab.log)."""
    from genie import _hip
    from genie import conv as gconv
    import scripts.microbench as mb
    spec = gconv.same_spec(256, 256, (3, 3, 3))
    out = {'layer': f'256->256 k3 @16x32x32, {batch} clips, forward', 'peak_tflops': BF16_MFMA_PEAK_TFLOPS}
    fl = 2.0 * batch * 16 * 32 * 32 * 256 * 256 * 27
    for tag, zero in (('random_operands', False), ('zero_operands', True)):
        x = mb.empty_cl(batch, 256, 16, 32, 32, 'cuda')
        x.zero_() if zero else x.copy_(torch.randn(batch, 256, 16, 32, 32, device='cuda'))
        wt = (torch.randn(256, 256, 3, 3, 3, device='cuda') * (0.0 if zero else 0.05)).contiguous(memory_format=torch.channels_last_3d)
        wf = gconv.pack_weight_fwd(wt, spec)
        ms = mb.timeit(lambda: gconv.conv_forward(x, wf, None, spec), 10)
        out[tag] = {'ms': round(ms, 4), 'tflops': round(fl / ms / 1e9, 1), 'mfma_frac': round(fl / ms / 1e9 / BF16_MFMA_PEAK_TFLOPS, 4)}
        # energy per useful FLOP: socket power while this launch repeats for ~2 s
        watts, samples = power_under_load(lambda: gconv.conv_forward(x, wf, None, spec), 2.0, ms)
        if watts is not None:
            out[tag].update({'socket_watts': watts, 'watt_samples': samples, 'pJ_per_flop': round(watts / (fl / ms / 1e9) , 3)})      # W / (TFLOP/s) = pJ / FLOP
        out['kernel'] = gconv.VARIANT_NAMES.get(_hip.load_library().genie_last_conv_variant())
        del x, wt, wf
    out['zero_over_random'] = round(out['zero_operands']['tflops'] / out['random_operands']['tflops'], 3)
    torch.cuda.empty_cache()
    return out


def side_kernels(batch: int = 64):
    """The other two quantities BASELINE.json's metric names, measured at kernel level with HIP events on resident synthetic inputs
    (a few ms in total): MFMA utilisation of the space-time attention kernels on the two long-sequence shapes of SURVEY.md 8a
    (a10), and HBM GB/s (algorithmic bytes) of the bandwidth-bound CausalConv3d / GroupNorm family."""
    import scripts.microbench as mb
    mb.RESULTS.clear()
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        mb.bench_attn(30, only=('lam spatial S=4096', 'yaml_tok spatial S=1024'))      # 30 launches per measurement, best of three
        mb.bench_hbm(10, quick=True)                     # B = 8: the round-1 / round-2 protocol (134 MB per tensor, partly cache-resident)
        mb.bench_hbm(6, quick=True, B=batch)             # the step's own batch: 1.07 GB per 128-channel tensor at 64 clips, pure HBM streams
    att, hbm = {}, {}
    for r in mb.RESULTS:
        if r['section'] == 'attn' and 'attention' in r['name']:
            att[r['name']] = {'ms': r['ms'], 'tflops': r['tflops'], 'mfma_frac': r['mfma_frac']}
        elif r['section'] == 'hbm' and 'gbps' in r:
            hbm[r['name']] = {'ms': r['ms'], 'gbps': r['gbps'], 'hbm_frac': r['hbm_frac']}
            if 'hbm_frac_padded_pitch' in r:             # narrow convs: `hbm_frac` prices SURVEY 8(d)'s (3 + 128) channels, this the 8-channel pitch the kernels move
                hbm[r['name']]['hbm_frac_padded_pitch'] = r['hbm_frac_padded_pitch']
            if 'hbm_frac_min' in r:                      # against the operation's minimal traffic (1R + 1W forward, 2R + 1W backward)
                hbm[r['name']].update({'gbps_min': r['gbps_min'], 'hbm_frac_min': r['hbm_frac_min']})
    best = max((v['mfma_frac'] for k, v in att.items() if 'fwd' in k), default=None)
    best_bwd = max((v['mfma_frac'] for k, v in att.items() if 'bwd' in k), default=None)
    from genie import _hip
    lib = _hip.load_library()
    fam = {'lean_mode': lib.genie_attention_lean_mode(-1), 'resident_blocks_per_cu': {k: lib.genie_attention_lean_occupancy(i) for i, k in enumerate(('fwd', 'bwd_dq', 'bwd_dkv'))},
           'source': 'open-genie_amd/csrc/attention_lean.hip (d_head 64; bits of lean_mode: include/genie_hip.h genie_attention_lean_mode)'}
    # (not under the profiler: the probe launches the dominant kernel itself and would mix into its rocprofv3 / PMC averages)
    return {**({} if os.environ.get('GENIE_BENCH_NO_PROBE') else {'power_cap': power_cap_probe(batch)}),
            'st_attention': {'peak_tflops': BF16_MFMA_PEAK_TFLOPS, 'flop_count': 'dense 4 S^2 C per sequence forward, 2.5x that backward',
                             'causal_skipping': 'none credited and none present: both shapes are SPATIAL attention (non-causal, every key tile is executed), so the dense count '
                                                'IS the executed count; causal attention here is temporal, T <= 32, on the packed traffic-bound kernels (priced in GB/s, not TFLOP/s)',
                             'best_fwd_mfma_frac': best, 'best_bwd_mfma_frac': best_bwd, 'kernel_family': fam,
                             'batch': "no suffix: 1 clip (S = 4096) / 2 clips (S = 1024), the grids of rounds 1-5 (2-4 rounds of workgroups: mostly ramp and tail); "
                                      "'(4 clips)': the batch BASELINE configs[0] quotes -- the steady state a training step runs in",
                             'kernels': att},
            'hbm_kernels': {'peak_gbps': 8000.0, 'bytes': 'what the passes of the call move (stated per entry); *_min: the minimal traffic of the operation', 'kernels': hbm}}


def launcher_command(gpus: int, argv, port: int) -> list:
    """The `torch.distributed.run` command line of `python bench.py --gpus N` started plainly: one rank per GPU of this node, rendezvous on
    127.0.0.1 (the container hostname may not resolve) -- the same line the driver uses."""
    return [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={gpus}', '--master-addr', '127.0.0.1',
            '--master-port', str(port), os.path.abspath(__file__), *argv]


def become_launcher(gpus: int, argv) -> None:
    """`python bench.py --gpus N` without a launcher around it: replace this process by the launcher (never run a smaller job under the label)."""
    import socket
    if torch.cuda.device_count() < gpus:
        raise SystemExit(f'bench.py --gpus {gpus}: only {torch.cuda.device_count()} GPU(s) visible; refusing to run a smaller job under that label')
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    # dmabuf IPC is the only kind the host driver supports: without this RCCL's peer mappings fail with hipIpcGetMemHandle: invalid argument
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    cmd = launcher_command(gpus, argv, port)
    os.execv(cmd[0], cmd)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=6)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--batch', type=int, default=int(os.environ.get('GENIE_BENCH_BATCH', 64)),
                    help='clips per GPU per step (64: 69 GB of the 288 GB HBM; same-box sweep 8 / 16 / 32 / 64 / 96 clips: 1520 / 1766 / 1816 / 1863 / 1863 frames/s).  '
                         'Weak scaling (the default, `scaling: weak`) keeps this fixed as N grows; for a STRONG-scaling reading of a fixed global batch of 64 '
                         'run --gpus N --batch 64/N (8 clips per GPU at N = 8: the low-resolution layers and the fixed per-step costs then weigh ~20 %% more)')
    ap.add_argument('--buckets', type=int, default=int(os.environ.get('GENIE_DP_BUCKETS', 8)), help='N > 1: gradient all-reduce buckets of equal bytes (cut at layer boundaries)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-kernel-events', action='store_true')
    ap.add_argument('--all-kernel-events', action='store_true', help='HIP events around EVERY conv launch (full conv_kernels table; costs ~4 %% of the step)')
    ap.add_argument('--dump', type=str, default='')
    ap.add_argument('--no-in-order-pass', action='store_true', help='skip the three extra in-order steps behind roofline_in_order (rocprofv3 runs: keeps every launch of the trace in the timed regime)')
    ap.add_argument('--grad-compress', choices=['none', 'bf16'], default=os.environ.get('GENIE_GRAD_COMPRESS', 'none'),
                    help='gradient all-reduce payload: fp32 (exact, default) or bf16 (half the xGMI bytes)')
    ap.add_argument('--allreduce', choices=['allreduce', 'rs_ag'], default=os.environ.get('GENIE_DP_ALGORITHM', 'allreduce'),
                    help='N > 1: one all_reduce per bucket (default) or an explicit reduce-scatter + all-gather over all ranks (SURVEY.md 5 / 8e)')
    ap.add_argument('--async-wgrad', type=int, default=int(os.environ.get('GENIE_ASYNC_WGRAD', 0)),
                    help='1: weight-gradient kernels on a side stream, overlapping the HBM-bound GroupNorm / element-wise passes of backward (conv launches wait for it); 2: unordered; 0: off')
    ap.add_argument('--no-other-configs', action='store_true', help='skip the configs[2] / [3] / [4] side measurements (`other_configs`)')
    ap.add_argument('--lam-batch', type=int, default=16)
    ap.add_argument('--dyn-batch', type=int, default=32)
    ap.add_argument('--genie-batch', type=int, default=4, help='clips of 32x128x128 in the configs[4] side measurement (16 GB each)')
    ap.add_argument('--dp-loopback', action='store_true', help='N = 1 only: run the RCCL bucket all-reduces on a single-rank group (side-stream path on one GPU)')
    args = ap.parse_args()

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        become_launcher(args.gpus, sys.argv[1:])

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs the MI355X (no CPU fallback for the product path)')
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('nccl', device_id=dev)
    if world != args.gpus:
        raise SystemExit(f'bench.py: --gpus {args.gpus} but WORLD_SIZE={world}')
    if args.dp_loopback and world == 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)

    from genie import MAGVIT2_DEC_DESC, MAGVIT2_ENC_DESC, VideoTokenizer, conv as gconv
    from genie import functional as GF
    from genie.trainer import DataParallel, ParamArena, Trainer, sync_replicas
    GF.ASYNC_WGRAD = int(args.async_wgrad)

    torch.manual_seed(0)
    model = VideoTokenizer(MAGVIT2_ENC_DESC, MAGVIT2_DEC_DESC, d_codebook=18, gan_loss_weight=0., perc_loss_weight=0.).to(dev).train()
    arena = ParamArena(model)
    sync_replicas(arena, model)                            # every rank starts from rank 0's weights (not from "everybody seeded alike")
    arena.attach_weight_packs(model)                       # bf16 packs ride on the optimiser kernel + one batched transpose
    dp = DataParallel(arena.grads, compress=args.grad_compress, loopback=args.dp_loopback, algorithm=args.allreduce)
    if dp.active:                                          # decoder gradients reduce while the encoder is still in backward
        # the same cut rule Trainer.fit applies (genie/trainer.py::Trainer.bucket_modules): the bench measures the path users run
        dp.install_overlap_hooks(arena, model, Trainer.bucket_modules(arena, model, max(1, args.buckets)))
    B = args.batch
    gen = torch.Generator(device=dev).manual_seed(1234 + rank)      # rank r holds clips r::world of the synthetic stream
    clips = [torch.randn(B, *CLIP, device=dev, generator=gen) for _ in range(2)]

    def step(i):
        loss, aux = model(clips[i % len(clips)])
        loss.backward()
        dp.finish()
        arena.adamw_step(lr=1e-3, weight_decay=0.01)
        return loss, aux

    for i in range(args.warmup):
        loss, aux = step(i)
    torch.cuda.synchronize()
    prof = None
    if not args.no_kernel_events:
        prof = gconv.PROFILER = gconv.LaunchProfiler(only_triple=not args.all_kernel_events and not args.dump)
    dp.trace = dp.active                                   # HIP events around the bucket all-reduces and finish()'s wait (the `comm` object)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss, aux = step(args.warmup + i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    gconv.PROFILER = None
    dp.trace = False
    comm = dp.comm_report() if dp.active else None
    # The timed region runs every kernel in order on one stream (kernel durations = the kernels' own rates; `roofline` comes from
    # here).  Putting the weight-gradient kernels and the LFQ loss on a side stream (functional.ASYNC_WGRAD = 2) shortens the step by
    # ~4 % but a forward / backward-data kernel's wall time then includes the share of the chip a concurrent weight-gradient kernel
    # took; three more steps OUTSIDE the timed region report that variant next to the headline (`wgrad_side_stream`).
    prof_inorder, side_alt = None, None
    # what the HIP events around the kw-triple launches cost the headline: three more steps with the profiler off, same
    # data, same stream -- outside the timed region, reported as `event_overhead`
    no_events = None
    if prof is not None and world == 1:
        t_ne = time.perf_counter()
        for i in range(3):
            step(args.warmup + args.steps + i)
        torch.cuda.synchronize()
        no_events = (time.perf_counter() - t_ne) / 3 * 1e3
    if prof is not None and world == 1 and not args.no_in_order_pass:
        if GF.ASYNC_WGRAD:                                 # --async-wgrad 1 / 2: the in-order kernel rates next to the timed region
            GF.join_wgrad()
            GF.ASYNC_WGRAD = 0
            step(args.warmup + args.steps)
            torch.cuda.synchronize()
            prof_inorder = gconv.PROFILER = gconv.LaunchProfiler(only_triple=True)
            t_io = time.perf_counter()
            for i in range(3):
                step(args.warmup + args.steps + 1 + i)
            torch.cuda.synchronize()
            ms_inorder = (time.perf_counter() - t_io) / 3 * 1e3
            gconv.PROFILER = None
            GF.ASYNC_WGRAD = int(args.async_wgrad)
        else:
            GF.ASYNC_WGRAD = 2
            step(args.warmup + args.steps)
            torch.cuda.synchronize()
            t_io = time.perf_counter()
            for i in range(3):
                step(args.warmup + args.steps + 1 + i)
            torch.cuda.synchronize()
            ms_alt = (time.perf_counter() - t_io) / 3 * 1e3
            GF.join_wgrad()
            GF.ASYNC_WGRAD = 0
            side_alt = {'ms_per_step': round(ms_alt, 3), 'video_frames_per_sec': round(B * CLIP[1] / ms_alt * 1e3, 2), 'steps': 3,
                        'inside_timed_region': False, 'how': 'python bench.py --async-wgrad 2'}
    t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = t.item()
    scal = dp.reduce_scalars([loss, aux[0], aux[4]])
    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    ms_per_step = elapsed / args.steps * 1e3
    frames = world * B * CLIP[1] * args.steps
    value = frames / elapsed
    out = {
        'metric': 'video-frames/sec (VideoTokenizer train step, MAGVIT2 blueprint, 16x64x64 clips)',
        'value': round(value, 2), 'unit': 'video-frames/sec', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': round(ms_per_step, 3), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': 'bf16', 'data': 'synthetic',
        'config': {'workload': 'configs[1]: VideoTokenizer (MAGVIT2_ENC/DEC_DESC, d_codebook=18) training, 16x64x64 random clips, bf16 activations / fp32 master weights; '
                               'step = encode + LFQ(train) + decode + MSE + quant loss + backward + AdamW (R-fwd loss)',
                   'clips_per_gpu': B, 'global_batch': B * world, 'clip': list(CLIP), 'params': 375554837, 'parallelism': f'dp{world}',
                   'wgrad_stream': {0: 'in order', 1: 'on (conv launches wait for it: overlaps GroupNorm / element-wise passes only)', 2: 'on (unordered)'}.get(int(args.async_wgrad)),
                   'grad_allreduce': {'algorithm': args.allreduce, 'payload': 'fp32' if args.grad_compress == 'none' else 'bf16', 'buckets': len(dp.buckets),
                                      'bytes_per_step': dp.bytes_reduced // max(1, args.steps + args.warmup), 'overlapped_with_backward': dp.active,
                                      'loopback': bool(args.dp_loopback and world == 1)},
                   'final_loss': round(scal[0].item(), 5), 'peak_mem_GB': round(torch.cuda.max_memory_allocated() / 2 ** 30, 1)},
        'model_tflops_per_gpu': round(TRAIN_GFLOP_PER_CLIP * B * args.steps / elapsed / 1e3, 2),
    }
    if no_events is not None:
        out['event_overhead'] = {'ms_per_step_without_events': round(no_events, 3), 'video_frames_per_sec_without_events': round(B * CLIP[1] / no_events * 1e3, 2),
                                 'event_overhead_pct': round((ms_per_step / no_events - 1.0) * 100, 2), 'steps': 3, 'inside_timed_region': False,
                                 'what': 'the timed region records two HIP events around every kw-triple conv launch (the `roofline` measurement); '
                                         'these three extra steps run without them'}
    if prof is not None:
        summ = prof.summary()
        dom = max(summ, key=lambda k: summ[k]['ms']) if summ else None      # the kernel variant with the largest share of the step
        if dom is not None:
            d = summ[dom]
            ach = d['flops'] / (d['ms'] * 1e-3) / 1e12
            ach_x = d.get('flops_exec', d['flops']) / (d['ms'] * 1e-3) / 1e12
            out['roofline'] = {'kernel': dom, 'bound': 'mfma', 'achieved': round(ach, 2), 'peak': BF16_MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                               'frac': round(ach / BF16_MFMA_PEAK_TFLOPS, 4), 'traffic': None,
                               # `achieved` / `frac` price the ALGORITHMIC count (2 M Cout Cin 27 per launch, zero padding included);
                               # the kernel skips the (frame, dt) pairs whose source frame is time padding (tri_trim_range: 2 of 48 at 16
                               # frames, 2 of 24 at 8, 3 of 48 causal), so what the matrix pipe EXECUTES is this much less:
                               'flop_count': 'algorithmic: dense 2*M*Cout*Cin*27 per launch incl. zero time-padding (SURVEY 8d); flops_executed excludes the (frame, dt) pairs the kernel skips',
                               'flops_algorithmic': d['flops'], 'flops_executed': d.get('flops_exec', d['flops']),
                               'achieved_executed': round(ach_x, 2), 'frac_executed': round(ach_x / BF16_MFMA_PEAK_TFLOPS, 4),
                               'launches': d['launches'], 'avg_launch_ms': round(d['ms'] / d['launches'], 4),
                               'share_of_step_time': round(d['ms'] / (elapsed * 1e3), 4)}
        if dom is not None and prof_inorder is not None:
            si = prof_inorder.summary().get(dom)
            if si and si['ms'] > 0:
                ach_i = si['flops'] / (si['ms'] * 1e-3) / 1e12
                out['roofline']['note'] = ('timed region runs the weight-gradient kernels on a side stream: this kernel\'s wall time includes the share of '
                                           'the chip they took; roofline_in_order = the same kernel with everything in order (3 extra steps)')
                out['roofline_in_order'] = {'kernel': dom, 'bound': 'mfma', 'achieved': round(ach_i, 2), 'peak': BF16_MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                                            'frac': round(ach_i / BF16_MFMA_PEAK_TFLOPS, 4), 'launches': si['launches'],
                                            'avg_launch_ms': round(si['ms'] / si['launches'], 4), 'steps': 3, 'ms_per_step': round(ms_inorder, 3),
                                            'inside_timed_region': False}
        if side_alt is not None:
            out['wgrad_side_stream'] = side_alt
        kern = {k: {'launches': v['launches'], 'ms_per_step': round(v['ms'] / args.steps, 3),
                    'tflops': round(v['flops'] / (v['ms'] * 1e-3) / 1e12, 2) if v['ms'] > 0 else None} for k, v in summ.items()}
        out['conv_kernels'] = kern
        # the stem / head CausalConv3d launches INSIDE the timed region (HIP events around five launches per step), on SURVEY 8(d)'s bytes:
        # (3 + 128) channels x 2 B per pixel + the weight tensor.  The stand-alone figures under `hbm_kernels` depend on what ran on the chip
        # before them (the head forward read 0.54 or 0.59 in the same process on the same box); these are the rates the training step sees.
        npx = B * 16 * 64 * 64
        nb = npx * (3 + 128) * 2 + 128 * 3 * 27 * 2
        in_step = {}
        for var in ('conv_narrow_in_kernel', 'conv_narrow_out_kernel', 'conv_narrow_wgrad_kernel'):
            for label, b in summ.get(var, {}).get('by_label', {}).items():
                if b['launches'] and b['ms'] > 0:
                    ms1 = b['ms'] / b['launches']
                    in_step[f'{var} | {label}'] = {'launches': b['launches'], 'ms': round(ms1, 5), 'gbps': round(nb / ms1 / 1e6, 1), 'hbm_frac': round(nb / ms1 / 1e6 / 8000.0, 4)}
        if in_step:
            out['causal_conv3d_in_step'] = {'peak_gbps': 8000.0, 'bytes_per_launch': nb, 'bytes': 'SURVEY 8(d): (3 + 128) channels x 2 B x pixels + weights, per launch of %d clips' % B,
                                            'kernels': in_step}
        if args.dump:
            os.makedirs(os.path.dirname(args.dump) or '.', exist_ok=True)
            with open(args.dump, 'w') as f:
                json.dump(summ, f, indent=1)
    if comm:
        # why it scales (or does not): what backward hid of the gradient all-reduce and what it did not, bucket by bucket (rank 0's view)
        out['comm'] = comm
    if world == 1 and not args.no_kernel_events:
        out.update(side_kernels(B))
    # `traffic`: HBM bytes per launch of the dominant kernel from PMC counters.  Counters cannot be collected inside this run (rocprofv3
    # wraps the process), so the figure comes from the committed PMC passes of THIS command (scripts/profile_bench.sh ->
    # profiles/rNN_summary.json) -- and only if that profile was taken on the same kernel sources and batch; otherwise it is stale and
    # stays null (a silently carried-over number is worse than none).
    # the newest round's summary first: profiles/r04_summary.json, r03_..., ...
    import re
    rounds = {int(mt.group(1)): f for f in os.listdir(os.path.join(ROOT, 'profiles')) for mt in [re.match(r'r(\d+)_summary\.json$', f)] if mt}
    prof_summary = os.path.join(ROOT, 'profiles', rounds[max(rounds)]) if rounds else ''      # newest round by NUMBER (r10 after r9)
    if 'roofline' in out and prof_summary:
        try:
            ps = json.load(open(prof_summary))
            tr = ps.get('traffic', {}).get(out['roofline']['kernel'])
            meta = ps.get('meta', {})
            fresh = meta.get('csrc_sha16') == csrc_sha16() and int(meta.get('batch', -1)) == B
            if tr and fresh:
                out['roofline']['traffic'] = tr['bytes_per_launch']
                out['roofline']['traffic_source'] = tr['source']
            elif tr:
                out['roofline']['traffic_stale'] = {'bytes_per_launch': tr['bytes_per_launch'], 'profile': os.path.basename(prof_summary),
                                                    'why': f"profile taken at batch {meta.get('batch')} / kernel sources {meta.get('csrc_sha16')}, this run: batch {B} / {csrc_sha16()}"}
        except Exception:
            pass
    if world == 1 and not args.no_kernel_events and not os.environ.get('GENIE_BENCH_NO_PROBE') and not args.no_other_configs:
        # BASELINE configs[2], [3] and [4] at chip-filling batches, one training step each on THIS box (a driver-run record
        # for them): ms per step, units per second and the kernel family with the largest share, priced like `roofline`.  Side
        # measurements AFTER the timed region; the headline above is unaffected.
        try:
            clips.clear()
            model.zero_grad(set_to_none=True)
            torch.cuda.empty_cache()
            import scripts.bench_models as bm
            oc = {}
            for key, fn, b in (('configs[2] LatentAction', bm.bench_lam, args.lam_batch), ('configs[3] DynamicsModel', bm.bench_dyn, args.dyn_batch),
                                ('configs[4] Genie 32x128x128', bm.bench_genie4, args.genie_batch)):
                r = bm.run_quiet(fn, b)
                oc[key] = {'model': r['model'], 'batch': b, 'ms_per_step': r['ms_per_step'], 'ms_per_step_all': r.get('ms_per_step_all'), 'units_per_s': r['units_per_s'], 'peak_mem_GB': r['peak_mem_GB'],
                           'roofline': r.get('roofline'), 'top_kernels': dict(list(r.get('kernels', {}).items())[:4])}
            out['other_configs'] = oc
        except Exception as ex:                            # a side measurement must never cost the headline line
            out['other_configs'] = {'error': f'{type(ex).__name__}: {ex}'}
    if world == 1 and not args.no_cpu_baseline:
        out['cpu_baseline'] = cpu_baseline()
        # the GPU boxes have no copy of the reference, so the on-box figure is the oracle port; the REAL reference modules timed in the build
        # container (scripts/cpu_baseline_reference.py, committed) ride along so that the line carries both kinds
        recs = {int(mt.group(1)): f for f in os.listdir(os.path.join(ROOT, 'profiles')) for mt in [re.match(r'r(\d+)_cpu_baseline_reference\.json$', f)] if mt}
        if recs and out['cpu_baseline'].get('kind') != 'reference':
            try:
                rr = json.load(open(os.path.join(ROOT, 'profiles', recs[max(recs)])))
                out['cpu_baseline']['reference_record'] = {'value': rr['value'], 'unit': rr['unit'], 'cores': rr['cores'], 'kind': rr['kind'],
                                                           'sample': rr['sample'], 'file': 'profiles/' + recs[max(recs)], 'measured': 'build container, not this box'}
            except Exception:
                pass
    if world > 1 or dist.is_initialized():
        dist.destroy_process_group()
    # RCCL prints its version banner through C stdio; when stdout is a file that buffer is flushed at exit, i.e. AFTER Python's prints:
    # flush it now so that the JSON line is the LAST line of the output
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    print(json.dumps(out), flush=True)


if __name__ == '__main__':
    main()
