#!/usr/bin/env python
"""Kernel-level microbenchmarks on one MI355X (run via gpurun): HIP-event timing of single C-ABI entry points on
synthetic inputs resident in HBM.  Writes gpurun_out/microbench.json and prints one line per case.

  python scripts/microbench.py [attn] [hbm] [conv] [--iters 20]

Sections
  attn : rotary+LayerNorm prologue, attention forward, attention backward for the ST-block shapes of SURVEY.md 8a (a9-a12)
  hbm  : the HBM-bound family (GroupNorm fwd/bwd, stem/head CausalConv3d, AdamW, layout conversion) as GB/s of ALGORITHMIC bytes
  conv : interior Conv3d shapes of the MAGVIT2 tokenizer, forward / dgrad / wgrad, TFLOP/s
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'open-genie_amd')):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch

from genie import _hip, conv as gconv, functional as GF
from genie.cl import empty_cl, pitch_of, to_cl

PEAK_TF, PEAK_HBM = 2500.0, 8000.0
RESULTS = []
FILTER = os.environ.get('MB_FILTER', '')
NO_WGRAD = os.environ.get('MB_NO_WGRAD', '') == '1'
NO_RESID, NO_OATTN, NO_LSE = (os.environ.get(k, '') == '1' for k in ('MB_ATTN_NO_RESID', 'MB_ATTN_NO_OATTN', 'MB_ATTN_NO_LSE'))    # forward epilogue probes


def timeit(fn, iters, warm=5, reps=3):
    """min over `reps` back-to-back measurements of `iters` calls (the first measurement of a process can carry a one-off
    ~40 ms runtime hiccup, and the clock ramps over the first few hundred ms)."""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    best = float('inf')
    for _ in range(reps):
        s = torch.cuda.Event(enable_timing=True)
        e = torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(iters):
            fn()
        e.record()
        torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / iters)
    return best


def report(section, name, ms, flops=None, bytes_=None, min_bytes=None, **extra):
    """bytes_: what the kernels of this call move by construction; min_bytes: SURVEY.md 8(d)'s algorithmic minimum for the operation (one
    read + one write for a normalisation forward; read x, read dy, write dx for its backward) -- reported as gbps_min / hbm_frac_min."""
    r = {'section': section, 'name': name, 'ms': round(ms, 5)}
    if flops is not None:
        r['tflops'] = round(flops / ms / 1e9, 2)
        r['mfma_frac'] = round(flops / ms / 1e9 / PEAK_TF, 4)
    if bytes_ is not None:
        r['gbps'] = round(bytes_ / ms / 1e6, 1)
        r['hbm_frac'] = round(bytes_ / ms / 1e6 / PEAK_HBM, 4)
    if min_bytes is not None:
        r['gbps_min'] = round(min_bytes / ms / 1e6, 1)
        r['hbm_frac_min'] = round(min_bytes / ms / 1e6 / PEAK_HBM, 4)
    r.update(extra)
    RESULTS.append(r)
    print(json.dumps(r), flush=True)


def rand_cl(n, c, t, h, w, scale=1.0):
    x = empty_cl(n, c, t, h, w, 'cuda')
    if os.environ.get('MB_ZERO', '') == '1':              # all-zero operands: the same instruction stream at a fraction of the switching
        x.zero_()                                         # power -- how far the DVFS power cap (not the schedule) holds a kernel back
    else:
        x.copy_(torch.randn(n, c, t, h, w, device='cuda') * scale)
    return x


# ------------------------------------------------------------------------------------------------
def bench_attn(iters, only=None):
    lib = _hip.load_library()
    P = _hip.ptr
    cases = [
        # name, B, T, H, W, n_head, d_head, mode
        ('repr_tok spatial S=256 C=512', 8, 16, 16, 16, 8, 64, 'space'),
        ('yaml_tok spatial S=1024 C=512', 2, 16, 32, 32, 8, 64, 'space'),
        ('lam spatial S=4096 C=256', 1, 16, 64, 64, 4, 64, 'space'),
        # the same two shapes at the batch BASELINE configs[0] quotes (4 clips): 2 / 4 rounds of workgroups become 4 / 16 -- what a training step
        # launches; the 1- and 2-clip lines above stay for continuity with rounds 1-5 (their grids are mostly ramp and tail)
        ('yaml_tok spatial S=1024 C=512 (4 clips)', 4, 16, 32, 32, 8, 64, 'space'),
        ('lam spatial S=4096 C=256 (4 clips)', 4, 16, 64, 64, 4, 64, 'space'),
        ('lam spatial S=1024 C=256', 4, 16, 32, 32, 4, 64, 'space'),
        ('dynamics spatial S=64 C=512', 32, 16, 8, 8, 8, 64, 'space'),
        ('repr_tok temporal T=16 C=512', 8, 16, 16, 16, 8, 64, 'time'),
        ('dynamics temporal T=16 C=512', 32, 16, 8, 8, 8, 64, 'time'),
        ('lam temporal T=16 C=256', 1, 16, 64, 64, 4, 64, 'time'),
    ]
    if os.environ.get('MB_ATTN_CASES'):                    # "name,b,t,h,w,n_head,d_head,mode;..." replaces the table
        cases = [(f[0], *[int(v) for v in f[1:7]], f[7]) for f in (c.split(',') for c in os.environ['MB_ATTN_CASES'].split(';'))]
    for name, b, t, h, w, nh, dh, mode in cases:
        if only is not None and not any(o in name for o in only):
            continue
        c = nh * dh
        hw, ntok = h * w, b * t * h * w
        x = rand_cl(b, c, t, h, w)
        u = torch.empty_like(x)
        if mode == 'space':
            nseq, S, qmap, pos_div, pos_mod, causal = b * t, hw, (1, hw * c, 0, c), 1, hw, 0
        else:
            nseq, S, qmap, pos_div, pos_mod, causal = b * hw, t, (hw, t * hw * c, c, hw * c), hw, t, 1
        table = torch.randn(pos_mod, c, device='cuda')
        gamma = torch.ones(c, device='cuda'); beta = torch.zeros(c, device='cuda')
        stats = torch.empty(ntok * 2, device='cuda')
        out = torch.empty_like(x); oattn = torch.empty_like(x)
        lse = torch.empty(ntok * nh, device='cuda')
        D = torch.empty(3 * ntok * nh, device='cuda')                    # D, lse * log2 e, -D (ABI 10)
        dout = rand_cl(b, c, t, h, w, 0.1)
        du = torch.empty_like(x); dx = torch.empty_like(x)
        dg = torch.zeros(c, device='cuda'); db = torch.zeros(c, device='cuda')
        scale = nh * dh ** -0.5 / 8.0           # keep softmax away from saturation on random data
        qm = _hip.i64(qmap)
        s = _hip.stream_ptr()

        def f_ln():
            _hip.check(lib.genie_rotary_layernorm_fwd(P(x), P(u), ntok, c, c, P(table), pos_div, pos_mod, P(gamma), P(beta), 1e-5, P(stats), s), 'ln')

        def f_fwd():
            _hip.check(lib.genie_attention_fwd(P(u), P(u), P(u), None if NO_RESID else P(x), P(out), None if NO_OATTN else P(oattn), None if NO_LSE else P(lse),
                                               nseq, nh, dh, S, S, qm, qm, qm, scale, causal, c, s), 'fwd')

        def f_bwd():
            _hip.check(lib.genie_attention_bwd(P(u), P(u), P(u), P(oattn), None, P(dout), P(lse), P(D), P(du), None, None, nseq, nh, dh, S, S,
                                               qm, qm, qm, None, scale, causal, c, ntok, s), 'bwd')

        def f_lnb():
            _hip.check(lib.genie_rotary_layernorm_bwd(P(x), P(du), P(dout), P(dx), ntok, c, c, P(table), pos_div, pos_mod, P(gamma), P(stats), P(dg), P(db), s), 'lnb')

        tok_bytes = ntok * c * 2
        fl = 4.0 * S * S * c * nseq                      # dense count (causal masking not credited)
        report('attn', name + ' | rotary+LN fwd', timeit(f_ln, iters), bytes_=2 * tok_bytes)
        report('attn', name + ' | attention fwd', timeit(f_fwd, iters), flops=fl, bytes_=4 * tok_bytes, S=S, nseq=nseq)
        report('attn', name + ' | attention bwd', timeit(f_bwd, iters), flops=2.5 * fl, bytes_=5 * tok_bytes, S=S, nseq=nseq)
        report('attn', name + ' | rotary+LN bwd', timeit(f_lnb, iters), bytes_=4 * tok_bytes)
        del x, u, out, oattn, dout, du, dx
        torch.cuda.empty_cache()


# ------------------------------------------------------------------------------------------------
def bench_hbm(iters, quick=False, B=None):
    """B: clips per call (default MB_BATCH or 8).  8 clips = 134 MB per 128-channel tensor -- half of it fits the 256-MB last-level cache,
    which flatters write streams and penalises launches with a fixed tail; bench.py reports B = 8 AND its own batch (64 clips, 1.07 GB)."""
    lib = _hip.load_library()
    P = _hip.ptr
    B = int(B if B is not None else os.environ.get('MB_BATCH', 8))
    for (c, t, h, w, g) in ([(128, 16, 64, 64, 1)] if quick else [(128, 16, 64, 64, 1), (256, 16, 32, 32, 1), (512, 4, 8, 8, 1), (128, 16, 64, 64, 8)]):
        x = rand_cl(B, c, t, h, w)
        gamma = torch.ones(c, device='cuda', requires_grad=True); beta = torch.zeros(c, device='cuda', requires_grad=True)
        nbytes = B * c * t * h * w * 2
        y = None

        def f():
            nonlocal y
            y = GF.group_norm(x, g, gamma, beta, 1e-5, act=True)
        report('hbm', f'GroupNorm+SiLU fwd C={c} {t}x{h}x{w} G={g} B={B} (alg: 2 reads + 1 write)', timeit(f, iters), bytes_=3 * nbytes, min_bytes=2 * nbytes)
        xr = x.detach().requires_grad_(True)
        yy = GF.group_norm(xr, g, gamma, beta, 1e-5, act=True)
        dy = rand_cl(B, c, t, h, w)

        def fb():
            torch.autograd.grad(yy, [xr], [dy], retain_graph=True)
        report('hbm', f'GroupNorm+SiLU bwd C={c} {t}x{h}x{w} G={g} B={B} (alg: 4 reads + 1 write)', timeit(fb, iters), bytes_=5 * nbytes, min_bytes=3 * nbytes)
        del x, xr, yy, dy, y
    # stem / head CausalConv3d
    from genie.module.video import CausalConv3d
    stem = CausalConv3d(3, 128, 3).cuda()
    head = CausalConv3d(128, 3, 3).cuda()
    vid = to_cl(torch.randn(B, 3, 16, 64, 64, device='cuda'))
    feat = rand_cl(B, 128, 16, 64, 64)
    npx = B * 16 * 64 * 64
    # 17.19 MB per clip = (3 + 128) channels x 2 B x 65536 pixels + the 20.7-kB weight tensor; the kernels move the 3-channel tensor at its
    # padded 8-channel pitch (pricing that pitch flattered the line by 3.8 %) -- reported beside it as gbps_padded_pitch
    nb, nb_pad = npx * (3 + 128) * 2 + 128 * 3 * 27 * 2, npx * (8 + 128) * 2
    pad = lambda ms: {'gbps_padded_pitch': round(nb_pad / ms / 1e6, 1), 'hbm_frac_padded_pitch': round(nb_pad / ms / 1e6 / PEAK_HBM, 4)}
    with torch.no_grad():
        ms = timeit(lambda: stem(vid), iters)
        report('hbm', f'stem CausalConv3d 3->128 k3 B={B} fwd', ms, flops=2.0 * npx * 128 * 3 * 27, bytes_=nb, **pad(ms))
        ms = timeit(lambda: head(feat), iters)
        report('hbm', f'head CausalConv3d 128->3 k3 B={B} fwd', ms, flops=2.0 * npx * 128 * 3 * 27, bytes_=nb, **pad(ms))
        # weight gradients of the two narrow convs: one pass over the 128-channel tensor (+ the 8-channel-pitch narrow one)
        from genie.conv import conv_wgrad
        gy = rand_cl(B, 128, 16, 64, 64)
        dws = torch.zeros_like(stem.conv3d.weight); dbs = torch.zeros(128, device='cuda')
        ms = timeit(lambda: conv_wgrad(vid, gy, stem.conv3d.spec, dws, dbs), iters)
        report('hbm', f'stem CausalConv3d 3->128 k3 B={B} wgrad', ms, flops=2.0 * npx * 128 * 3 * 27, bytes_=nb, **pad(ms))
        g3 = to_cl(torch.randn(B, 3, 16, 64, 64, device='cuda'))
        dwh = torch.zeros_like(head.conv3d.weight); dbh = torch.zeros(3, device='cuda')
        ms = timeit(lambda: conv_wgrad(feat, g3, head.conv3d.spec, dwh, dbh), iters)
        report('hbm', f'head CausalConv3d 128->3 k3 B={B} wgrad', ms, flops=2.0 * npx * 128 * 3 * 27, bytes_=nb, **pad(ms))
    if quick:
        return
    # layout conversion + mse
    v32 = torch.randn(B, 3, 16, 64, 64, device='cuda')
    report('hbm', f'to_channels_last fp32->bf16 3ch B={B}', timeit(lambda: to_cl(v32), iters), bytes_=npx * (12 + 16))
    # AdamW over a 375.6M-parameter arena
    n = 375_554_840
    p = torch.zeros(n, device='cuda'); gr = torch.zeros(n, device='cuda'); m = torch.zeros(n, device='cuda'); v = torch.zeros(n, device='cuda')
    report('hbm', 'AdamW 375.6M params (alg: 28 B/param)',
           timeit(lambda: _hip.check(lib.genie_adamw_step(P(p), P(gr), P(m), P(v), n, 1e-3, 0.9, 0.999, 1e-8, 0.01, 1, 1.0, 1, _hip.stream_ptr()), 'adamw'), iters),
           bytes_=28.0 * n)
    # plain device copy as the achievable-bandwidth yardstick
    a = torch.empty(1 << 28, dtype=torch.float32, device='cuda'); bb = torch.empty_like(a)
    report('hbm', 'torch copy 1 GiB (yardstick: read + write)', timeit(lambda: bb.copy_(a), iters), bytes_=2.0 * a.numel() * 4)
    # one-directional yardsticks: the stem conv is a WRITE stream (16.8 MB out per 0.39 MB in), the head conv and the weight gradients READ streams
    report('hbm', 'torch fill 1 GiB (yardstick: write only)', timeit(lambda: bb.fill_(1.0), iters), bytes_=1.0 * a.numel() * 4)
    a16 = a.view(torch.bfloat16)
    report('hbm', 'torch sum 1 GiB bf16 (yardstick: read only)', timeit(lambda: a16.sum(dtype=torch.float32), iters), bytes_=1.0 * a.numel() * 4)


# ------------------------------------------------------------------------------------------------
def bench_conv(iters):
    from genie.conv import ConvSpec, causal_spec, conv_dgrad, conv_forward, conv_wgrad, pack_weight_bwd, pack_weight_fwd, same_spec
    B = int(os.environ.get('MB_BATCH', 8))
    shapes = [
        ('res 128->128 k3 @16x64x64', same_spec(128, 128, (3, 3, 3)), (16, 64, 64)),
        ('res 256->256 k3 @16x32x32', same_spec(256, 256, (3, 3, 3)), (16, 32, 32)),
        ('res 256->256 k3 @8x16x16', same_spec(256, 256, (3, 3, 3)), (8, 16, 16)),
        ('res 512->512 k3 @4x8x8', same_spec(512, 512, (3, 3, 3)), (4, 8, 8)),
        ('res 128->256 k3 @16x32x32', same_spec(128, 256, (3, 3, 3)), (16, 32, 32)),
        ('1x1 128->256 @16x32x32', same_spec(128, 256, (1, 1, 1)), (16, 32, 32)),
        ('down 128->128 s(1,2,2) @16x64x64', causal_spec(128, 128, (3, 3, 3), (1, 2, 2)), (16, 64, 64)),
        ('down 256->256 s(2,2,2) @16x32x32', causal_spec(256, 256, (3, 3, 3), (2, 2, 2)), (16, 32, 32)),
        ('up 512->4096 (2,2,2) @4x8x8', causal_spec(512, 4096, (3, 3, 3), shuffle=(2, 2, 2)), (4, 8, 8)),
        ('up 256->2048 (2,2,2) @8x16x16', causal_spec(256, 2048, (3, 3, 3), shuffle=(2, 2, 2)), (8, 16, 16)),
        ('up 256->1024 (1,2,2) @16x32x32', causal_spec(256, 1024, (3, 3, 3), shuffle=(1, 2, 2)), (16, 32, 32)),
    ]
    for name, spec, size in shapes:
        if FILTER and FILTER not in name:
            continue
        t, h, w = size
        x = rand_cl(B, spec.cin, t, h, w)
        wt = (torch.randn(spec.cout, spec.cin, *spec.kernel, device='cuda') * (0.0 if os.environ.get('MB_ZERO', '') == '1' else 0.05)).contiguous(memory_format=torch.channels_last_3d)
        wf, wb = pack_weight_fwd(wt, spec), pack_weight_bwd(wt, spec)
        y = conv_forward(x, wf, None, spec)
        dy = y                                   # any CL tensor of the output's shape serves as the gradient
        dw = torch.zeros_like(wt)
        to, ho, wo = spec.out_size(size)
        fl = 2.0 * B * to * ho * wo * spec.cout * spec.cin * spec.ntaps
        lib = _hip.load_library()
        ms = timeit(lambda: conv_forward(x, wf, None, spec), iters)
        report('conv', name + ' | fwd', ms, flops=fl, kernel=gconv.VARIANT_NAMES.get(lib.genie_last_conv_variant()))
        # upsample convs: the training step un-shuffles the output gradient ONCE (genie_unshuffle_cl) and hands it to both the backward-data
        # and the weight-gradient pass (functional._Conv3dFn.backward) -- timed the same way here, the un-shuffle as its own line
        # (round 3 timed the gather-through-the-shuffle kernels the step no longer uses: 0.32-0.37 where the step runs 0.48)
        dyu = None
        if spec.shuffle is not None and gconv.wgrad_unshuffled_ok(spec, x):
            dyu = gconv.unshuffle_dy(dy, spec)
            ms = timeit(lambda: gconv.unshuffle_dy(dy, spec), iters)
            report('conv', name + ' | un-shuffle of dy (shared by dgrad and wgrad)', ms, bytes_=2.0 * dy.numel() * 2)
        ms = timeit(lambda: conv_dgrad(dy, wb, spec, size, dy_unshuffled=dyu), iters)
        report('conv', name + ' | dgrad', ms, flops=fl, kernel=gconv.VARIANT_NAMES.get(lib.genie_last_conv_variant()))
        if not NO_WGRAD:
            ms = timeit(lambda: conv_wgrad(x, dy if dyu is None else dyu, spec, dw, None, dyu is not None), iters)
            report('conv', name + ' | wgrad', ms, flops=fl, kernel=gconv.VARIANT_NAMES.get(lib.genie_last_conv_variant()))
            # the training step's convs have a bias: its gradient (column sums of dy) rides in the same launch
            db = torch.zeros(spec.cout, device='cuda')
            ms = timeit(lambda: conv_wgrad(x, dy if dyu is None else dyu, spec, dw, db, dyu is not None), iters)
            report('conv', name + ' | wgrad + bias gradient', ms, flops=fl, kernel=gconv.VARIANT_NAMES.get(lib.genie_last_conv_variant()))
        del x, y, dw, wf, wb, wt
        torch.cuda.empty_cache()
    if FILTER:
        return
    # MaskGIT vocabulary head: Linear(512 -> 2^18) over B*T*h*w = 4096 tokens (B=4)
    spec = ConvSpec(512, 1 << 18, (1, 1, 1))
    x = rand_cl(4, 512, 16, 8, 8)
    wt = (torch.randn(1 << 18, 512, 1, 1, 1, device='cuda') * 0.05)
    wf = pack_weight_fwd(wt, spec)
    report('conv', 'dynamics head 512->262144, 4096 tokens | fwd', timeit(lambda: conv_forward(x, wf, None, spec), max(3, iters // 4)),
           flops=2.0 * 4096 * 512 * (1 << 18))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('sections', nargs='*', default=['attn', 'hbm', 'conv'])
    ap.add_argument('--iters', type=int, default=20)
    ap.add_argument('--out', default=os.path.join(ROOT, 'gpurun_out', 'microbench.json'))
    args = ap.parse_args()
    assert torch.cuda.is_available()
    t0 = time.time()
    for sec in args.sections:
        try:
            {'attn': bench_attn, 'hbm': bench_hbm, 'conv': bench_conv}[sec](args.iters)
        except Exception as ex:           # keep the other sections' numbers
            print(f'section {sec} failed: {type(ex).__name__}: {ex}', flush=True)
            RESULTS.append({'section': sec, 'error': f'{type(ex).__name__}: {ex}'})
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, 'w') as f:
        json.dump({'device': torch.cuda.get_device_name(0), 'wall_s': round(time.time() - t0, 1), 'results': RESULTS}, f, indent=1)


if __name__ == '__main__':
    main()
