#!/usr/bin/env python
"""A few launches of ONE stride-1 3x3x3 conv layer (forward, optionally backward-data) on random operands -- the unit rocprofv3 / PMC passes and
environment-switch A/Bs are run on.   python scripts/conv_once.py --cin 256 --cout 256 --size 16,32,32 --batch 64 [--iters 8] [--dgrad 1]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'open-genie_amd')):
    if p not in sys.path:
        sys.path.insert(0, p)
import torch   # noqa: E402

from genie import _hip, conv as gconv   # noqa: E402
import scripts.microbench as mb         # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--cin', type=int, default=256)
ap.add_argument('--cout', type=int, default=256)
ap.add_argument('--size', default='16,32,32')
ap.add_argument('--batch', type=int, default=64)
ap.add_argument('--iters', type=int, default=8)
ap.add_argument('--dgrad', type=int, default=0)
args = ap.parse_args()
t, h, w = [int(v) for v in args.size.split(',')]
B, ci, co = args.batch, args.cin, args.cout
spec = gconv.same_spec(ci, co, (3, 3, 3))
x = mb.empty_cl(B, ci, t, h, w, 'cuda')
x.copy_(torch.randn(B, ci, t, h, w, device='cuda'))
wt = (torch.randn(co, ci, 3, 3, 3, device='cuda') * 0.05).contiguous(memory_format=torch.channels_last_3d)
wf = gconv.pack_weight_fwd(wt, spec)
fl = 2.0 * B * t * h * w * ci * co * 27
ms = mb.timeit(lambda: gconv.conv_forward(x, wf, None, spec), args.iters)
rec = {'layer': f'{ci}->{co} k3 @{t}x{h}x{w}, {B} clips', 'fwd_ms': round(ms, 4), 'fwd_tflops': round(fl / ms / 1e9, 1), 'fwd_kernel': gconv.VARIANT_NAMES.get(_hip.load_library().genie_last_conv_variant()),
       'TRI_DH_INNER': int(gconv.TRI_DH_INNER), 'algorithmic_MB': round((B * t * h * w * (ci + co) * 2 + ci * co * 27 * 2) / 1e6, 1)}
if args.dgrad:
    dy = mb.empty_cl(B, co, t, h, w, 'cuda')
    dy.copy_(torch.randn(B, co, t, h, w, device='cuda'))
    wb = gconv.pack_weight_bwd(wt, spec)
    ms2 = mb.timeit(lambda: gconv.conv_dgrad(dy, wb, spec, (t, h, w)), args.iters)
    rec.update({'dgrad_ms': round(ms2, 4), 'dgrad_tflops': round(fl / ms2 / 1e9, 1)})
print(json.dumps(rec), flush=True)
