#!/usr/bin/env python
"""Same-box A/B of the 256 x 256 kw-triple conv kernel in its two wave decompositions on the step's dominant launch shapes (64 clips):
igemm3w_kernel (8 waves of 64 x 128, two per SIMD) vs igemm3x_kernel (4 waves of 128 x 128, one per SIMD; conv_igemm3x.hip), random and
all-zero operands (the power-cap reading), interleaved rounds.   python scripts/ab_igemm3x.py [--batch 64] [--rounds 5]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'open-genie_amd')):
    if p not in sys.path:
        sys.path.insert(0, p)
import torch   # noqa: E402

from genie import conv as gconv   # noqa: E402
import scripts.microbench as mb   # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=64)
    ap.add_argument('--rounds', type=int, default=5)
    args = ap.parse_args()
    B = args.batch
    shapes = [('256->256 k3 @16x32x32', 256, 256, (16, 32, 32)), ('512->512 k3 @8x16x16 (REPR-like)', 512, 512, (8, 16, 16)), ('256->256 k3 @8x16x16', 256, 256, (8, 16, 16))]
    for name, ci, co, (t, h, w) in shapes:
        spec = gconv.same_spec(ci, co, (3, 3, 3))
        fl = 2.0 * B * t * h * w * ci * co * 27
        for tag, zero in (('random', False), ('zeros', True)):
            x = mb.empty_cl(B, ci, t, h, w, 'cuda')
            x.zero_() if zero else x.copy_(torch.randn(B, ci, t, h, w, device='cuda'))
            wt = (torch.randn(co, ci, 3, 3, 3, device='cuda') * (0.0 if zero else 0.05)).contiguous(memory_format=torch.channels_last_3d)
            wf = gconv.pack_weight_fwd(wt, spec)
            res = {0: [], 4096: []}
            for _ in range(args.rounds):
                for flag in (0, 4096):
                    gconv.TRI_FLAGS = flag
                    res[flag].append(mb.timeit(lambda: gconv.conv_forward(x, wf, None, spec), 8))
            gconv.TRI_FLAGS = 0
            med = {k: sorted(v)[len(v) // 2] for k, v in res.items()}
            print(json.dumps({'layer': f'{name}, {B} clips, forward', 'operands': tag,
                              'igemm3w_8waves_ms': round(med[0], 4), 'igemm3w_tflops': round(fl / med[0] / 1e9, 1), 'igemm3w_frac': round(fl / med[0] / 1e9 / 2500, 4),
                              'igemm3x_4waves_ms': round(med[4096], 4), 'igemm3x_tflops': round(fl / med[4096] / 1e9, 1), 'igemm3x_frac': round(fl / med[4096] / 1e9 / 2500, 4),
                              'x_over_w': round(med[0] / med[4096], 4)}), flush=True)
            del x, wt, wf
            torch.cuda.empty_cache()


if __name__ == '__main__':
    main()
