#!/usr/bin/env python
"""GroupNorm + SiLU at the step's batch (64 clips, 128 channels, 16x64x64 = 16.8 MB per clip, 1.07 GB per tensor), forward and backward, as ONE
launch pair over all clips (statistics pass over 1.07 GB, then the apply pass re-reads it from HBM) versus CHUNKS of k clips launched pair by pair
(statistics of k clips, apply of the same k clips while they are still in the 256-MB Infinity Cache).  One JSON line per chunk size."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'open-genie_amd'), os.path.join(ROOT, 'scripts')):
    if p not in sys.path:
        sys.path.insert(0, p)
import torch
from microbench import rand_cl, timeit
from genie import functional as GF


def main():
    B = int(os.environ.get('GN_B', 64))
    shapes = [(128, 16, 64, 64), (256, 16, 32, 32)]
    for (c, t, h, w) in shapes:
        x = rand_cl(B, c, t, h, w)
        dy = rand_cl(B, c, t, h, w)
        gamma = torch.ones(c, device='cuda', requires_grad=True); beta = torch.zeros(c, device='cuda', requires_grad=True)
        nbytes = B * c * t * h * w * 2
        for k in (B, 32, 16, 8, 4, 2):
            if k > B:
                continue

            def fwd():
                with torch.no_grad():
                    for n0 in range(0, B, k):
                        GF.group_norm(x[n0:n0 + k], 1, gamma, beta, 1e-5, act=True)
            xr = [x[n0:n0 + k].detach().requires_grad_(True) for n0 in range(0, B, k)]
            ys = [GF.group_norm(xi, 1, gamma, beta, 1e-5, act=True) for xi in xr]

            def bwd():
                for i, n0 in enumerate(range(0, B, k)):
                    torch.autograd.grad(ys[i], [xr[i]], [dy[n0:n0 + k]], retain_graph=True)
            f_ms, b_ms = timeit(fwd, 6), timeit(bwd, 6)
            print(json.dumps({'C': c, 'thw': [t, h, w], 'clips': B, 'chunk': k, 'launch_pairs': B // k, 'fwd_ms': round(f_ms, 4), 'bwd_ms': round(b_ms, 4),
                              'fwd_GBps_of_3_passes': round(3 * nbytes / f_ms / 1e6, 1), 'bwd_GBps_of_5_passes': round(5 * nbytes / b_ms / 1e6, 1)}), flush=True)
            del xr, ys


if __name__ == '__main__':
    main()
