#!/usr/bin/env python
"""Narrow convs of the tokenizer (stem 3 -> 128, head 128 -> 3; conv_narrow.hip) at 8 and 64 clips: ms and fraction of 8 TB/s on the bytes the
operation moves (one 128-channel tensor + one 8-channel-pitch tensor).  One JSON line per case; env switches are read by the library at first use,
so A/B = two processes:   GENIE_NARROW_OUT_CUT=1 python scripts/ab_narrow.py;  python scripts/ab_narrow.py"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'open-genie_amd')):
    if p not in sys.path:
        sys.path.insert(0, p)
import torch
from scripts.microbench import timeit, rand_cl
from genie.cl import to_cl
from genie.conv import conv_wgrad
from genie.module.video import CausalConv3d

which = sys.argv[1:] or ['head_fwd', 'stem_fwd', 'stem_wgrad', 'head_wgrad']
tag = os.environ.get('AB_TAG', '')
stem = CausalConv3d(3, 128, 3).cuda()
head = CausalConv3d(128, 3, 3).cuda()
for B in (8, 64):
    npx = B * 16 * 64 * 64
    nbytes = npx * (8 + 128) * 2
    vid = to_cl(torch.randn(B, 3, 16, 64, 64, device='cuda'))
    feat = rand_cl(B, 128, 16, 64, 64)
    gy = rand_cl(B, 128, 16, 64, 64)
    g3 = to_cl(torch.randn(B, 3, 16, 64, 64, device='cuda'))
    dws = torch.zeros_like(stem.conv3d.weight); dbs = torch.zeros(128, device='cuda')
    dwh = torch.zeros_like(head.conv3d.weight); dbh = torch.zeros(3, device='cuda')
    fns = {'head_fwd': lambda: head(feat), 'stem_fwd': lambda: stem(vid),
           'stem_wgrad': lambda: conv_wgrad(vid, gy, stem.conv3d.spec, dws, dbs), 'head_wgrad': lambda: conv_wgrad(feat, g3, head.conv3d.spec, dwh, dbh)}
    with torch.no_grad():
        for k in which:
            ms = timeit(fns[k], 30 if B == 8 else 10)
            print(json.dumps({'tag': tag, 'case': k, 'B': B, 'ms': round(ms, 5), 'gbps': round(nbytes / ms / 1e6, 1), 'hbm_frac': round(nbytes / ms / 1e6 / 8000.0, 4)}), flush=True)
    del vid, feat, gy, g3
