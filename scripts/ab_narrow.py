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

which = sys.argv[1:] or ['head_fwd', 'stem_fwd', 'stem_wgrad', 'head_wgrad', 'stem_wgrad_k', 'head_wgrad_k']
from genie import _hip
from genie.cl import pitch_of
lib = _hip.load_library()
tag = os.environ.get('AB_TAG', '')
stem = CausalConv3d(3, 128, 3).cuda()
head = CausalConv3d(128, 3, 3).cuda()
for B in (8, 64):
    npx = B * 16 * 64 * 64
    nbytes = npx * (3 + 128) * 2 + 128 * 3 * 27 * 2      # 17.19 MB per clip ((3 + 128) channels + the weights), not the 8-channel pitch
    vid = to_cl(torch.randn(B, 3, 16, 64, 64, device='cuda'))
    feat = rand_cl(B, 128, 16, 64, 64)
    gy = rand_cl(B, 128, 16, 64, 64)
    g3 = to_cl(torch.randn(B, 3, 16, 64, 64, device='cuda'))
    dws = torch.zeros_like(stem.conv3d.weight); dbs = torch.zeros(128, device='cuda')
    dwh = torch.zeros_like(head.conv3d.weight); dbh = torch.zeros(3, device='cuda')
    fns = {'head_fwd': lambda: head(feat), 'stem_fwd': lambda: stem(vid),
           'stem_wgrad': lambda: conv_wgrad(vid, gy, stem.conv3d.spec, dws, dbs), 'head_wgrad': lambda: conv_wgrad(feat, g3, head.conv3d.spec, dwh, dbh),
           # the C entry point alone (no zeroing of G, no scatter into dW / db by torch)
           'stem_wgrad_k': lambda: _hip.check(lib.genie_conv_narrow_wgrad(gy.data_ptr(), vid.data_ptr(), pitch_of(vid), G.data_ptr(), B, 16, 64, 64, -2, 1, _hip.stream_ptr()), 'k'),
           'head_wgrad_k': lambda: _hip.check(lib.genie_conv_narrow_wgrad(feat.data_ptr(), g3.data_ptr(), pitch_of(g3), G.data_ptr(), B, 16, 64, 64, 0, 0, _hip.stream_ptr()), 'k')}
    G = torch.zeros(128, 128, device='cuda')
    def acc(stem_, bias, wcl):
        big, small = (gy, vid) if stem_ else (feat, g3)
        dw = (dws if stem_ else dwh); db = (dbs if stem_ else dbh)
        return lambda: _hip.check(lib.genie_conv_narrow_wgrad_acc(big.data_ptr(), small.data_ptr(), pitch_of(small), dw.data_ptr(), db.data_ptr() if bias else None,
                                                                    B, 16, 64, 64, -2 if stem_ else 0, int(stem_), 3, wcl, _hip.stream_ptr()), 'acc')
    for st in (1, 0):
        for bias in (1, 0):
            for wcl in (1, 0):
                fns[f'acc_stem{st}_bias{bias}_wcl{wcl}'] = acc(st, bias, wcl)
    with torch.no_grad():
        for k in which:
            ms = timeit(fns[k], 30 if B == 8 else 10)
            print(json.dumps({'tag': tag, 'case': k, 'B': B, 'ms': round(ms, 5), 'gbps': round(nbytes / ms / 1e6, 1), 'hbm_frac': round(nbytes / ms / 1e6 / 8000.0, 4)}), flush=True)
    del vid, feat, gy, g3
