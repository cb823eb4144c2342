"""Probe: GroupNorm + SiLU (128 channels, 16x64x64, 64 clips) timed alone vs interleaved with the dominant MFMA kernel (256 -> 256 conv @16x32x32), i.e.
at the clocks the power cap leaves inside a training step.  HIP events bracket the GroupNorm launches only."""
import sys
import torch
sys.path[:0] = ['/root/repo', '/root/repo/open-genie_amd', '/root/repo/scripts']
from genie import _hip, cl, conv as gconv
from genie import functional as GF
lib = _hip.load_library()
P = _hip.ptr
B = 64
torch.manual_seed(0)
x = cl.to_cl(torch.randn(B, 128, 16, 64, 64, device='cuda'))
y, dx = cl.empty_like_cl(x), cl.empty_like_cl(x)
dy = cl.to_cl(torch.randn(B, 128, 16, 64, 64, device='cuda'))
npix, cp = 16 * 64 * 64, cl.pitch_of(x)
g_, b_ = torch.ones(128, device='cuda'), torch.zeros(128, device='cuda')
mean, rstd = torch.empty(B, device='cuda'), torch.empty(B, device='cuda')
ws = torch.empty(lib.genie_groupnorm_ws_floats(B, 128, 1), device='cuda')
dg, db = torch.zeros(128, device='cuda'), torch.zeros(128, device='cuda')
s = _hip.stream_ptr()
gn_f = lambda: _hip.check(lib.genie_groupnorm_fwd(P(x), P(y), B, npix, 128, cp, 1, P(g_), P(b_), None, None, 1e-5, 1, P(mean), P(rstd), P(ws), s), 'gn fwd')
gn_b = lambda: _hip.check(lib.genie_groupnorm_bwd(P(x), P(dy), P(dx), B, npix, 128, cp, 1, P(g_), P(b_), None, None, 1, P(mean), P(rstd), P(dg), P(db), None, None, P(ws), s), 'gn bwd')
# the dominant conv launch
xc = cl.to_cl(torch.randn(B, 256, 16, 32, 32, device='cuda'))
spec = gconv.same_spec(256, 256, (3, 3, 3))
wpk = gconv.pack_weight_fwd(torch.randn(256, 256, 3, 3, 3, device='cuda') * 0.01, spec)
conv = lambda: gconv.conv_forward(xc, wpk, None, spec)
nb_f, nb_b = 3 * B * npix * 256, 5 * B * npix * 256

def timed(fn, load, iters=12):
    for _ in range(4):
        if load: conv(); conv()
        fn()
    tot = 0.0
    evs = []
    for _ in range(iters):
        if load: conv(); conv()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record()
        evs.append((e0, e1))
    torch.cuda.synchronize()
    return sum(a.elapsed_time(b) for a, b in evs) / iters

gn_f(); gn_b(); conv(); torch.cuda.synchronize()
for rep in range(2):
    for load in (False, True):
        f, b = timed(gn_f, load), timed(gn_b, load)
        print(f'{"between MFMA launches" if load else "alone               "}: GN+SiLU fwd {f:.4f} ms = {nb_f / f / 1e6 / 8000:.3f} of 8 TB/s, bwd {b:.4f} ms = {nb_b / b / 1e6 / 8000:.3f}')
