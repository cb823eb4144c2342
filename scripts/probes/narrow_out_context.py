"""Probe: the head conv's forward (128 -> 3, 64 clips) timed alone, after two launches of the dominant MFMA kernel, and right after the kernel that WRITES its input
(GroupNorm + SiLU apply, as in the training step; a device copy into the input as a second writer).  HIP events bracket the head-forward launch only."""
import sys
import torch
sys.path[:0] = ['/root/repo', '/root/repo/open-genie_amd', '/root/repo/scripts']
from genie import _hip, cl, conv as gconv
lib = _hip.load_library()
P = _hip.ptr
B, T, H, W = 64, 16, 64, 64
npx = B * T * H * W
torch.manual_seed(0)
pack = gconv.pack_narrow_out(torch.randn(3, 128, 3, 3, 3, device='cuda') * 0.05)
bias = torch.zeros(3, device='cuda')
src = cl.to_cl(torch.randn(B, 128, T, H, W, device='cuda'))
feat = cl.empty_like_cl(src)
feat.copy_(src)
out = torch.empty(npx * 8, device='cuda', dtype=torch.bfloat16)
g_, b_ = torch.ones(128, device='cuda'), torch.zeros(128, device='cuda')
mean, rstd = torch.empty(B, device='cuda'), torch.empty(B, device='cuda')
ws = torch.empty(lib.genie_groupnorm_ws_floats(B, 128, 1), device='cuda')
s = _hip.stream_ptr()
head = lambda: _hip.check(lib.genie_conv_narrow_out(feat.data_ptr(), pack.data_ptr(), bias.data_ptr(), out.data_ptr(), B, T, H, W, 3, -2, s), 'narrow_out')
gn = lambda: _hip.check(lib.genie_groupnorm_fwd(P(src), P(feat), B, T * H * W, 128, cl.pitch_of(src), 1, P(g_), P(b_), None, None, 1e-5, 1, P(mean), P(rstd), P(ws), s), 'gn')
xc = cl.to_cl(torch.randn(B, 256, 16, 32, 32, device='cuda'))
spec = gconv.same_spec(256, 256, (3, 3, 3))
wpk = gconv.pack_weight_fwd(torch.randn(256, 256, 3, 3, 3, device='cuda') * 0.01, spec)
conv = lambda: gconv.conv_forward(xc, wpk, None, spec)
nb = npx * (3 + 128) * 2 + 128 * 3 * 27 * 2
pre = {'alone': lambda: None, 'after 2 MFMA launches': lambda: (conv(), conv()), 'after GN+SiLU writes its input': gn, 'after a copy into its input': lambda: feat.copy_(src),
       'after 2 MFMA launches + GN': lambda: (conv(), conv(), gn())}
for rep in range(2):
    for name, p in pre.items():
        for _ in range(4):
            p(); head()
        evs = []
        for _ in range(12):
            p()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); head(); e1.record()
            evs.append((e0, e1))
        torch.cuda.synchronize()
        ms = sum(a.elapsed_time(b) for a, b in evs) / len(evs)
        print(f'{name:34s}: {ms:.4f} ms = {nb / ms / 1e6 / 8000:.3f} of 8 TB/s')
