"""Probe: does the head conv's forward (conv_narrow_out2_kernel, 128 -> 3 channels, 64 clips) depend on WHERE its input and output sit in memory?
Input and output are carved out of large buffers at chosen byte offsets (round 6: the bench's figure moved 0.59 -> 0.54 when the microbench in front of
it changed and with it the caching allocator's placement)."""
import sys
import torch
sys.path[:0] = ['/root/repo', '/root/repo/open-genie_amd', '/root/repo/scripts']
from genie import _hip, conv as gconv
lib = _hip.load_library()
B, T, H, W = 64, 16, 64, 64
npx = B * T * H * W
torch.manual_seed(0)
wt = torch.randn(3, 128, 3, 3, 3, device='cuda') * 0.05
pack = gconv.pack_narrow_out(wt)
bias = torch.zeros(3, device='cuda')
xbuf = torch.empty(npx * 128 * 2 + (64 << 20), dtype=torch.uint8, device='cuda')
obuf = torch.empty(npx * 8 * 2 + (64 << 20), dtype=torch.uint8, device='cuda')
xbuf.view(torch.bfloat16)[:].normal_()
base_x, base_o = xbuf.data_ptr(), obuf.data_ptr()
print('buffer bases mod 2 MiB:', base_x % (2 << 20), base_o % (2 << 20))
s = _hip.stream_ptr()

def run(ox, oo, iters=20):
    f = lambda: _hip.check(lib.genie_conv_narrow_out(base_x + ox, pack.data_ptr(), bias.data_ptr(), base_o + oo, B, T, H, W, 3, -2, s), 'narrow_out')
    for _ in range(3):
        f()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            f()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters)
    return best

nb = npx * (3 + 128) * 2 + 128 * 3 * 27 * 2
offs = [0, 256, 4096, 65536, 1 << 20, (1 << 20) + 4096, 2 << 20, (2 << 20) + 256, 16 << 20, (16 << 20) + (1 << 20)]
for ox in offs:
    row = []
    for oo in (0, 256, 4096, 1 << 20, (1 << 20) + 4096):
        ms = run(ox, oo)
        row.append(f'{nb / ms / 1e6 / 8000:.3f}')
    print(f'x +{ox:>9d}: out +0/+256/+4K/+1M/+1M4K ->', ' '.join(row))
