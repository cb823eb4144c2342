import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'open-genie_amd')]
import torch
from genie import _hip, conv as gconv
from genie.conv import same_spec
from scripts.microbench import timeit, rand_cl
B = 8
orig = _hip.GenieWgradDesc
for (cin, cout, size, k) in [(128, 128, (16, 64, 64), 1), (256, 256, (16, 32, 32), 1), (512, 512, (4, 8, 8), 1), (512, 512, (4, 8, 8), 3), (256, 256, (8, 16, 16), 1)]:
    spec = same_spec(cin, cout, (k, k, k))
    x = rand_cl(B, cin, *size); dy = rand_cl(B, cout, *size)
    dw = torch.zeros(cout, cin, k, k, k, device='cuda').contiguous(memory_format=torch.channels_last_3d)
    for sk in [0, 16, 32, 64, 128, 256, 512, 1024]:
        gconv.FORCE_SPLIT_K = sk
        try:
            ms = timeit(lambda: gconv.conv_wgrad(x, dy, spec, dw, None), 20)
        except Exception as e:
            print('err', e); continue
        fl = 2.0 * B * size[0] * size[1] * size[2] * cin * cout * k ** 3
        print(f'{cin}->{cout} k{k} @{size} split_k={sk:5d}: {ms:.4f} ms  {fl/ms/1e9:7.1f} TF  {(x.numel()+dy.numel())*2/ms/1e6:7.0f} GB/s')
