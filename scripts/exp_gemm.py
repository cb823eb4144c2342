"""Plain-GEMM shapes (ST-block FFN, MaskGIT vocabulary head) through the gather-GEMM conv kernels vs torch.matmul (hipBLASLt)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'open-genie_amd')]
import torch
from genie import conv as gconv
from genie.conv import same_spec
from scripts.microbench import timeit, rand_cl
for (name, m_shape, cin, cout) in [('dyn head 4096 tok', (4, 16, 8, 8), 512, 1 << 18), ('ffn up 4096 tok', (4, 16, 8, 8), 512, 2048), ('ffn down 4096 tok', (4, 16, 8, 8), 2048, 512),
                                   ('ffn up 65536 tok', (8, 16, 16, 32), 512, 2048), ('ffn down 65536 tok', (8, 16, 16, 32), 2048, 512)]:
    b, t, h, w = m_shape
    M = b * t * h * w
    spec = same_spec(cin, cout, (1, 1, 1))
    x = rand_cl(b, cin, t, h, w); dy = rand_cl(b, cout, t, h, w, 0.1)
    wt = (torch.randn(cout, cin, 1, 1, 1, device='cuda') * 0.02)
    fl = 2.0 * M * cin * cout
    res = {}
    try:
        wp = gconv.pack_weight_fwd(wt, spec); wd = gconv.pack_weight_bwd(wt, spec)
        res['fwd'] = timeit(lambda: gconv.conv_forward(x, wp, None, spec), 10)
        res['dgrad'] = timeit(lambda: gconv.conv_dgrad(dy, wd, spec, (t, h, w)), 10)
    except Exception as e:
        res['err'] = repr(e)[:200]
    dw = torch.zeros(cout, cin, 1, 1, 1, device='cuda').contiguous(memory_format=torch.channels_last_3d)
    try:
        res['wgrad'] = timeit(lambda: gconv.conv_wgrad(x, dy, spec, dw, None), 10)
    except Exception as e:
        res['werr'] = repr(e)[:200]
    x2 = x.permute(0, 2, 3, 4, 1).reshape(M, cin); dy2 = dy.permute(0, 2, 3, 4, 1).reshape(M, cout); w2 = wt.reshape(cout, cin).to(torch.bfloat16)
    res['torch_fwd'] = timeit(lambda: x2 @ w2.t(), 10)
    res['torch_dgrad'] = timeit(lambda: dy2 @ w2, 10)
    res['torch_wgrad'] = timeit(lambda: dy2.t() @ x2, 10)
    print(name, {k: (f'{v:.3f} ms {fl / v / 1e9:.0f} TF' if isinstance(v, float) else v) for k, v in res.items()})
