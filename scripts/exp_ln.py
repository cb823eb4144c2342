"""Timing experiment for the rotary + LayerNorm kernels: variants with / without the rotary table, residual and dgamma."""
import os, sys, json, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'open-genie_amd'))
from genie import _hip
lib = _hip.load_library(); P = _hip.ptr

def timeit(f, iters=50):
    for _ in range(5): f()
    best = 1e9
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(iters): f()
        b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / iters)
    return best * 1e3

for ntok, c, pos_mod in ((65536, 512, 256), (8192, 512, 256), (65536, 256, 4096)):
    x = torch.randn(ntok, c, device='cuda').to(torch.bfloat16); u = torch.empty_like(x); du = torch.randn_like(x); dr = torch.randn_like(x); dx = torch.empty_like(x)
    table = torch.randn(pos_mod, c, device='cuda'); gamma = torch.ones(c, device='cuda'); beta = torch.zeros(c, device='cuda')
    stats = torch.empty(ntok * 2, device='cuda'); dg = torch.zeros(c, device='cuda'); db = torch.zeros(c, device='cuda')
    s = _hip.stream_ptr()
    res = {'ntok': ntok, 'C': c}
    res['copy'] = timeit(lambda: u.copy_(x))
    res['noop_call'] = timeit(lambda: _hip.check(lib.genie_rotary_layernorm_fwd(P(x), P(u), 0, c, c, P(table), 1, pos_mod, P(gamma), P(beta), 1e-5, P(stats), s), 'ln'))
    res['fwd'] = timeit(lambda: _hip.check(lib.genie_rotary_layernorm_fwd(P(x), P(u), ntok, c, c, P(table), 1, pos_mod, P(gamma), P(beta), 1e-5, P(stats), s), 'ln'))
    res['fwd_norot'] = timeit(lambda: _hip.check(lib.genie_rotary_layernorm_fwd(P(x), P(u), ntok, c, c, None, 1, pos_mod, P(gamma), P(beta), 1e-5, P(stats), s), 'ln'))
    res['fwd_norot_nogb'] = timeit(lambda: _hip.check(lib.genie_rotary_layernorm_fwd(P(x), P(u), ntok, c, c, None, 1, pos_mod, None, None, 1e-5, P(stats), s), 'ln'))
    res['bwd'] = timeit(lambda: _hip.check(lib.genie_rotary_layernorm_bwd(P(x), P(du), P(dr), P(dx), ntok, c, c, P(table), 1, pos_mod, P(gamma), P(stats), P(dg), P(db), s), 'lnb'))
    res['bwd_nodg'] = timeit(lambda: _hip.check(lib.genie_rotary_layernorm_bwd(P(x), P(du), P(dr), P(dx), ntok, c, c, P(table), 1, pos_mod, P(gamma), P(stats), None, None, s), 'lnb'))
    res['bwd_norot_nores'] = timeit(lambda: _hip.check(lib.genie_rotary_layernorm_bwd(P(x), P(du), None, P(dx), ntok, c, c, None, 1, pos_mod, P(gamma), P(stats), P(dg), P(db), s), 'lnb'))
    print(json.dumps({k: (round(v, 2) if isinstance(v, float) else v) for k, v in res.items()}))
