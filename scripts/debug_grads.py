import sys, os
sys.path.insert(0, 'tests'); sys.path.insert(0, '.'); sys.path.insert(0, 'open-genie_amd')
import torch
from util import bf16_round
import test_gpu_tokenizer as T
from oracle import genie_oracle as O

for enc, dec, d, shape in [(T.SMALL_ENC, T.SMALL_DEC, 6, (2, 3, 4, 16, 16)), (T.MID_ENC, T.MID_DEC, 10, (2, 3, 4, 32, 32))]:
    m, sd = T.build(enc, dec, d, seed=2)
    torch.manual_seed(3)
    x = bf16_round(torch.randn(shape))
    sd_req = {k: (v.clone().requires_grad_(True) if v.is_floating_point() else v) for k, v in sd.items()}
    # oracle with hooks on latent
    encv = O.tokenizer_encode(x, sd_req, enc); encv.retain_grad()
    (q, idx), ql = O.lfq_forward(encv, sd_req, 'quant.', d, 1, training=True, transpose=True)
    q.retain_grad()
    rec = O.tokenizer_decode(q, sd_req, dec)
    loss_ref = torch.nn.functional.mse_loss(rec, x) + ql
    loss_ref.backward()
    m.train()
    e = m.encode(x.cuda()); e.retain_grad()
    (qh, ih), qlh = m.quant(e, transpose=True); qh.retain_grad()
    r = m.decode(qh)
    from genie import functional as GF
    loss = GF.mse_loss(r, x.cuda()) + qlh
    loss.backward()
    print('loss', loss.item(), loss_ref.item(), 'qloss', qlh.item(), ql.item())
    print('dq   rel', T.rel_rms(qh.grad, q.grad))
    print('denc rel', T.rel_rms(e.grad, encv.grad), 'enc rel', T.rel_rms(e, encv))
    # gradient of latent given oracle's own latent through our LFQ
    for name, p in m.named_parameters():
        g = sd_req[name].grad
        if g is None or g.abs().max() == 0: continue
        print(f'{name:45s} {T.rel_rms(p.grad, g):.4f}')
