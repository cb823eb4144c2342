#!/usr/bin/env python
"""Training-step timing of the other BASELINE configs on one MI355X (configs[2] LatentAction, configs[3] DynamicsModel, and the
repaired yaml / REPR tokenizer): ms per step and clips (or latent frames) per second, synthetic inputs resident in HBM."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'open-genie_amd')]
import torch

from genie import (LATENT_ACT_DEC, LATENT_ACT_ENC, MAGVIT2_DEC_DESC, MAGVIT2_ENC_DESC, REPR_TOK_DEC, REPR_TOK_ENC, DynamicsModel, Genie, LatentAction,
                   VideoTokenizer)
from genie.trainer import ParamArena


def run(name, model, step_fn, units, steps=4, warm=2):
    arena = ParamArena(model)
    arena.attach_weight_packs(model)
    for _ in range(warm):
        step_fn().backward(); arena.adamw_step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step_fn().backward(); arena.adamw_step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    out = {'model': name, 'ms_per_step': round(ms, 2), 'units_per_s': round(units / ms * 1e3, 1), 'params': sum(p.numel() for p in model.parameters()),
           'peak_mem_GB': round(torch.cuda.max_memory_allocated() / 2 ** 30, 2)}
    print(json.dumps(out), flush=True)
    del arena
    torch.cuda.empty_cache(); torch.cuda.reset_peak_memory_stats()
    return out


def main():
    torch.manual_seed(0)
    res = []
    which = sys.argv[1:] or ['lam', 'dyn', 'repr']
    if 'lam' in which:
        B = 2
        lam = LatentAction(LATENT_ACT_ENC, LATENT_ACT_DEC, d_codebook=8, inp_channels=3, inp_shape=(64, 64), n_embd=256).cuda().train()
        v = torch.randn(B, 3, 16, 64, 64, device='cuda')
        res.append(run(f'LatentAction (R-lam, n_embd 256, 16x64x64, B={B}) [frames/s]', lam, lambda: lam(v)[1], B * 16))
        del lam
    if 'dyn' in which:
        B = 4
        desc = (('space-time_attn', {'n_rep': 8, 'n_head': 8, 'd_head': 64}),)
        dyn = DynamicsModel(desc, tok_vocab=2 ** 18, act_vocab=8, embed_dim=512).cuda().train()
        tok = torch.randint(0, 2 ** 18, (B, 16, 8, 8), device='cuda'); act = torch.randint(0, 8, (B, 16), device='cuda')
        g = torch.Generator().manual_seed(1)
        mask = (torch.rand(B, 16, 8, 8, generator=g) < 0.75)
        res.append(run(f'DynamicsModel (8 x ST(8x64), V=2^18, (16,8,8) tokens, B={B}) [latent frames/s]', dyn,
                       lambda: dyn.compute_loss(tok, act, mask=mask), B * 16))
        del dyn
    if 'repr' in which:
        B = 2
        tokz = VideoTokenizer(REPR_TOK_ENC, REPR_TOK_DEC, d_codebook=10, gan_loss_weight=0., perc_loss_weight=0.).cuda().train()
        v = torch.randn(B, 3, 16, 64, 64, device='cuda')
        res.append(run(f'VideoTokenizer REPR_TOK (8+8 ST blocks, C=512, 16x16x16 latent, B={B}) [frames/s]', tokz, lambda: tokz(v)[0], B * 16))
    if 'genie4' in which:
        # BASELINE configs[4]: full Genie on 32x128x128 clips -- frozen MAGVIT2 tokenizer (8x16x16 tokens of 2^18 codes), latent-action
        # model over 32 frames of 128x128 pixels (spatial attention over S = 16384 positions), MaskGIT dynamics; two clips per GPU
        # (a batch of one dies in the reference's mask.squeeze() indexing, dynamics.py:89-97, and so does it here)
        B = 2
        tokz = VideoTokenizer(MAGVIT2_ENC_DESC, MAGVIT2_DEC_DESC, d_codebook=18, gan_loss_weight=0., perc_loss_weight=0.).cuda().eval()
        gen = Genie(tokz, inp_shape=(128, 128)).cuda().train()
        gen.tokenizer.eval()
        v = torch.randn(B, 3, 32, 128, 128, device='cuda')
        res.append(run(f'Genie (configs[4]: frozen MAGVIT2 tokenizer + R-lam + dynamics, 32x128x128, B={B}) [frames/s]', gen,
                       lambda: gen.compute_loss(v)[0], B * 32, steps=3, warm=1))
        del gen, tokz
    os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
    json.dump(res, open(os.path.join(ROOT, 'gpurun_out', 'bench_models.json'), 'w'), indent=1)


if __name__ == '__main__':
    main()
