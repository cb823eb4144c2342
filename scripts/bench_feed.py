#!/usr/bin/env python
"""Feed rate of the data path in front of the hot path (reference genie/module/data.py:139-234): clips per second that
Platformer2D -> DataLoader -> pinned memory -> DevicePrefetcher sustains into HBM, against what the tokenizer's training step consumes
(BENCH: ~144 clips/s = 2300 frames/s on one MI355X at 64 clips per step).

    python scripts/bench_feed.py [--clips 512] [--batch 64] [--workers 0,4,8] [--with-step 1]

Writes synthetic episodes (uint8 .npy frame arrays, 64x64, 64 frames each) under a temp directory, then measures, per worker count and for
both forms of the dataset -- the reference's (float clips made on the host) and device_decode=True (uint8 to the GPU, /255 + layout there):
  feed_only   clips/s with nothing consuming the batches but a device-side sum (the loader + H2D ceiling)
  with_step   clips/s of Trainer-style steps of the MAGVIT2 tokenizer fed by that loader (optional; 64 clips per step)
One JSON line per configuration."""
import argparse
import json
import os
import shutil
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'open-genie_amd')):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np   # noqa: E402
import torch         # noqa: E402


def make_episodes(root, n, frames=64, hw=64, seed=0):
    d = os.path.join(root, 'Coinrun', 'train')
    os.makedirs(d, exist_ok=True)
    rng = np.random.default_rng(seed)
    for i in range(n):
        np.save(os.path.join(d, f'ep_{i:05d}.npy'), rng.integers(0, 256, (frames, hw, hw, 3), dtype=np.uint8))
    return root


def feed_rate(root, batch, workers, device_decode, max_batches=None):
    from torch.utils.data import DataLoader
    from genie.module.data import DevicePrefetcher, Platformer2D
    ds = Platformer2D(root, split='train', randomize=True, num_frames=16, output_format='c t h w', device_decode=device_decode)
    dl = DataLoader(ds, batch_size=batch, shuffle=True, num_workers=workers, pin_memory=True, drop_last=True, persistent_workers=workers > 0,
                    prefetch_factor=4 if workers > 0 else None)
    acc = torch.zeros((), device='cuda')
    first = None
    for b in DevicePrefetcher(dl):                         # epoch 0: worker start-up, page cache -- not feed rate
        first = b if first is None else first
        acc += b.float().mean()
    torch.cuda.synchronize()
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < 4.0:                  # whole epochs for at least 4 s: the workers' prefetch queues start every epoch empty
        for b in DevicePrefetcher(dl):
            acc += b.float().mean()                        # touch the batch on the device
            n += b.shape[0]
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return n / dt, tuple(first.shape), str(first.dtype)


def step_rate(root, batch, workers, device_decode, steps=6):
    from torch.utils.data import DataLoader
    from genie import MAGVIT2_DEC_DESC, MAGVIT2_ENC_DESC, VideoTokenizer
    from genie.module.data import DevicePrefetcher, Platformer2D
    from genie.trainer import ParamArena
    ds = Platformer2D(root, split='train', randomize=True, num_frames=16, output_format='c t h w', device_decode=device_decode)
    dl = DataLoader(ds, batch_size=batch, shuffle=True, num_workers=workers, pin_memory=True, drop_last=True, persistent_workers=workers > 0,
                    prefetch_factor=4 if workers > 0 else None)
    torch.manual_seed(0)
    model = VideoTokenizer(MAGVIT2_ENC_DESC, MAGVIT2_DEC_DESC, d_codebook=18, gan_loss_weight=0., perc_loss_weight=0.).cuda().train()
    arena = ParamArena(model)
    arena.attach_weight_packs(model)
    n, t0 = 0, None
    for i, b in enumerate(DevicePrefetcher(dl)):
        if i == 2:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        loss, _ = model(b)
        loss.backward()
        arena.adamw_step(lr=1e-3, weight_decay=0.01)
        if i >= 2:
            n += b.shape[0]
        if i >= 2 + steps - 1:
            break
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    del model, arena
    torch.cuda.empty_cache()
    return n / dt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--clips', type=int, default=512)
    ap.add_argument('--batch', type=int, default=64)
    ap.add_argument('--workers', default='0,4,8')
    ap.add_argument('--with-step', type=int, default=0)
    args = ap.parse_args()
    tmp = tempfile.mkdtemp(prefix='genie_feed_')
    try:
        make_episodes(tmp, args.clips)
        for dd in (False, True):
            for w in [int(x) for x in args.workers.split(',')]:
                rate, shape, dtype = feed_rate(tmp, args.batch, w, dd)
                rec = {'dataset': 'Platformer2D (uint8 .npy episodes, 16-frame clips of 64x64)', 'device_decode': dd, 'workers': w, 'batch': args.batch,
                       'feed_only_clips_per_s': round(rate, 1), 'feed_only_frames_per_s': round(rate * 16, 1), 'batch_on_device': f'{dtype} {shape}',
                       'host_cpus': len(os.sched_getaffinity(0))}
                if args.with_step:
                    sr = step_rate(tmp, args.batch, w, dd)
                    rec.update({'with_step_clips_per_s': round(sr, 1), 'with_step_frames_per_s': round(sr * 16, 1)})
                print(json.dumps(rec), flush=True)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == '__main__':
    main()
