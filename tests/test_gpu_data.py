"""Data path in front of the hot path on the GPU (reference genie/module/data.py:139-234): the device-side frame decode
(genie_u8_frames_to_cl) against the reference's host arithmetic, the DevicePrefetcher end to end, and the FEED RATE the loader sustains into
HBM against what the tokenizer step consumes."""
import os

import numpy as np
import pytest
import torch

from util import report

pytestmark = pytest.mark.gpu


@pytest.fixture()
def episodes(tmp_path):
    rng = np.random.default_rng(3)
    d = tmp_path / 'Coinrun' / 'train'
    d.mkdir(parents=True)
    for i in range(96):
        np.save(d / f'ep{i:03d}.npy', rng.integers(0, 256, size=(40, 64, 64, 3), dtype=np.uint8))
    return str(tmp_path)


def test_device_decode_is_bit_identical_to_the_host_path():
    """uint8 frames -> CL bf16 on the GPU == `video / 255.` + 't h w c -> c t h w' on the host (data.py:218-231) followed by the model-boundary
    layout conversion: same fp32 quotient, one rounding to bf16 -- bit for bit, for 3 and for 4 channels, every byte value present."""
    from genie import cl
    from genie.module.data import decode_frames_on_device
    g = torch.Generator().manual_seed(0)
    for c in (3, 4, 1):
        raw = torch.randint(0, 256, (2, 5, 16, 24, c), generator=g, dtype=torch.uint8)
        raw.view(-1)[:256] = torch.arange(256, dtype=torch.uint8)
        dev = decode_frames_on_device(raw.cuda())
        assert cl.is_cl(dev) and tuple(dev.shape) == (2, c, 5, 16, 24)
        host = cl.to_cl((raw.float() / 255.).permute(0, 4, 1, 2, 3).contiguous().cuda())
        assert torch.equal(dev.float().cpu(), host.float().cpu())
        if cl.pitch_of(dev) != c:                          # pad channels of the CL buffer are zero
            buf = torch.as_strided(dev, (2, 5, 16, 24, cl.pitch_of(dev)), (5 * 16 * 24 * cl.pitch_of(dev), 16 * 24 * cl.pitch_of(dev), 24 * cl.pitch_of(dev), cl.pitch_of(dev), 1))
            assert buf[..., c:].abs().max().item() == 0


def test_prefetcher_feeds_decoded_batches_and_the_rate_covers_the_step(episodes):
    """DataLoader(Platformer2D(device_decode=True)) -> pinned uint8 -> DevicePrefetcher -> CL bf16 batches; batches equal the host path's; and
    the sustained feed (loader workers + H2D + decode, nothing else on the GPU) exceeds the ~144 clips/s (2300 frames/s) one MI355X's
    tokenizer step consumes.  The reference-style host path is measured next to it."""
    import time
    from torch.utils.data import DataLoader
    from genie import cl
    from genie.module.data import DevicePrefetcher, Platformer2D
    raw_ds = Platformer2D(episodes, num_frames=16, output_format='c t h w', device_decode=True)
    flt_ds = Platformer2D(episodes, num_frames=16, output_format='c t h w')
    b_raw = next(iter(DevicePrefetcher(DataLoader(raw_ds, batch_size=4, shuffle=False, pin_memory=True))))
    b_flt = next(iter(DevicePrefetcher(DataLoader(flt_ds, batch_size=4, shuffle=False, pin_memory=True))))
    assert cl.is_cl(b_raw) and tuple(b_raw.shape) == (4, 3, 16, 64, 64) and b_flt.dtype == torch.float32
    assert torch.equal(b_raw.float().cpu(), b_flt.to(torch.bfloat16).float().cpu())

    def rate(ds, workers, epochs=3):
        dl = DataLoader(ds, batch_size=32, shuffle=True, num_workers=workers, pin_memory=True, drop_last=True, persistent_workers=workers > 0,
                        prefetch_factor=4 if workers > 0 else None)
        acc = torch.zeros((), device='cuda')
        for b in DevicePrefetcher(dl):                    # first epoch: worker start-up, page cache
            acc += b.float().mean()
        torch.cuda.synchronize()
        n, t0 = 0, time.perf_counter()
        for _ in range(epochs):
            for b in DevicePrefetcher(dl):
                acc += b.float().mean()
                n += b.shape[0]
        torch.cuda.synchronize()
        return n / (time.perf_counter() - t0)
    cpus = len(os.sched_getaffinity(0))
    w = max(2, min(8, cpus // 2))
    r_raw, r_flt = rate(raw_ds, w), rate(flt_ds, w)
    report('data_feed_rate', workers=w, host_cpus=cpus, device_decode_clips_per_s=r_raw, host_float_clips_per_s=r_flt, step_consumes_clips_per_s=144.0)
    assert r_raw > 1.5 * 144.0, (r_raw, r_flt)
