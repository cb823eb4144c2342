"""Data path on CPU: Platformer2D clip slicing / padding / axis order on raw frame arrays, the data modules'
loaders, rank sharding, and the prefetcher's pass-through form."""
import os

import numpy as np
import pytest
import torch

from util import ROOT  # noqa: F401


@pytest.fixture()
def clips(tmp_path):
    rng = np.random.default_rng(0)
    for split, n in (('train', 5), ('val', 2), ('test', 2)):
        d = tmp_path / 'Coinrun' / split
        d.mkdir(parents=True)
        for i in range(n):
            frames = rng.integers(0, 256, size=(10 + 3 * i, 8, 12, 3), dtype=np.uint8)
            if i % 2:
                np.savez(d / f'ep{i}.npz', frames=frames)
            else:
                np.save(d / f'ep{i}.npy', frames)
    (tmp_path / 'Coinrun' / 'train' / 'notes.txt').write_text('not a clip')
    return str(tmp_path)


def test_platformer2d_slicing_padding_and_format(clips):
    from genie.module.data import Platformer2D
    ds = Platformer2D(clips, split='train', num_frames=6, output_format='c t h w')
    assert len(ds) == 5
    v = ds[0]
    raw = np.load(os.path.join(clips, 'Coinrun', 'train', 'ep0.npy'))
    assert tuple(v.shape) == (3, 6, 8, 12) and v.dtype == torch.float32
    assert torch.equal(v, torch.from_numpy(raw[:6].copy()).float().div(255.).permute(3, 0, 1, 2))
    assert tuple(Platformer2D(clips, num_frames=6)[1].shape) == (6, 3, 8, 12)                  # default 't c h w'
    # a clip shorter than the request comes back whole (reference data.py:190-192) ...
    assert tuple(Platformer2D(clips, num_frames=64, padding='repeat')[0].shape) == (10, 3, 8, 12)
    # ... random starts stay inside the clip
    ds = Platformer2D(clips, num_frames=9, randomize=True)
    for _ in range(10):
        assert tuple(ds[0].shape) == (9, 3, 8, 12)
    with pytest.raises(ValueError):
        Platformer2D(clips, padding='mirror')
    with pytest.raises(ValueError):
        Platformer2D(clips, output_format='t c h')


def test_padding_modes_fill_truncated_reads(clips, monkeypatch):
    """A reader that delivers fewer frames than its header promises (what a damaged mp4 does) triggers the padding modes."""
    from genie.module import data as D

    class Short(D._ArrayReader):
        def read(self, start, count):
            return super().read(start, count)[: max(1, count - 3)]

    monkeypatch.setattr(D, 'open_video', lambda p: Short(p))
    base = D.Platformer2D(clips, num_frames=8, padding='none')[0]
    assert base.shape[0] == 5
    rep = D.Platformer2D(clips, num_frames=8, padding='repeat')[0]
    assert rep.shape[0] == 8 and torch.equal(rep[:5], base) and all(torch.equal(rep[i], base[-1]) for i in range(5, 8))
    zero = D.Platformer2D(clips, num_frames=8, padding='zero')[0]
    assert zero.shape[0] == 8 and zero[5:].abs().sum() == 0
    rnd = D.Platformer2D(clips, num_frames=8, padding='random')[0]
    assert rnd.shape[0] == 8 and torch.equal(rnd[5], rnd[7]) and 0 <= rnd[5:].min() and rnd[5:].max() <= 1


def test_video_container_needs_opencv(tmp_path):
    from genie.module.data import open_video
    p = tmp_path / 'clip.mp4'
    p.write_bytes(b'')
    try:
        import cv2  # noqa: F401
    except ImportError:
        with pytest.raises(ImportError, match='OpenCV'):
            open_video(str(p))
    with pytest.raises(ValueError):
        open_video(str(tmp_path / 'clip.gif'))


def test_data_modules_and_prefetcher(clips):
    from genie.dataset import LightningKinetics, LightningPlatformer2D, LightningSynthetic
    from genie.module.data import DevicePrefetcher, LightningDataset
    dm = LightningPlatformer2D(clips, num_frames=8, output_format='c t h w', batch_size=2, train_shuffle=False)
    with pytest.raises(RuntimeError):
        dm.train_dataloader()
    dm.setup('fit')
    batches = list(DevicePrefetcher(dm.train_dataloader(), device=None if not torch.cuda.is_available() else 'cuda'))
    assert len(batches) == 3 and tuple(batches[0].shape) == (2, 3, 8, 8, 12) and tuple(batches[-1].shape) == (1, 3, 8, 8, 12)
    assert len(list(dm.val_dataloader())) == 1
    dm.setup('test')
    assert len(dm.test_dataloader().dataset) == 2
    with pytest.raises(ValueError):
        dm.setup('predict')
    syn = LightningSynthetic(num_clips=10, shape=(3, 4, 8, 8), batch_size=4)
    syn.setup('fit')
    a, b = next(iter(syn.train_dataloader())), next(iter(syn.train_dataloader()))
    assert tuple(a.shape) == (4, 3, 4, 8, 8) and torch.equal(a, b) and 0 <= a.min() and a.max() <= 1
    with pytest.raises(NotImplementedError):
        LightningDataset().setup('fit')
    with pytest.raises(ImportError):
        LightningKinetics('x', 16)
    import yaml
    cfg = os.path.join(clips, 'data.yaml')
    yaml.safe_dump({'dataset': {'root': clips, 'num_frames': 4, 'batch_size': 3}}, open(cfg, 'w'))
    dm2 = LightningPlatformer2D.from_config(cfg)
    assert dm2.batch_size == 3 and dm2.num_frames == 4


# ----------------------------------------------------------------------------------------------------------------------------------
# against the REAL reference reader (reference genie/module/data.py:139-234), wherever /root/reference is present
# ----------------------------------------------------------------------------------------------------------------------------------
class _FakeCv2:
    """The five OpenCV names the reference binds (data.py:4-8), served from frame arrays: a '.mp4' here is an .npy dump of BGR uint8
    frames (T, H, W, 3), optionally followed by a header lie (`<name>.mp4.count` holds the frame count the container CLAIMS -- a
    damaged clip runs out of frames before that)."""
    CAP_PROP_FRAME_COUNT, CAP_PROP_POS_FRAMES, COLOR_BGR2RGB = 7, 1, 4

    class VideoCapture:
        def __init__(self, path):
            with open(path, 'rb') as f:
                self.frames = np.load(f)
            self.claimed = int(open(path + '.count').read()) if os.path.exists(path + '.count') else len(self.frames)
            self.pos = 0

        def get(self, prop):
            assert prop == _FakeCv2.CAP_PROP_FRAME_COUNT
            return float(self.claimed)

        def set(self, prop, value):
            assert prop == _FakeCv2.CAP_PROP_POS_FRAMES
            self.pos = int(value)

        def read(self):
            if self.pos >= len(self.frames):
                return False, None
            self.pos += 1
            return True, self.frames[self.pos - 1]

        def release(self):
            pass

    @staticmethod
    def cvtColor(frame, code):
        assert code == _FakeCv2.COLOR_BGR2RGB
        return np.ascontiguousarray(frame[..., ::-1])


@pytest.fixture()
def mp4_clips(tmp_path):
    rng = np.random.default_rng(1)
    d = tmp_path / 'Coinrun' / 'train'
    d.mkdir(parents=True)
    for i, (n, claimed) in enumerate(((12, None), (7, None), (9, 14), (20, None))):
        with open(d / f'ep{i}.mp4', 'wb') as f:
            np.save(f, rng.integers(0, 256, size=(n, 6, 10, 3), dtype=np.uint8))
        if claimed:
            (d / f'ep{i}.mp4.count').write_text(str(claimed))
    return str(tmp_path)


def test_platformer2d_matches_the_reference_reader(mp4_clips, monkeypatch):
    """Same clips through the reference's Platformer2D (its cv2 names bound to the array-backed fake above) and through ours (whose
    OpenCV reader imports the same fake): identical tensors for every padding mode the reference can run, both axis orders, short
    clips, a damaged clip whose header promises more frames than it has, and seeded random starts (the readers had only
    been compared with hand-built expectations)."""
    import random
    import sys

    from oracle.ref_import import ref_module, reference_available
    if not reference_available():
        pytest.skip('/root/reference is not present on this box')
    from genie.module import data as D
    R = ref_module('module.data')
    for name in ('VideoCapture', 'cvtColor', 'COLOR_BGR2RGB', 'CAP_PROP_POS_FRAMES', 'CAP_PROP_FRAME_COUNT'):
        monkeypatch.setattr(R, name, getattr(_FakeCv2, name))
    monkeypatch.setitem(sys.modules, 'cv2', _FakeCv2)
    monkeypatch.setattr(D, 'VIDEO_EXT', ('.mp4',))                      # the '.count' side files are not clips
    monkeypatch.setattr(R, 'listdir', lambda p: sorted(f for f in os.listdir(p) if f.endswith('.mp4')))
    checked = 0
    for padding in ('none', 'repeat', 'zero'):
        for fmt in ('t c h w', 'c t h w'):
            for nf in (5, 8, 16):
                ref = R.Platformer2D(mp4_clips, padding=padding, num_frames=nf, output_format=fmt)
                ours = D.Platformer2D(mp4_clips, padding=padding, num_frames=nf, output_format=fmt)
                assert [os.path.basename(f) for f in ref.file_names] == [os.path.basename(f) for f in ours.file_names]
                for i in range(len(ref)):
                    a, b = ref[i], ours[i]
                    assert a.shape == b.shape and a.dtype == b.dtype == torch.float32, (padding, fmt, nf, i, a.shape, b.shape)
                    assert torch.equal(a, b), (padding, fmt, nf, i)
                    checked += 1
    assert checked == 72
    # random starts: both draw from Python's global generator with the same call
    ref = R.Platformer2D(mp4_clips, randomize=True, num_frames=6)
    ours = D.Platformer2D(mp4_clips, randomize=True, num_frames=6)
    for i in range(len(ref)):
        random.seed(100 + i); a = ref[i]
        random.seed(100 + i); b = ours[i]
        assert torch.equal(a, b)
    # 'random' padding: the reference calls torch.rand_like on a uint8 frame, which torch rejects; ours pads with a float frame in [0, 1)
    with pytest.raises(Exception):
        R.Platformer2D(mp4_clips, padding='random', num_frames=16)[2]
    v = D.Platformer2D(mp4_clips, padding='random', num_frames=16)[2]
    assert tuple(v.shape) == (14, 3, 6, 10) and 0 <= v.min() and v.max() <= 1


def test_platformer2d_device_decode_returns_raw_frames(clips, monkeypatch):
    """device_decode=True: the clip leaves the dataset as the decoder produced it -- uint8 (t, h, w, c), same frames, same padding rule --
    and the float / layout work is left to the GPU (DevicePrefetcher, genie_u8_frames_to_cl; its numbers are checked on the GPU in
    tests/test_gpu_data.py)."""
    from genie.module import data as D
    ds = D.Platformer2D(clips, split='train', num_frames=6, output_format='c t h w', device_decode=True)
    ref = D.Platformer2D(clips, split='train', num_frames=6, output_format='c t h w')
    for i in range(len(ds)):
        raw, flt = ds[i], ref[i]
        assert raw.dtype == torch.uint8 and tuple(raw.shape) == (6, 8, 12, 3) and raw.is_contiguous()
        assert torch.equal(raw.float().div(255.).permute(3, 0, 1, 2), flt)
    with pytest.raises(ValueError):
        D.Platformer2D(clips, device_decode=True)                              # default 't c h w' is not what the models take
    with pytest.raises(ValueError):
        D.Platformer2D(clips, output_format='c t h w', device_decode=True, transform=lambda x: x)
    # truncated read + 'repeat' padding in uint8
    real = D.open_video

    class Short:
        def __init__(self, path):
            self.rd = real(path)

        def __len__(self):
            return len(self.rd)

        def read(self, start, count):
            return self.rd.read(start, count)[:4]

        def close(self):
            self.rd.close()
    monkeypatch.setattr(D, 'open_video', Short)
    v = D.Platformer2D(clips, num_frames=6, padding='repeat', output_format='c t h w', device_decode=True)[0]
    assert tuple(v.shape) == (6, 8, 12, 3) and torch.equal(v[4], v[3]) and torch.equal(v[5], v[3])
    # the prefetcher's CPU form passes raw batches through untouched
    batch = torch.stack([ds[0], ds[1]])
    if not torch.cuda.is_available():
        assert torch.equal(next(iter(D.DevicePrefetcher([batch]))), batch)
