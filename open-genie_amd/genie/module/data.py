"""Data path in front of the hot path (drop-in for reference genie/module/data.py:26-234).

``Platformer2D`` yields fixed-length clips of recorded game play as float tensors in [0, 1]; ``LightningDataset`` turns datasets
into train / val / test ``DataLoader``s with the reference's knobs.  Two things are specific to this implementation:

* video decoding is behind a tiny reader interface (``open_video``): ``.mp4`` / ``.avi`` through OpenCV when it is importable
  (what the reference hard-wires, data.py:4-8), and raw ``uint8 (T, H, W, 3)`` frame arrays (``.npy`` / ``.npz``) always -- the
  container has no OpenCV, and at ~1.5 k frames/s/GPU a single-process mp4 decoder starves the training step anyway;
* ``DevicePrefetcher`` moves batches host -> HBM from PINNED memory on a side stream one batch ahead of the step, so the copy
  (12.6 MB per 16x64x64 fp32 clip) hides under the previous step's kernels.
"""
from __future__ import annotations

import os
import random
from typing import Callable, Iterator, Optional

import numpy as np
import torch
import yaml
from torch import Tensor
from torch.utils.data import DataLoader, Dataset, IterableDataset

from .._lightning import LightningDataModule
from ..utils import default, default_iterdata_worker_init, exists

VIDEO_EXT = ('.mp4', '.avi', '.mkv', '.mov')
ARRAY_EXT = ('.npy', '.npz')


class _ArrayReader:
    """uint8 (T, H, W, C) RGB frames in a .npy (memory-mapped) or .npz (key 'frames', or the only array)."""

    def __init__(self, path: str) -> None:
        if path.endswith('.npz'):
            z = np.load(path)
            self.frames = z['frames'] if 'frames' in z.files else z[z.files[0]]
        else:
            self.frames = np.load(path, mmap_mode='r')
        if self.frames.ndim != 4 or self.frames.dtype != np.uint8:
            raise ValueError(f'{path}: expected uint8 frames of shape (T, H, W, C), got {self.frames.dtype} {self.frames.shape}')

    def __len__(self) -> int:
        return self.frames.shape[0]

    def read(self, start: int, count: int) -> Tensor:
        return torch.from_numpy(np.array(self.frames[start:start + count]))            # a copy: the memory map is read-only

    def close(self) -> None:
        self.frames = None


class _Cv2Reader:
    """OpenCV reader: seek to `start`, decode `count` frames, BGR -> RGB (reference data.py:196-214)."""

    def __init__(self, path: str) -> None:
        try:
            import cv2
        except ImportError as e:
            raise ImportError(f'{path}: decoding video containers needs OpenCV (cv2), which is not installed; '
                              f'convert the clips to .npy frame arrays (uint8, T x H x W x 3) or install opencv-python') from e
        self.cv2 = cv2
        self.cap = cv2.VideoCapture(path)
        self.total = int(self.cap.get(cv2.CAP_PROP_FRAME_COUNT))

    def __len__(self) -> int:
        return self.total

    def read(self, start: int, count: int) -> Tensor:
        self.cap.set(self.cv2.CAP_PROP_POS_FRAMES, start)
        frames = []
        for _ in range(count):
            ok, frame = self.cap.read()
            if not ok:
                break
            frames.append(torch.from_numpy(self.cv2.cvtColor(frame, self.cv2.COLOR_BGR2RGB)))
        return torch.stack(frames) if frames else torch.empty(0, dtype=torch.uint8)

    def close(self) -> None:
        self.cap.release()


def open_video(path: str):
    ext = os.path.splitext(path)[1].lower()
    if ext in ARRAY_EXT:
        return _ArrayReader(path)
    if ext in VIDEO_EXT:
        return _Cv2Reader(path)
    raise ValueError(f'{path}: unknown clip format (expected one of {VIDEO_EXT + ARRAY_EXT})')


def _format_perm(output_format: str):
    """'c t h w' -> permutation of the decoded (t, h, w, c) axes (the reference's einops pattern 't h w c -> <output_format>')."""
    axes = output_format.lower().replace(' ', '')
    if sorted(axes) != sorted('thwc'):
        raise ValueError(f"output_format must be a permutation of 't c h w', got {output_format!r}")
    return tuple('thwc'.index(a) for a in axes)


class Platformer2D(Dataset):
    """reference data.py:139-234: one clip per recorded episode under ``root/env_name/split``; `num_frames` consecutive frames
    from frame 0 or from a random start (`randomize`); a clip that runs out of frames is cut ('none'), or padded by repeating its
    last frame ('repeat'), with zeros ('zero') or with one random frame ('random'); values / 255; axes per `output_format`."""

    def __init__(self, root: str, split: str = 'train', env_name: str = 'Coinrun', padding: str = 'none', randomize: bool = False,
                 transform: Callable | None = None, num_frames: int = 16, output_format: str = 't c h w', device_decode: bool = False) -> None:
        """`device_decode=True` (not in the reference): return the clip as the decoder produced it -- uint8 (t, h, w, c) -- and leave the division
        by 255, the axis order and the cast to ``DevicePrefetcher`` (``genie_u8_frames_to_cl`` on the GPU): a quarter of the bytes through the
        loader, the pinned staging and PCIe, no float pass on the host.  `output_format` must then be 'c t h w' (what the models take) and
        `transform` unset; 'random' padding draws its frame in uint8."""
        super().__init__()
        self.device_decode = bool(device_decode)
        if self.device_decode and (exists(transform) or output_format.lower().replace(' ', '') != 'cthw'):
            raise ValueError("Platformer2D(device_decode=True) needs output_format='c t h w' and no transform (the GPU produces that layout)")
        if padding not in ('none', 'repeat', 'zero', 'random'):
            raise ValueError(f'Invalid padding type: {padding}')
        self.root = os.path.join(root, env_name, split)
        self.split, self.padding, self.randomize, self.num_frames = split, padding, randomize, num_frames
        self.output_format = output_format
        self._perm = _format_perm(output_format)
        self.transform = transform if exists(transform) else (lambda x: x)
        self.file_names = sorted(os.path.join(self.root, f) for f in os.listdir(self.root)
                                 if os.path.splitext(f)[1].lower() in VIDEO_EXT + ARRAY_EXT)

    def __len__(self) -> int:
        return len(self.file_names)

    def __getitem__(self, idx: int) -> Tensor:
        return self.load_video_slice(self.file_names[idx], self.num_frames, None if self.randomize else 0)

    def load_video_slice(self, video_path: str, num_frames: int, start_frame: int | None = None) -> Tensor:
        rd = open_video(video_path)
        try:
            total = len(rd)
            want = min(num_frames, total)                       # a shorter video is returned whole (data.py:190-192)
            start = start_frame if exists(start_frame) else random.randint(0, total - want)
            frames = rd.read(start, want)
        finally:
            rd.close()
        got = frames.shape[0]
        if got == 0:
            raise RuntimeError(f'{video_path}: no frame could be decoded')
        if self.device_decode:                               # uint8 (t, h, w, c), padded the same way; DevicePrefetcher finishes the job on the GPU
            if got < want and self.padding != 'none':
                last = frames[-1:]
                fill = {'repeat': last, 'zero': torch.zeros_like(last), 'random': torch.randint(0, 256, last.shape, dtype=torch.uint8)}[self.padding]
                frames = torch.cat([frames, fill.expand(want - got, *last.shape[1:])])
            return frames.contiguous()
        video = frames.float() / 255.
        if got < want and self.padding != 'none':
            last = video[-1:]
            fill = {'repeat': last, 'zero': torch.zeros_like(last), 'random': torch.rand_like(last)}[self.padding]
            video = torch.cat([video, fill.expand(want - got, *last.shape[1:])])
        return self.transform(video.permute(*self._perm).contiguous())


class SyntheticVideos(Dataset):
    """Seeded random clips of a fixed shape (c, t, h, w): the benchmark's input, and the stand-in when no data root exists."""

    def __init__(self, num_clips: int = 1024, shape=(3, 16, 64, 64), seed: int = 0) -> None:
        self.num_clips, self.shape, self.seed = num_clips, tuple(shape), seed

    def __len__(self) -> int:
        return self.num_clips

    def __getitem__(self, idx: int) -> Tensor:
        g = torch.Generator().manual_seed(self.seed * 1_000_003 + idx)
        return torch.rand(self.shape, generator=g)


class LightningDataset(LightningDataModule):
    """reference data.py:26-137 (same constructor knobs) + ``pin_memory`` and the rank sharding Lightning would inject
    (``DistributedSampler`` when ``torch.distributed`` is initialised and no sampler / shuffle flag was given)."""

    @classmethod
    def from_config(cls, conf_path: str, *args, key: str = 'dataset') -> 'LightningDataset':
        with open(conf_path, 'r') as f:
            conf = yaml.safe_load(f)
        return cls(*args, **conf[key])

    def __init__(self, *args, batch_size: int = 16, num_workers: int = 0, train_shuffle: bool | None = None, val_shuffle: bool | None = None,
                 val_batch_size: None | int = None, worker_init_fn: None | Callable = None, collate_fn: None | Callable = None,
                 train_sampler: None | Callable = None, val_sampler: None | Callable = None, test_sampler: None | Callable = None,
                 pin_memory: bool = True) -> None:
        super().__init__()
        self.train_dataset = self.valid_dataset = self.test__dataset = None
        self.num_workers, self.batch_size, self.val_batch_size = num_workers, batch_size, default(val_batch_size, batch_size)
        self.train_shuffle, self.val_shuffle = train_shuffle, val_shuffle
        self.train_sampler, self.valid_sampler, self.test__sampler = train_sampler, val_sampler, test_sampler
        self.collate_fn, self.worker_init_fn, self.pin_memory = collate_fn, worker_init_fn, pin_memory

    def setup(self, stage: str) -> None:
        raise NotImplementedError('This is an abstract datamodule class. You should use one of the concrete subclasses that represents '
                                  'an actual dataset.')

    def _loader(self, dataset, sampler, batch_size, shuffle) -> DataLoader:
        if dataset is None:
            raise RuntimeError('dataset not set up: call setup("fit") / setup("test") first')
        init = default(self.worker_init_fn, default_iterdata_worker_init) if isinstance(dataset, IterableDataset) else self.worker_init_fn
        import torch.distributed as dist
        if sampler is None and not isinstance(dataset, IterableDataset) and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            from torch.utils.data.distributed import DistributedSampler
            sampler, shuffle = DistributedSampler(dataset, shuffle=bool(shuffle)), None           # rank r takes clips r::world
        return DataLoader(dataset, sampler=sampler, batch_size=batch_size, shuffle=shuffle, collate_fn=self.collate_fn,
                          num_workers=self.num_workers, worker_init_fn=init, pin_memory=self.pin_memory and torch.cuda.is_available())

    def train_dataloader(self) -> DataLoader:
        return self._loader(self.train_dataset, self.train_sampler, self.batch_size, self.train_shuffle)

    def val_dataloader(self) -> DataLoader:
        return self._loader(self.valid_dataset, self.valid_sampler, self.val_batch_size, self.val_shuffle)

    def test_dataloader(self) -> DataLoader:
        return self._loader(self.test__dataset, self.test__sampler, self.val_batch_size, self.val_shuffle)


def decode_frames_on_device(frames_u8: Tensor) -> Tensor:
    """uint8 (N, T, H, W, C) frames on the GPU -> the models' input: a bf16 channels-last (CL, genie/cl.py) tensor of logical shape (N, C, T, H, W)
    with value / 255 -- reference data.py:218-231 (`video / 255.`, 't h w c -> c t h w') fused with the model-boundary layout conversion, on the
    current stream (genie_u8_frames_to_cl)."""
    from .. import _hip
    from ..cl import empty_cl, pitch_of
    _hip.require_gpu(frames_u8, 'decode_frames_on_device')
    if frames_u8.dtype != torch.uint8 or frames_u8.dim() != 5:
        raise ValueError(f'decode_frames_on_device: expected uint8 (N, T, H, W, C), got {frames_u8.dtype} {tuple(frames_u8.shape)}')
    f = frames_u8.contiguous()
    n, t, h, w, c = f.shape
    out = empty_cl(n, c, t, h, w, f.device, zero_pad=False)          # the kernel writes the pad channels
    _hip.check(_hip.load_library().genie_u8_frames_to_cl(f.data_ptr(), n * t * h * w, c, out.data_ptr(), pitch_of(out), _hip.stream_ptr()),
               'genie_u8_frames_to_cl')
    f.record_stream(torch.cuda.current_stream(f.device))
    return out


class DevicePrefetcher:
    """Iterate a loader with the NEXT batch's host -> device copy already in flight on a side stream (pinned source memory, so the
    copy is a real DMA that overlaps kernels).  On a CPU-only box it is a plain pass-through."""

    def __init__(self, loader, device=None, decode_u8: Optional[bool] = None) -> None:
        """`decode_u8`: turn 5-D uint8 tensors -- raw (N, T, H, W, C) frames -- into the models' CL bf16 (N, C, T, H, W) / 255 on the device.  Opt-in:
        the default (None) follows the loader's dataset tag (``dataset.device_decode``, set by ``Platformer2D(device_decode=True)``); a uint8 mask or
        label tensor of some other dataset is never reinterpreted on dtype and rank alone."""
        self.loader = loader
        self.decode_u8 = bool(getattr(getattr(loader, 'dataset', None), 'device_decode', False)) if decode_u8 is None else bool(decode_u8)
        self.device = torch.device(device) if device is not None else (torch.device('cuda', torch.cuda.current_device()) if torch.cuda.is_available() else None)
        self.stream = torch.cuda.Stream(self.device) if self.device is not None and self.device.type == 'cuda' else None

    def __len__(self) -> int:
        return len(self.loader)

    def _one(self, b):
        if not isinstance(b, Tensor):
            return b
        d = b.to(self.device, non_blocking=True)
        if self.decode_u8 and d.dtype == torch.uint8 and d.dim() == 5:          # (N, T, H, W, C) raw frames of Platformer2D(device_decode=True): -> CL bf16 (N, C, T, H, W) / 255
            d = decode_frames_on_device(d)
        return d

    def _to_device(self, batch):
        if self.stream is None:
            return batch
        with torch.cuda.stream(self.stream):
            if isinstance(batch, Tensor):
                return self._one(batch)
            if isinstance(batch, (list, tuple)):
                return type(batch)(self._one(b) for b in batch)
        return batch

    def __iter__(self) -> Iterator:
        it = iter(self.loader)
        try:
            nxt = self._to_device(next(it))
        except StopIteration:
            return
        while True:
            cur = nxt
            if self.stream is not None:
                torch.cuda.current_stream(self.device).wait_stream(self.stream)          # the batch about to be used has landed
                for t in ([cur] if isinstance(cur, Tensor) else cur):
                    if isinstance(t, Tensor):
                        t.record_stream(torch.cuda.current_stream(self.device))
            try:
                nxt = self._to_device(next(it))                                         # overlaps the step that consumes `cur`
            except StopIteration:
                yield cur
                return
            yield cur
