"""Small helpers with the reference's names (reference genie/utils.py:15-75)."""
from __future__ import annotations

from typing import Tuple, TypeVar

import torch
from torch import Tensor

T = TypeVar('T')
D = TypeVar('D')

Blueprint = Tuple[str | Tuple[str, dict], ...]


def exists(var) -> bool:
    return var is not None


def default(var, val):
    return var if var is not None else val


def enlarge_as(src: Tensor, other: Tensor) -> Tensor:
    """Append singleton dims to `src` until it has as many dims as `other` (reference utils.py:21-28)."""
    return src.reshape(src.shape + (1,) * (other.dim() - src.dim())).contiguous()


def pick_frames(video: Tensor, frames_idxs: Tensor | None = None, frames_per_batch: int | None = None) -> Tensor:
    """Pick `frames_per_batch` random frames per clip (reference utils.py:30-56)."""
    assert exists(frames_idxs) ^ exists(frames_per_batch), 'Either `frames_idxs` or `frames_per_batch` must be provided.'
    b, c, t, h, w = video.shape
    if frames_idxs is None:
        frames_idxs = torch.cat([torch.randperm(t, device=video.device)[:frames_per_batch] for _ in range(b)])
    per = frames_per_batch if frames_per_batch is not None else frames_idxs.numel() // b
    batch_idxs = torch.repeat_interleave(torch.arange(b, device=video.device), per)
    return video[batch_idxs, :, frames_idxs, ...]


def enc2dec_name(name: str) -> str:
    return name.replace('downsample', 'upsample')


def default_iterdata_worker_init(worker_id: int) -> None:
    """Give every DataLoader worker of an IterableDataset its own seed and its own [_start, _end) slice of the stream
    (reference utils.py:61-75)."""
    from torch.utils.data import get_worker_info
    torch.manual_seed(torch.initial_seed() + worker_id)
    info = get_worker_info()
    if info is None:
        return
    ds = info.dataset
    lo, hi = ds._start, ds._end
    per_worker = int((hi - lo) / info.num_workers)
    ds._start = lo + info.id * per_worker
    ds._end = min(ds._start + per_worker, hi)
