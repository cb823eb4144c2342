"""Concrete data modules (drop-in for reference genie/dataset.py:9-162)."""
from __future__ import annotations

from pathlib import Path
from typing import Callable, Tuple

from .module.data import LightningDataset, Platformer2D, SyntheticVideos


class LightningPlatformer2D(LightningDataset):
    """Recorded Procgen game play (reference dataset.py:99-162): train / val / test splits of ``Platformer2D``."""

    def __init__(self, root: str | Path, env_name: str = 'Coinrun', padding: str = 'none', randomize: bool = False,
                 transform: Callable | None = None, num_frames: int = 16, output_format: str = 't c h w', **kwargs) -> None:
        super().__init__(**kwargs)
        self.root, self.env_name, self.padding, self.randomize = str(root), env_name, padding, randomize
        self.transform, self.num_frames, self.output_format = transform, num_frames, output_format
        self.save_hyperparameters()

    def _split(self, split: str) -> Platformer2D:
        return Platformer2D(root=self.root, split=split, padding=self.padding, env_name=self.env_name, transform=self.transform,
                            randomize=self.randomize, num_frames=self.num_frames, output_format=self.output_format)

    def setup(self, stage: str) -> None:
        match stage:
            case 'fit':
                self.train_dataset, self.valid_dataset = self._split('train'), self._split('val')
            case 'test':
                self.test__dataset = self._split('test')
            case _:
                raise ValueError(f'Invalid stage: {stage}')


class LightningSynthetic(LightningDataset):
    """Seeded random clips (the benchmark's input; also what the entry points fall back to when ``data.root`` does not exist)."""

    def __init__(self, num_clips: int = 1024, shape: Tuple[int, ...] = (3, 16, 64, 64), seed: int = 0, **kwargs) -> None:
        super().__init__(**kwargs)
        self.num_clips, self.shape, self.seed = num_clips, tuple(shape), seed

    def setup(self, stage: str) -> None:
        match stage:
            case 'fit':
                self.train_dataset = SyntheticVideos(self.num_clips, self.shape, self.seed)
                self.valid_dataset = SyntheticVideos(max(self.num_clips // 16, 1), self.shape, self.seed + 1)
            case 'test':
                self.test__dataset = SyntheticVideos(max(self.num_clips // 16, 1), self.shape, self.seed + 2)
            case _:
                raise ValueError(f'Invalid stage: {stage}')


class LightningKinetics(LightningDataset):
    """reference dataset.py:9-97 wraps ``torchvision.datasets.Kinetics``; torchvision is not part of this stack."""

    def __init__(self, *args, **kwargs) -> None:
        raise ImportError('LightningKinetics needs torchvision.datasets.Kinetics (reference genie/dataset.py:3), which is not installed')
