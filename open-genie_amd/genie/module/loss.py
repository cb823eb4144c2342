"""Critic losses of the tokenizer (drop-in for reference genie/module/loss.py).

``GANLoss`` (loss.py:109-164): hinge loss on a FrameDiscriminator over a few random frames per clip -- the critic's convolutions,
GroupNorm + LeakyReLU and residual sums run on the HIP kernels (genie/module/discriminator.py), the hinge itself is a handful of
scalar torch ops.  ``PerceptualLoss`` (loss.py:34-107) needs torchvision's VGG16 with downloaded ImageNet weights, which do not
exist offline: constructing it raises (SURVEY.md sections 0 and 2, row 12).
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.nn as nn
from torch import Tensor
from torch.nn.functional import relu

from ..utils import pick_frames
from .discriminator import FrameDiscriminator, VideoDiscriminator


class GANLoss(nn.Module):
    def __init__(self, discriminate: str = 'frames', num_frames: int = 4, **kwargs) -> None:
        super().__init__()
        assert discriminate in ('frames', 'video'), 'Invalid discriminator type. Must be either "frames" or "video".'
        self.disc = FrameDiscriminator(**kwargs) if discriminate == 'frames' else VideoDiscriminator(**kwargs)
        self.num_frames = num_frames
        self.discriminate = discriminate

    def get_examples(self, rec_video: Tensor, inp_video: Tensor, frame_idxs: Optional[Tensor] = None) -> Tuple[Tensor, Tensor]:
        """loss.py:127-145.  `frame_idxs` (b * num_frames,) injects the random frame choice (parity tests); default = the
        reference's per-clip ``torch.randperm(t)[:num_frames]``."""
        b, c, t, h, w = inp_video.shape
        if self.discriminate == 'video':
            return rec_video, inp_video
        if frame_idxs is None:
            frame_idxs = torch.cat([torch.randperm(t, device=inp_video.device)[:self.num_frames] for _ in range(b)])
        frame_idxs = frame_idxs.to(inp_video.device)
        return pick_frames(rec_video, frame_idxs), pick_frames(inp_video, frame_idxs)

    def forward(self, rec_video: Tensor, inp_video: Tensor, train_gen: bool, frame_idxs: Optional[Tensor] = None) -> Tensor:
        fake, real = self.get_examples(rec_video, inp_video, frame_idxs)
        fake_score = self.disc(fake) if train_gen else self.disc(fake.detach())
        real_score = self.disc(real) if not train_gen else None
        # hinge: the generator raises the critic's opinion of the fakes, the critic separates fakes from reals by a margin
        return -fake_score.mean() if train_gen else (relu(1 + fake_score) + relu(1 - real_score)).mean()


class PerceptualLoss(nn.Module):
    def __init__(self, *args, **kwargs) -> None:
        super().__init__()
        raise NotImplementedError('PerceptualLoss needs torchvision VGG16 with downloaded weights (reference loss.py:46), which are not '
                                  'available offline; construct the VideoTokenizer with perc_loss_weight=0')
