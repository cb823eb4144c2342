"""VideoTokenizer (drop-in for reference genie/tokenizer.py:225-442) on the HIP hot path.

encode -> LFQ -> decode run entirely on ``libgenie_hip.so`` kernels in bf16 CL layout.  The loss of
``forward`` is the reference's expression (tokenizer.py:375-379) with the GAN and perceptual terms available
only at weight 0: those two critics (FrameDiscriminator, frozen VGG16) are outside the hot path (SURVEY.md
section 8f-2) -- the reference itself cannot run with them disabled (it calls ``nn.Identity`` with extra
arguments); here weight 0 means "term omitted" (R-fwd).
"""
from __future__ import annotations

from typing import Any, Callable, Dict, Iterable, Tuple

import torch
import torch.nn as nn
from torch import Tensor
from torch.optim import AdamW, Optimizer

from . import functional as GF
from ._lightning import LightningModule
from .module import parse_blueprint
from .module.loss import GANLoss
from .module.norm import GroupNorm, SiLU
from .module.quantization import LookupFreeQuantization
from .utils import Blueprint, default, exists

OptimizerCallable = Callable[[Iterable], Optimizer]

MAGVIT2_ENC_DESC = (
    ('causal-conv3d', {'in_channels': 3, 'out_channels': 128, 'kernel_size': 3}),
    ('video-residual', {'n_rep': 4, 'in_channels': 128}),
    ('spacetime_downsample', {'in_channels': 128, 'out_channels': 128, 'kernel_size': 3, 'time_factor': 1, 'space_factor': 2}),
    ('video-residual', {'in_channels': 128, 'out_channels': 256}),
    ('video-residual', {'n_rep': 3, 'in_channels': 256}),
    ('spacetime_downsample', {'in_channels': 256, 'out_channels': 256, 'kernel_size': 3, 'time_factor': 2, 'space_factor': 2}),
    ('video-residual', {'n_rep': 4, 'in_channels': 256}),
    ('spacetime_downsample', {'in_channels': 256, 'out_channels': 256, 'kernel_size': 3, 'time_factor': 2, 'space_factor': 2}),
    ('video-residual', {'in_channels': 256, 'out_channels': 512}),
    ('video-residual', {'n_rep': 7, 'in_channels': 512}),
    ('group_norm', {'num_groups': 8, 'num_channels': 512}),
    ('silu', {}),
    ('causal-conv3d', {'in_channels': 512, 'out_channels': 18, 'kernel_size': 1}),
)

MAGVIT2_DEC_DESC = (
    ('causal-conv3d', {'in_channels': 18, 'out_channels': 512, 'kernel_size': 3}),
    ('video-residual', {'n_rep': 4, 'in_channels': 512}),
    ('adaptive_group_norm', {'dim_cond': 18, 'num_groups': 8, 'num_channels': 512, 'has_ext': True}),
    ('video-residual', {'n_rep': 4, 'in_channels': 512}),
    ('depth2spacetime_upsample', {'in_channels': 512, 'kernel_size': 3, 'time_factor': 2, 'space_factor': 2}),
    ('adaptive_group_norm', {'dim_cond': 18, 'num_groups': 8, 'num_channels': 512, 'has_ext': True}),
    ('video-residual', {'in_channels': 512, 'out_channels': 256}),
    ('video-residual', {'n_rep': 3, 'in_channels': 256}),
    ('depth2spacetime_upsample', {'in_channels': 256, 'kernel_size': 3, 'time_factor': 2, 'space_factor': 2}),
    ('adaptive_group_norm', {'dim_cond': 18, 'num_groups': 8, 'num_channels': 256, 'has_ext': True}),
    ('video-residual', {'n_rep': 4, 'in_channels': 256}),
    ('depth2spacetime_upsample', {'in_channels': 256, 'kernel_size': 3, 'time_factor': 1, 'space_factor': 2}),
    ('adaptive_group_norm', {'dim_cond': 18, 'num_groups': 8, 'num_channels': 256, 'has_ext': True}),
    ('video-residual', {'in_channels': 256, 'out_channels': 128}),
    ('video-residual', {'n_rep': 3, 'in_channels': 128}),
    ('group_norm', {'num_groups': 8, 'num_channels': 128}),
    ('silu', {}),
    ('causal-conv3d', {'in_channels': 128, 'out_channels': 3, 'kernel_size': 3}),
)

REPR_TOK_ENC = (
    ('spacetime_downsample', {'in_channels': 3, 'kernel_size': 3, 'out_channels': 512, 'time_factor': 1, 'space_factor': 4}),
    ('space-time_attn', {'n_rep': 8, 'n_head': 8, 'd_head': 64, 'transpose': True}),
)

REPR_TOK_DEC = (
    ('space-time_attn', {'n_rep': 8, 'n_head': 8, 'd_head': 64, 'transpose': True}),
    ('depth2spacetime_upsample', {'in_channels': 512, 'kernel_size': 3, 'out_channels': 3, 'time_factor': 1, 'space_factor': 4}),
)


def get_enc(name: str) -> Blueprint:
    match name:
        case 'magvit2':
            return MAGVIT2_ENC_DESC
        case 'repr_tok':
            return REPR_TOK_ENC
        case _:
            raise ValueError(f'Unknown encoder: {name}')


def get_dec(name: str) -> Blueprint:
    match name:
        case 'magvit2':
            return MAGVIT2_DEC_DESC
        case 'repr_tok':
            return REPR_TOK_DEC
        case _:
            raise ValueError(f'Unknown decoder: {name}')


def _copy_desc(desc):
    """parse_blueprint pops keys from the caller's dicts (as the reference does); the module-level descs are
    shared, so they are copied before parsing."""
    return tuple(d if isinstance(d, str) else (d[0], dict(d[1])) for d in desc)


class _OutOfScopeCritic(nn.Module):
    def __init__(self, what: str) -> None:
        super().__init__()
        self.what = what

    def forward(self, *args, **kwargs):
        raise NotImplementedError(f'{self.what} is outside the implemented hot path (SURVEY.md section 8f-2); construct the '
                                  f'VideoTokenizer with the corresponding loss weight set to 0')


def run_layers(layers: nn.ModuleList, ext: list, x: Tensor, cond: Tensor | None, record=None) -> Tensor:
    """The reference's layer loop (tokenizer.py:314-315, 326-328) with one peephole: GroupNorm immediately
    followed by SiLU runs as one fused pass.  `record(lo, hi, x, y)`, when given, sees every step's input and output (layers [lo, hi)
    of the list): the per-layer parity tests read the stage boundaries through it."""
    i, n = 0, len(layers)
    while i < n:
        layer, has_ext = layers[i], ext[i]
        if isinstance(layer, GroupNorm) and not has_ext and i + 1 < n and isinstance(layers[i + 1], SiLU) and not ext[i + 1]:
            y, span = GF.group_norm(x, layer.num_groups, layer.weight, layer.bias, layer.eps, act=True), 2
        else:
            y, span = (layer(x, cond) if has_ext else layer(x)), 1
        if record is not None:
            record(i, i + span, x, y)
        x = y
        i += span
    return x


class VideoTokenizer(LightningModule):
    def __init__(self, enc_desc: Blueprint, dec_desc: Blueprint, disc_kwargs: Dict[str, Any] = {}, d_codebook: int = 18,
                 n_codebook: int = 1, lfq_bias: bool = True, lfq_frac_sample: float = 1., lfq_commit_weight: float = 0.25,
                 lfq_entropy_weight: float = 0.1, lfq_diversity_weight: float = 1., optimizer: OptimizerCallable = AdamW,
                 perceptual_model: str = 'vgg16',
                 perc_feat_layers: str | Iterable[str] = ('features.6', 'features.13', 'features.18', 'features.25'),
                 gan_discriminate: str = 'frames', gan_frames_per_batch: int = 4, gan_loss_weight: float = 1.,
                 perc_loss_weight: float = 1., quant_loss_weight: float = 1.) -> None:
        super().__init__()
        self.optimizer = optimizer
        self.enc_layers, self.enc_ext = parse_blueprint(_copy_desc(enc_desc))
        self.dec_layers, self.dec_ext = parse_blueprint(_copy_desc(dec_desc))
        last_enc_dim = [m.out_channels for m in self.enc_layers.modules() if hasattr(m, 'out_channels')][-1]
        first_dec_dim = self.dec_layers[0].in_channels
        assert last_enc_dim == first_dec_dim, 'Inconsistent encoder/decoder dimensions'
        self.quant = LookupFreeQuantization(codebook_dim=d_codebook, num_codebook=n_codebook, input_dim=last_enc_dim,
                                            use_bias=lfq_bias, frac_sample=lfq_frac_sample, commit_weight=lfq_commit_weight,
                                            entropy_weight=lfq_entropy_weight, diversity_weight=lfq_diversity_weight)
        self.perc_crit = _OutOfScopeCritic('PerceptualLoss (frozen VGG16 with downloaded weights)') if perc_loss_weight > 0 else nn.Identity()
        # hinge GAN critic on a few frames per clip (reference tokenizer.py:294-299)
        self.gan_crit = GANLoss(discriminate=gan_discriminate, num_frames=gan_frames_per_batch, **disc_kwargs) if gan_loss_weight > 0 else nn.Identity()
        self.gan_loss_weight, self.perc_loss_weight, self.quant_loss_weight = gan_loss_weight, perc_loss_weight, quant_loss_weight
        self.save_hyperparameters()

    def forward_order(self):
        """Sub-modules in execution order (trainer.execution_order lays the parameter arena out this way).  The GAN critic is NOT listed:
        its `dis_loss` branch is detached from the reconstruction, so no tensor hook can tell when its gradients are complete; unlisted
        parameters are laid out first and reduced last, by DataParallel.finish(), when every gradient is known to exist."""
        return [self.enc_layers, self.quant, self.dec_layers]

    def encode(self, video: Tensor, cond: Tensor | None = None) -> Tensor:
        return run_layers(self.enc_layers, self.enc_ext, video, cond)

    def decode(self, quant: Tensor, cond: Tensor | None = None) -> Tensor:
        cond = default(cond, quant)
        return run_layers(self.dec_layers, self.dec_ext, quant, cond)

    @torch.no_grad()
    def tokenize(self, video: Tensor, beta: float = 100., transpose: bool = True) -> Tuple[Tensor, Tensor]:
        self.eval()
        enc_video = self.encode(video)
        (quant_video, idxs), _ = self.quant(enc_video, beta=beta, transpose=transpose)
        self.train()
        return quant_video, idxs

    def forward(self, video: Tensor, beta: float = 100., transpose: bool = True) -> Tuple[Tensor, Tuple[Tensor, ...]]:
        enc_video = self.encode(video)
        with GF.deferred_lfq_loss():                       # (side stream, when enabled: the entropy loss runs under the decoder)
            (quant_video, idxs), quant_loss = self.quant(enc_video, beta=beta, transpose=transpose)
        rec_video = self.decode(quant_video)
        rec_loss = GF.mse_loss(rec_video, video)
        GF.join_wgrad()
        gen_loss = self.gan_crit(rec_video, video, train_gen=True) if self.gan_loss_weight > 0 else 0
        dis_loss = self.gan_crit(rec_video, video, train_gen=False) if self.gan_loss_weight > 0 else 0
        perc_loss = self.perc_crit(rec_video, video) if self.perc_loss_weight > 0 else 0
        # reference tokenizer.py:375-379 parses as (sum of all terms) if exists(quant_loss) else 0
        loss = (rec_loss + gen_loss * self.gan_loss_weight + dis_loss * self.gan_loss_weight + perc_loss * self.perc_loss_weight
                + quant_loss * self.quant_loss_weight) if exists(quant_loss) else 0
        return loss, (
            rec_loss,
            gen_loss if self.gan_loss_weight > 0 else 0,
            dis_loss if self.gan_loss_weight > 0 else 0,
            perc_loss if self.perc_loss_weight > 0 else 0,
            quant_loss if exists(quant_loss) and self.quant_loss_weight > 0 else 0,
        )

    def _step(self, batch: Tensor, prefix: str) -> Tensor:
        loss, aux = self(batch)
        self.log_dict({f'{prefix}_loss': loss, f'{prefix}_rec_loss': aux[0], f'{prefix}_gen_loss': aux[1],
                       f'{prefix}_dis_loss': aux[2], f'{prefix}_perc_loss': aux[3], f'{prefix}_quant_loss': aux[4]},
                      logger=True, on_step=True, sync_dist=True)
        return loss

    def training_step(self, batch: Tensor, batch_idx: int) -> Tensor:
        return self._step(batch, 'train')

    @property
    def graph_capture_safe(self) -> bool:
        """May ``Trainer(graph=True)`` record this model's training step once and replay it?  Yes without the GAN critic (its random frame
        choice is drawn per step); the rest of the step is shape-stable device work with no host round trip."""
        return not (self.gan_loss_weight > 0)

    def validation_step(self, batch: Tensor, batch_idx: int) -> Tensor:
        return self._step(batch, 'val')

    def on_validation_end(self) -> None:
        pass

    def configure_optimizers(self) -> Optimizer:
        return self.optimizer(self.parameters())
