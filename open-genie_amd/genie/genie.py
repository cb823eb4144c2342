"""``Genie``: tokenizer + latent-action model + MaskGIT dynamics as one trainable / sampling model.

The reference class (genie/genie.py:18-181) states the intent but cannot be constructed at HEAD: ``__init__`` reads attributes
nobody sets (``self.enc_desc``, ``self.d_codebook``, ``TEST_DESC`` ...), ``compute_loss`` feeds the ``(quant, idxs)`` tuple of
``tokenize`` to the dynamics model, and ``forward`` stacks token grids along a new axis.  This is the R-genie repaired form -- the
same public surface (constructor order ``tokenizer, optimizer, img_prompt``; ``forward(prompt, actions, num_frames,
steps_per_frame)``; ``compute_loss(video) -> (loss, named aux losses)``; Lightning step hooks) with these documented repairs:

1. everything ``__init__`` read from undefined attributes is a keyword argument with the README's values as defaults;
2. the dynamics model trains on the tokenizer's INDEX grid (``tokenize(video)[1]``), under ``no_grad`` with the tokenizer frozen;
3. the latent-action model labels every VIDEO frame while the tokenizer compresses time by ``tf``: latent frame j takes the action
   of its last video frame, ``act_id[:, (j + 1) * tf - 1]``;
4. generation appends each generated token frame (``generate`` returns context + new frame) and hands the dynamics model as many
   actions as it has context frames; the indices are turned back into the {-1, +1} codes the tokenizer's decoder consumes.
"""
from __future__ import annotations

from typing import Callable, Iterable, Tuple

import torch
from torch import Tensor
from torch.optim import AdamW, Optimizer

from ._lightning import LightningModule
from .action import LatentAction
from .blueprints import DYNAMICS_DESC, LATENT_ACT_DEC, LATENT_ACT_ENC
from .dynamics import DynamicsModel
from .tokenizer import VideoTokenizer
from .utils import Blueprint, default

OptimizerCallable = Callable[[Iterable], Optimizer]


class Genie(LightningModule):
    def __init__(self, tokenizer: VideoTokenizer, optimizer: OptimizerCallable = AdamW, img_prompt: Tensor | None = None, *,
                 enc_desc: Blueprint = LATENT_ACT_ENC, dec_desc: Blueprint = LATENT_ACT_DEC, d_codebook: int = 8, inp_channels: int = 3,
                 inp_shape: int | Tuple[int, int] = (64, 64), ker_size: int | Tuple[int, int] = 3, n_embd: int = 256, n_codebook: int = 1,
                 lfq_bias: bool = True, lfq_frac_sample: float = 1., lfq_commit_weight: float = 0.25, lfq_entropy_weight: float = 0.1,
                 lfq_diversity_weight: float = 1., dyn_desc: Blueprint = DYNAMICS_DESC, tok_codebook: int | None = None,
                 act_codebook: int | None = None, embed_dim: int = 512, device_masks: bool = False) -> None:
        super().__init__()
        # device_masks (not in the reference): draw the MaskGIT training mask on the DEVICE (rate ~ U(0.5, 1), mask = rand < rate: the
        # distribution of the reference's Bernoulli(uniform(0.5, 1)) from the device generator instead of the host's) and use the
        # shape-stable dynamics loss -- the training step then has no host round trip and can be replayed as a hipGraph (Trainer(graph=True))
        self.device_masks = bool(device_masks)
        self.tokenizer = tokenizer
        for p in self.tokenizer.parameters():                     # pre-trained and frozen (reference genie.py:34 "Pre-trained video tokenizer")
            p.requires_grad_(False)
        self.latent_action = LatentAction(enc_desc, dec_desc, d_codebook=d_codebook, inp_channels=inp_channels, inp_shape=inp_shape,
                                          ker_size=ker_size, n_embd=n_embd, n_codebook=n_codebook, lfq_bias=lfq_bias,
                                          lfq_frac_sample=lfq_frac_sample, lfq_commit_weight=lfq_commit_weight,
                                          lfq_entropy_weight=lfq_entropy_weight, lfq_diversity_weight=lfq_diversity_weight)
        self.tok_codebook = default(tok_codebook, tokenizer.quant.codebook_size)
        self.act_codebook = default(act_codebook, 2 ** d_codebook)
        self.dynamics_model = DynamicsModel(desc=dyn_desc, tok_vocab=self.tok_codebook, act_vocab=self.act_codebook, embed_dim=embed_dim)
        self.optimizer = optimizer
        self.img_prompt = img_prompt
        self.save_hyperparameters(ignore=['tokenizer'])

    def forward_order(self):
        return [self.latent_action, self.dynamics_model]

    # -- helpers ---------------------------------------------------------------------------------------------------------------
    def _token_grid(self, video: Tensor) -> Tensor:
        """(B, t', h', w') int64 token indices of a clip.  The reference's ``idxs.squeeze()`` (quantization.py:110) drops EVERY size-1
        axis -- a batch of one, but also a single latent frame (an image prompt) -- so the grid is rebuilt from the quantised latent's
        shape (B, d, t', h', w') instead of guessed from ``idxs.dim()``."""
        quant, idxs = self.tokenizer.tokenize(video)
        if self.tokenizer.quant.num_codebooks != 1 or quant.dim() != 5:
            raise ValueError(f'Genie works on one-codebook tokenizers with (B, d, t, h, w) latents; got {tuple(quant.shape)} / {tuple(idxs.shape)}')
        grid = (quant.shape[0], *quant.shape[2:])
        if idxs.numel() != grid[0] * grid[1] * grid[2] * grid[3]:
            raise ValueError(f'token indices {tuple(idxs.shape)} do not fill the latent grid {grid}')
        return idxs.reshape(grid)

    def _codes(self, tokens: Tensor) -> Tensor:
        """token indices (B, t, h, w) -> the {-1, +1} latent (B, d, t, h, w) the decoder consumes (MSB-first bits, quantization.py:72)."""
        q = self.tokenizer.quant
        codes = ((tokens.unsqueeze(-1) & q.bit_mask) != 0).float() * 2 - 1
        if not isinstance(q.proj_out, torch.nn.Identity):             # a projecting LFQ hands the decoder proj_out(code) (quantization.py:104-108)
            codes = q.proj_out(codes.to(q.proj_out.weight.dtype))
        return codes.permute(0, 4, 1, 2, 3).contiguous()

    # -- inference -------------------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def forward(self, prompt: Tensor, actions: Tensor, num_frames: int | None = None, steps_per_frame: int = 25) -> Tensor:
        """Generate `num_frames` further latent frames from an image / clip prompt and a sequence of latent-action ids, then decode
        (reference genie.py:66-103).  `actions`: (B, >= context frames) int64."""
        if actions.dim() == 1:
            actions = actions.unsqueeze(0)
        num_frames = default(num_frames, actions.shape[1])
        match prompt.dim():
            case 3: prompt = prompt[:, None, None]                 # 'b h w -> b 1 1 h w'
            case 4: prompt = prompt[:, :, None]                    # 'b c h w -> b c 1 h w'
            case 5: pass
            case _: raise ValueError('Prompt must have 3, 4 or 5 dimensions')
        tokens = self._token_grid(prompt)
        for _ in range(num_frames):
            ctx = tokens.shape[1]
            if actions.shape[1] < ctx:
                raise ValueError(f'{actions.shape[1]} actions for {ctx} context frames')
            tokens = self.dynamics_model.generate(tokens, actions[:, :ctx], steps=steps_per_frame)
        return self.tokenizer.decode(self._codes(tokens))

    # -- training --------------------------------------------------------------------------------------------------------------
    def compute_loss(self, video: Tensor):
        tokens = self._token_grid(video)                                        # frozen tokenizer, no graph
        act_id, act_loss, (act_rec_loss, act_q_loss) = self.latent_action(video)
        act_id = act_id.reshape(video.shape[0], video.shape[2])                 # (LFQ's squeeze() drops a batch / a clip of one)
        tf = video.shape[2] // tokens.shape[1]                                  # the tokenizer's time compression
        if tf < 1 or tf * tokens.shape[1] != video.shape[2]:
            raise ValueError(f'{video.shape[2]} video frames do not map onto {tokens.shape[1]} latent frames')
        if self.device_masks:
            rate = torch.empty((), device=tokens.device).uniform_(0.5, 1.)
            mask = torch.rand(tokens.shape, device=tokens.device) < rate
            dyn_loss = self.dynamics_model.compute_loss(tokens, act_id[:, tf - 1::tf].detach(), mask=mask, fixed_rows=True)
        else:
            dyn_loss = self.dynamics_model.compute_loss(tokens, act_id[:, tf - 1::tf].detach())
        loss = act_loss + dyn_loss
        return loss, (('act_loss', act_loss), ('dyn_loss', dyn_loss), ('act_rec_loss', act_rec_loss), ('act_q_loss', act_q_loss))

    def _step(self, batch: Tensor, prefix: str) -> Tensor:
        loss, aux = self.compute_loss(batch)
        self.log_dict({f'{prefix}_loss': loss, **{f'{prefix}/{k}': v for k, v in aux}}, logger=True, on_step=True, sync_dist=True)
        return loss

    @property
    def graph_capture_safe(self) -> bool:
        """Trainer(graph=True)?  Only with ``device_masks``: the default dynamics mask is drawn on the host inside compute_loss and its
        masked rows are gathered with a data-dependent count -- a replay would reuse ONE mask."""
        return self.device_masks

    def training_step(self, batch: Tensor, batch_idx: int) -> Tensor:
        return self._step(batch, 'train')

    def validation_step(self, batch: Tensor, batch_idx: int) -> Tensor:
        return self._step(batch, 'val')

    def configure_optimizers(self) -> Optimizer:
        return self.optimizer([p for p in self.parameters() if p.requires_grad])
