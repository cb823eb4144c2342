"""One training step captured in a hipGraph.

A step of the models here is 10^3 kernel launches (MAGVIT2 tokenizer: ~2500 incl. the element-wise torch ops of the loss) issued from
Python at ~15-20 us each: at 64 clips per GPU the GPU is the slower side (470 ms of kernels), at the 4-8 clips per GPU of a
strong-scaling run, for the DynamicsModel (12 ms of kernels) and on a slow host it is not.  ``GraphedTrainStep`` records forward, backward
and the arena's AdamW once (``torch.cuda.CUDAGraph`` = hipGraph on ROCm) and replays them with one call per step.

What makes the step capturable: every launch goes to the current stream through the C ABI with device pointers only; scratch comes from
torch's allocator (the graph's private pool during capture); tap / step tables and weight packs are device tensors cached at first use
(the warm-up steps); nothing on the path reads a device value on the host; and the only launch arguments that change from step to step --
AdamW's step count and the learning rate -- live in device memory (``genie_adamw_step_graph``).  Shapes are fixed: a new batch is copied
into the captured input buffers.  Data-dependent shapes cannot be captured: the DynamicsModel's default loss gathers the masked rows
(their number varies); ``compute_loss(..., fixed_rows=True)`` is its shape-stable form, with the mask as an input of the step.
"""
from typing import Callable, Optional

import torch
from torch import Tensor

from .trainer import ParamArena


class GraphedTrainStep:
    def __init__(self, model: torch.nn.Module, arena: ParamArena, example, loss_fn: Optional[Callable] = None, lr: float = 1e-3,
                 betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 1e-2, warmup: int = 2) -> None:
        """`example`: a batch of the shape every step will have (its values are used by the `warmup` eager steps and by the capture
        step -- all of them REAL optimiser steps on `example`; ``warmup=0`` when the caller has already trained eagerly for a step or
        two, as ``Trainer.fit(graph=True)`` does).  `loss_fn(model, batch)` returns the loss (or a tuple whose first element is the
        loss); default ``model(batch)``."""
        from . import functional as GF
        self._single = torch.is_tensor(example)
        examples = (example,) if self._single else tuple(example)
        if not examples or not all(torch.is_tensor(e) and e.is_cuda for e in examples):
            raise ValueError('GraphedTrainStep needs CUDA tensors (no CPU path)')
        example = examples[0]
        if GF.ASYNC_WGRAD:
            raise RuntimeError('GraphedTrainStep: capture the in-order step (functional.ASYNC_WGRAD = 0); the side streams are not part of it')
        self.model, self.arena = model, arena
        self.loss_fn = loss_fn if loss_fn is not None else (lambda m, b: m(b))
        self.betas, self.eps = betas, eps
        self.batches = tuple(e.clone() for e in examples)   # the captured input buffers (a batch may be several tensors: tokens, actions, mask)
        self.batch = self.batches[0]
        arena.set_graph_hyperparameters(lr, weight_decay)
        self.steps_done = 0

        def step():
            out = self.loss_fn(self.model, self.batch if self._single else self.batches)
            loss = out[0] if isinstance(out, (tuple, list)) else out
            loss.backward()
            arena.adamw_step(betas=self.betas, eps=self.eps, graph_safe=True)
            return loss.detach(), out

        side = torch.cuda.Stream(device=example.device)
        side.wait_stream(torch.cuda.current_stream(example.device))
        with torch.cuda.stream(side):                       # warm-up off the default stream, as graph capture wants it
            for _ in range(max(0, warmup)):                 # 0: the caller has already run this step eagerly (tables, packs and scratch exist)
                step()
                self.steps_done += 1
        torch.cuda.current_stream(example.device).wait_stream(side)
        torch.cuda.synchronize(example.device)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.loss, self.out = step()
        self.graph.replay()                                 # capture executes nothing: this replay IS the step the capture recorded
        self.steps_done += 1
        arena.mark_updated()

    def set_lr(self, lr: float, weight_decay: Optional[float] = None) -> None:
        st = self.arena._opt_state
        st[1:2].fill_(float(lr))
        if weight_decay is not None:
            st[2:3].fill_(float(weight_decay))

    def __call__(self, *batch: Tensor) -> Tensor:
        """One optimiser step on `batch` (the tensors of the example, same shapes / dtypes).  Returns the captured loss tensor (overwritten
        by the next call)."""
        if len(batch) == 1 and not torch.is_tensor(batch[0]):
            batch = tuple(batch[0])
        if len(batch) != len(self.batches):
            raise ValueError(f'GraphedTrainStep: {len(batch)} tensors, captured {len(self.batches)}')
        for src, dst in zip(batch, self.batches):
            if src.shape != dst.shape or src.dtype != dst.dtype:
                raise ValueError(f'GraphedTrainStep: batch {tuple(src.shape)} {src.dtype} != captured {tuple(dst.shape)} {dst.dtype}')
            dst.copy_(src, non_blocking=True)
        self.graph.replay()
        self.arena.step_count += 1
        self.steps_done += 1
        self.arena.mark_updated()
        return self.loss
