"""Configuration surface of the reference's LightningCLI entry points (reference tokenizer.py:6-19, genie.py:6-19 and
config/tokenize.yaml) without Lightning: the same YAML layout -- ``seed_everything`` / ``model`` (constructor kwargs, blueprints as
lists of ``[name, kwargs]``, ``optimizer`` as ``class_path`` + ``init_args``) / ``data`` / ``trainer`` -- is read with PyYAML and
turned into the objects of this package.  ``python tokenizer.py fit --config config/tokenize_repaired.yaml`` is the drop-in for the
reference's command line; Lightning itself stays optional (genie/_lightning.py)."""
from __future__ import annotations

import importlib
from typing import Any, Callable, Dict

import yaml


def load_config(path: str) -> Dict[str, Any]:
    with open(path) as f:
        cfg = yaml.safe_load(f)
    if not isinstance(cfg, dict) or 'model' not in cfg:
        raise ValueError(f'{path}: expected a mapping with a `model` section (LightningCLI layout)')
    return cfg


def resolve_class(class_path: str):
    """'torch.optim.AdamW' -> the class (jsonargparse's class_path convention)."""
    mod, _, name = class_path.rpartition('.')
    if not mod:
        raise ValueError(f'class_path {class_path!r} must be a dotted path')
    return getattr(importlib.import_module(mod), name)


def optimizer_factory(spec) -> Callable:
    """``{class_path, init_args}`` -> ``params -> Optimizer`` (the reference's OptimizerCallable, tokenizer.py:22,250).  PyYAML reads
    ``1e-3`` as a string (no dot), so numeric strings are converted the way jsonargparse would."""
    if spec is None:
        from torch.optim import AdamW
        return AdamW
    if callable(spec):
        return spec
    cls = resolve_class(spec['class_path'])
    args = {k: _num(v) for k, v in (spec.get('init_args') or {}).items()}
    return lambda params: cls(params, **args)


def _num(v):
    if isinstance(v, str):
        try:
            return float(v)
        except ValueError:
            return v
    return v


def tokenizer_kwargs(model_cfg: Dict[str, Any], **overrides) -> Dict[str, Any]:
    kw = dict(model_cfg)
    kw.update(overrides)
    kw['enc_desc'] = tuple((d if isinstance(d, str) else (d[0], dict(d[1] or {}))) for d in kw['enc_desc'])
    kw['dec_desc'] = tuple((d if isinstance(d, str) else (d[0], dict(d[1] or {}))) for d in kw['dec_desc'])
    if 'optimizer' in kw:
        kw['optimizer'] = optimizer_factory(kw['optimizer'])
    for k in ('lfq_frac_sample', 'lfq_commit_weight', 'lfq_entropy_weight', 'lfq_diversity_weight', 'gan_loss_weight', 'perc_loss_weight',
              'quant_loss_weight'):
        if k in kw:
            kw[k] = float(kw[k])
    return kw


def build_tokenizer(cfg: Dict[str, Any], **overrides):
    """VideoTokenizer from the ``model`` section of a LightningCLI-style config (``cfg`` is the whole config or the section)."""
    from .tokenizer import VideoTokenizer
    model_cfg = cfg['model'] if 'model' in cfg else cfg
    return VideoTokenizer(**tokenizer_kwargs(model_cfg, **overrides))


# ----------------------------------------------------------------------------------------------------------------------------------
# command line: `python tokenizer.py fit --config cfg.yaml [--trainer.max_steps 10 --data.batch_size 4 ...]`
# ----------------------------------------------------------------------------------------------------------------------------------
def _set_dotted(cfg: Dict[str, Any], dotted: str, value: str) -> None:
    keys = dotted.split('.')
    node = cfg
    for k in keys[:-1]:
        node = node.setdefault(k, {})
    node[keys[-1]] = yaml.safe_load(value)


def parse_args(argv) -> Dict[str, Any]:
    """LightningCLI's argument shape: a sub-command, any number of ``--config file``, then dotted overrides ``--a.b.c value``."""
    if not argv or argv[0] not in ('fit', 'validate'):
        raise SystemExit('usage: <entry point> {fit,validate} --config CONFIG.yaml [--section.key value ...]')
    cfg: Dict[str, Any] = {'_command': argv[0]}
    i = 1
    while i < len(argv):
        a = argv[i]
        if not a.startswith('--') or i + 1 >= len(argv):
            raise SystemExit(f'cannot parse argument {a!r}')
        if a in ('--config', '-c'):
            new = load_config(argv[i + 1])
            for k, v in new.items():
                cfg[k] = {**cfg.get(k, {}), **v} if isinstance(v, dict) and isinstance(cfg.get(k), dict) else v
        else:
            _set_dotted(cfg, a[2:], argv[i + 1])
        i += 2
    if 'model' not in cfg:
        raise SystemExit('no --config given (a `model` section is required)')
    return cfg


def build_datamodule(data_cfg: Dict[str, Any] | None):
    """``data:`` section -> LightningPlatformer2D (the reference's data module, tokenizer.py:15) or, when ``synthetic`` is set or the
    data root does not exist, seeded random clips."""
    import os

    from .dataset import LightningPlatformer2D, LightningSynthetic
    d = dict(data_cfg or {})
    synthetic = d.pop('synthetic', False)
    loader_keys = ('batch_size', 'num_workers', 'train_shuffle', 'val_shuffle', 'val_batch_size', 'pin_memory')
    if synthetic or not os.path.isdir(str(d.get('root', ''))):
        shape = d.get('shape', (3, d.get('num_frames', 16), *(d.get('frame_size', (64, 64)))))
        return LightningSynthetic(num_clips=d.get('num_clips', 1024), shape=tuple(shape), seed=d.get('seed', 0),
                                  **{k: d[k] for k in loader_keys if k in d})
    d.pop('shape', None); d.pop('num_clips', None); d.pop('frame_size', None); d.pop('seed', None)
    return LightningPlatformer2D(**d)


def build_genie(cfg: Dict[str, Any]):
    """``model:`` section of a Genie config: ``tokenizer`` (a VideoTokenizer model section), optional ``tokenizer_ckpt``, and the
    keyword arguments of ``genie.Genie``."""
    import torch

    from .genie import Genie
    m = dict(cfg['model'] if 'model' in cfg else cfg)
    tok = build_tokenizer(m.pop('tokenizer'))
    ckpt = m.pop('tokenizer_ckpt', None)
    if ckpt:
        sd = torch.load(ckpt, map_location='cpu')
        tok.load_state_dict(sd.get('state_dict', sd))
    if 'optimizer' in m:
        m['optimizer'] = optimizer_factory(m['optimizer'])
    for k in ('enc_desc', 'dec_desc', 'dyn_desc'):
        if k in m:
            m[k] = tuple((d if isinstance(d, str) else (d[0], dict(d[1] or {}))) for d in m[k])
    if 'inp_shape' in m and not isinstance(m['inp_shape'], int):
        m['inp_shape'] = tuple(m['inp_shape'])
    return Genie(tok, **m)


def main(kind: str, argv=None) -> int:
    """Entry point behind ``tokenizer.py`` / ``genie.py``.  With Lightning installed this defers to ``LightningCLI`` exactly like the
    reference (tokenizer.py:13-16); without it, the same configuration drives ``genie.trainer.Trainer``."""
    import os
    import sys

    import torch
    argv = list(sys.argv[1:] if argv is None else argv)
    from ._lightning import HAVE_LIGHTNING
    if HAVE_LIGHTNING and os.environ.get('GENIE_USE_LIGHTNING', '1') != '0':        # pragma: no cover - lightning is not in this image
        from lightning.pytorch.cli import LightningCLI

        from .dataset import LightningPlatformer2D
        from .genie import Genie
        from .tokenizer import VideoTokenizer
        LightningCLI(VideoTokenizer if kind == 'tokenizer' else Genie, LightningPlatformer2D, args=argv)
        return 0
    cfg = parse_args(argv)
    if not torch.cuda.is_available():
        raise SystemExit('the genie hot path runs on the MI355X only (no CPU fallback)')
    world = int(os.environ.get('WORLD_SIZE', '1'))
    torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', '0')))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('nccl', device_id=torch.device('cuda', torch.cuda.current_device()))
    if 'seed_everything' in cfg:
        torch.manual_seed(int(cfg['seed_everything']))
    model = build_tokenizer(cfg) if kind == 'tokenizer' else build_genie(cfg)
    data = build_datamodule(cfg.get('data'))
    from .trainer import Trainer
    trainer = Trainer(**(cfg.get('trainer') or {}))
    if cfg['_command'] == 'fit':
        # `ckpt_path` (LightningCLI's `fit --ckpt_path`): resume weights, AdamW moments and the step counter from a last.ckpt;
        # replicas are synchronised from rank 0 and then reseeded per rank (seed_everything + rank)
        trainer.fit(model, data, ckpt_path=cfg.get('ckpt_path'), seed=int(cfg.get('seed_everything', 0)))
    else:
        from .trainer import DataParallel
        model.to('cuda')
        data.setup('fit')
        trainer.validate(model, data.val_dataloader(), DataParallel(torch.zeros(1, device='cuda')))
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()
    return 0
