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
