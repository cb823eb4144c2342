"""Import the upstream reference (``/root/reference``) on a box without lightning/torchvision/cv2.

TEST INFRASTRUCTURE ONLY.  Used by ``tests/golden/make_golden.py`` (fixture generation) and by the
``-m "not gpu"`` tests that pin ``oracle/`` against the real reference while it is present (this
container).  ``/root/reference`` does not exist on the GPU box: everything that runs there uses the
committed fixtures instead.

The reference eagerly imports three packages that are absent here:
``lightning`` (genie/tokenizer.py:13), ``torchvision`` (genie/tokenizer.py:6, genie/module/loss.py:4)
and ``cv2`` (genie/module/data.py:4-8).  None of them takes part in the hot path, so they are
replaced by inert stub modules before ``import genie``.
"""
import os
import sys
import types

import torch.nn as nn

REFERENCE_ROOT = os.environ.get('GENIE_REFERENCE_ROOT', '/root/reference')


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, 'genie'))


def _stub(name: str, **attrs) -> types.ModuleType:
    mod = types.ModuleType(name)
    mod.__dict__.update(attrs)
    sys.modules[name] = mod
    return mod


def import_reference():
    """Return the reference ``genie`` package, imported under a private name space.

    The reference package is called ``genie`` -- the same name as the drop-in package of this repo --
    so it is imported with ``/root/reference`` first on ``sys.path`` and then moved to
    ``sys.modules['ref_genie*']`` so that both can live in one process.
    """
    if 'ref_genie' in sys.modules:
        return sys.modules['ref_genie']
    if not reference_available():
        raise RuntimeError(f'reference not present at {REFERENCE_ROOT}')

    class LightningModule(nn.Module):
        def save_hyperparameters(self, *a, **k):
            pass

        def log_dict(self, *a, **k):
            pass

    class LightningDataModule:
        def __init__(self, *a, **k):
            pass

    saved = {k: v for k, v in sys.modules.items() if k == 'genie' or k.startswith('genie.')}
    for k in saved:
        del sys.modules[k]

    if 'lightning' not in sys.modules:
        _stub('lightning', LightningModule=LightningModule, LightningDataModule=LightningDataModule)
    if 'torchvision' not in sys.modules:
        tv = _stub('torchvision')
        tv.models = _stub('torchvision.models', get_model=lambda *a, **k: None)
        tv.datasets = _stub('torchvision.datasets', Kinetics=object)
        tv.transforms = _stub('torchvision.transforms', Compose=object, Lambda=object, Resize=object)
    if 'cv2' not in sys.modules:
        _stub('cv2', VideoCapture=object, CAP_PROP_FRAME_COUNT=7, CAP_PROP_POS_FRAMES=1,
              COLOR_BGR2RGB=4, cvtColor=lambda *a, **k: None, resize=lambda *a, **k: None)

    sys.path.insert(0, REFERENCE_ROOT)
    try:
        import genie as ref  # noqa: F401  (the reference package)
    finally:
        sys.path.remove(REFERENCE_ROOT)

    ours = {}
    for k in list(sys.modules):
        if k == 'genie' or k.startswith('genie.'):
            ours['ref_' + k] = sys.modules.pop(k)
    sys.modules.update(ours)
    sys.modules.update(saved)
    return sys.modules['ref_genie']


def _import_extra(path: str) -> None:
    """Import a reference sub-module that ``genie/__init__`` does not pull in (e.g. ``module.data``): the reference's absolute imports
    (``from genie.utils import ...``) need ``genie`` to BE the reference while that happens, so the two packages swap places for the
    duration of the import."""
    import importlib
    ours = {k: sys.modules.pop(k) for k in list(sys.modules) if k == 'genie' or k.startswith('genie.')}
    refs = {k: v for k, v in sys.modules.items() if k == 'ref_genie' or k.startswith('ref_genie.')}
    for k, v in refs.items():
        sys.modules[k[4:]] = v
    sys.path.insert(0, REFERENCE_ROOT)
    try:
        importlib.import_module('genie.' + path)
    finally:
        sys.path.remove(REFERENCE_ROOT)
        for k in list(sys.modules):
            if k == 'genie' or k.startswith('genie.'):
                sys.modules['ref_' + k] = sys.modules.pop(k)
        sys.modules.update(ours)


def ref_module(path: str):
    """``ref_module('module.video')`` -> the reference's ``genie.module.video``."""
    import_reference()
    key = 'ref_genie' + ('.' + path if path else '')
    if key not in sys.modules:
        _import_extra(path)
    return sys.modules[key]
