import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'open-genie_amd')):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    config.addinivalue_line('markers', 'default_grads: run with the default DIRECT_PARAM_GRADS mode (arena-managed parameters only)')


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason='no GPU visible')
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)


@pytest.fixture(autouse=True)
def _direct_param_grads(request):
    """GPU tests run with direct gradient accumulation for every parameter (the mode the arena-based trainer and bench.py use,
    including the fused residual-block node) unless marked `default_grads`; the default 'arena' mode -- gradients returned to
    autograd for parameters no arena manages -- has its own tests (tests/test_gpu_tokenizer.py::test_default_grad_mode_*)."""
    if 'gpu' not in request.keywords:
        yield
        return
    from genie import functional as GF
    old = GF.DIRECT_PARAM_GRADS
    GF.DIRECT_PARAM_GRADS = 'arena' if 'default_grads' in request.keywords else 'all'
    try:
        yield
    finally:
        GF.DIRECT_PARAM_GRADS = old
