#!/usr/bin/env python
"""Train / validate the full Genie (frozen tokenizer + latent-action model + MaskGIT dynamics) from a YAML configuration:

    python genie.py fit --config config/genie.yaml [--trainer.max_steps 100 ...]
"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'open-genie_amd'))

if __name__ == '__main__':
    sys.modules.pop('genie', None)
    from genie.cli import main
    raise SystemExit(main('genie'))
else:
    # imported as `genie` because the repository root precedes open-genie_amd on sys.path: hand over to the package of that name
    import importlib
    sys.modules.pop(__name__, None)
    sys.modules[__name__] = importlib.import_module(__name__)
