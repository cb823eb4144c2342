#!/usr/bin/env python
"""Train / validate the VideoTokenizer from a YAML configuration -- the command line of the reference's tokenizer.py:

    python tokenizer.py fit --config config/tokenize_repaired.yaml [--trainer.max_steps 100 --data.batch_size 8 ...]
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 tokenizer.py fit --config ...     (one rank per GPU)
"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'open-genie_amd'))

if __name__ == '__main__':
    from genie.cli import main
    raise SystemExit(main('tokenizer'))
