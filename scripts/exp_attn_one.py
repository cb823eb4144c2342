import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'open-genie_amd')]
import scripts.microbench as mb
mb.bench_attn(3, only=('lam spatial S=4096',))
