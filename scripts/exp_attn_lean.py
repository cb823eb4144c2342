#!/usr/bin/env python
"""A/B of the two MFMA attention kernel families on one box, alternating (run via gpurun):

  mode 0 = attention.hip's general kernels (3 / 2 waves per SIMD at d_head 64)
  mode 7 = attention_lean.hip (4 / 3 waves per SIMD); 1 / 2 / 4 = forward / dQ / dK-dV alone; + 8 = no s_setprio around the MFMA
  clusters; + 16 = deferred running maximum in the forward; + 32 = plain (sequence * tiles, head) grid instead of the XCD-aware one;
  + 128 (with 16) = the sum-triggered form of the deferred maximum ; + 256 = sequence-resident forward for S <= 1024 (parity-green, measured slower: off; round 6 default: 151)

for the spatial ST-attention shapes of scripts/microbench.py (MFMA-bound: S >= 1024; traffic-bound: S = 256 / 64).  Writes
gpurun_out/attn_lean_ab.json; every line is also printed."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'scripts'))
import microbench as mb   # noqa: E402

from genie import _hip    # noqa: E402


def main():
    lib = _hip.load_library()
    iters = int(os.environ.get('AB_ITERS', 30))
    only = os.environ.get('AB_ONLY', 'spatial S=4096,spatial S=1024,spatial S=256,spatial S=64').split(',')
    modes = [int(m) for m in os.environ.get('AB_MODES', '151,407').split(',')]
    reps = int(os.environ.get('AB_REPS', 2))
    base_report = mb.report
    rows = []
    for rep in range(reps):
        for mode in modes:
            lib.genie_attention_lean_mode(mode)

            def report(section, name, ms, **kw):
                if 'attention' in name:
                    base_report(section, f'lean={mode} rep={rep} | {name}', ms, **kw)
                    rows.append(mb.RESULTS[-1])
            mb.report = report
            mb.bench_attn(iters, only=only)
    lib.genie_attention_lean_mode(151)
    os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
    with open(os.path.join(ROOT, 'gpurun_out', 'attn_lean_ab.json'), 'w') as f:
        json.dump(rows, f, indent=1)
    # summary: best of the repetitions per (mode, case)
    best = {}
    for r in rows:
        mode = r['name'].split()[0]
        case = r['name'].split('| ', 1)[1]
        k = (case, mode)
        best[k] = max(best.get(k, 0.0), r.get('tflops', 0.0))
    for case in sorted({c for c, _ in best}):
        print('SUMMARY', case, {m: best[(c, m)] for (c, m) in sorted(best) if c == case}, flush=True)


if __name__ == '__main__':
    main()
