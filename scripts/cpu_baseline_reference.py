#!/usr/bin/env python
"""`cpu_baseline` of bench.py with kind = "reference": the REAL reference modules (/root/reference, imported through oracle/ref_import.py)
doing the benchmark's training step on this container's host cores.  /root/reference does not exist on the GPU boxes (their bench line
carries kind = "port", the oracle restatement), so this record is taken in the build container and committed next to it:

    python scripts/cpu_baseline_reference.py [r06]  ->  profiles/r06_cpu_baseline_reference.json

bench.py echoes the newest committed record next to the on-box port (`cpu_baseline.reference_record`)."""
import json
import os
import platform
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    os.environ['GENIE_CPU_BASELINE_MIN_TIMED'] = '2'      # the cold first step (oneDNN primitive creation) is not the rate: 1 warm-up + 2 timed steps
    import bench
    r = bench.cpu_baseline(budget_s=600.0)
    r['host'] = {'cpus': os.cpu_count(), 'machine': platform.processor() or platform.machine(),
                 'where': 'build container (no GPU); /root/reference modules through oracle/ref_import.py'}
    tag = sys.argv[1] if len(sys.argv) > 1 else 'r06'
    out = os.path.join(ROOT, 'profiles', f'{tag}_cpu_baseline_reference.json')
    json.dump(r, open(out, 'w'), indent=1)
    print(json.dumps(r))


if __name__ == '__main__':
    main()
