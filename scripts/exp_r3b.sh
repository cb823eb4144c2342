#!/bin/bash
# round 3, GPU call 3: full -m gpu suite + the default bench line
set -u
OUT=gpurun_out/r3b; mkdir -p $OUT
rm -f gpurun_out/parity_report.jsonl
timeout 1500 python -m pytest tests -q -m gpu -x --durations=12 2>&1 | tail -40 > $OUT/pytest_full.log
tail -4 $OUT/pytest_full.log
timeout 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
cat $OUT/bench_default.json | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print({k: d[k] for k in ('value', 'ms_per_step', 'model_tflops_per_gpu')}); print(d.get('roofline')); print(d.get('conv_kernels')); print(d.get('wgrad_side_stream')); print(d.get('cpu_baseline'))"
