#!/bin/bash
# round 3, after the last kernel-source change: kernel / norm / graph tests, default bench line, rocprofv3 profile of the bench
set -u
OUT=gpurun_out/r3final3; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_graph.py tests/test_gpu_properties.py -q -m gpu 2>&1 | tail -3 | cut -c1-200
timeout 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r3final3/bench_default.json') if l.startswith('{')][-1])
print(d['value'], d['ms_per_step'], d['roofline'])
PY
bash scripts/profile_bench.sh r03 64 > $OUT/prof.log 2>&1
python - <<'PY'
import json
s=json.load(open('gpurun_out/prof_r03/r03_summary.json'))
print(s['meta'], s['duration_agreement'])
PY
