#!/bin/bash
# round 3, closing evidence on the last build: full -m gpu suite, default bench line, rocprofv3 profile of the bench
set -u
OUT=gpurun_out/r3final2; mkdir -p $OUT
rm -f gpurun_out/parity_report.jsonl
timeout 1500 python -m pytest tests -q -m gpu --durations=5 2>&1 | tail -14 > $OUT/pytest_full.log
grep -E "passed|failed" $OUT/pytest_full.log | tail -2
timeout 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r3final2/bench_default.json') if l.startswith('{')][-1])
print(d['value'], d['ms_per_step'], d['roofline'], d['power_cap']['random_operands'], d['power_cap']['zero_operands'], d['wgrad_side_stream'])
PY
bash scripts/profile_bench.sh r03 64 > $OUT/prof.log 2>&1
python - <<'PY'
import json
s=json.load(open('gpurun_out/prof_r03/r03_summary.json'))
print(s['meta'], s['duration_agreement'])
PY
