#!/bin/bash
# round-2 batch E: narrow conv kernel tests + microbench, default bench, rocprof stats at the new default configuration
set -u
OUT=gpurun_out/r2e; mkdir -p $OUT
python -m pytest tests/test_gpu_kernels.py -k "narrow or conv_forward_dgrad" tests/test_gpu_tokenizer.py -m gpu -q --no-header -rf --timeout 900 -p no:cacheprovider > $OUT/pytest.log 2>&1
grep -E "^(FAILED|ERROR)|passed|failed" $OUT/pytest.log | head -20
python scripts/microbench.py hbm --iters 20 --out $OUT/mb_hbm.json > /dev/null 2>&1
python - <<PY
import json
for x in json.load(open('$OUT/mb_hbm.json'))['results']:
    if 'name' in x: print(x['name'][:60], x.get('ms'), x.get('gbps'), x.get('hbm_frac'))
PY
python bench.py > $OUT/bench_default.log 2>&1; tail -c 3000 $OUT/bench_default.log
bash scripts/profile_bench.sh r02a 32 > $OUT/profile.log 2>&1; tail -5 $OUT/profile.log
