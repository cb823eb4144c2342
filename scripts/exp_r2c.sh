#!/bin/bash
# round-2 experiment batch C: GroupNorm sample chunking (memory-side cache reuse) and clips-per-GPU sweep
set -u
OUT=gpurun_out/r2c; mkdir -p $OUT
python -m pytest tests/test_gpu_gan.py tests/test_gpu_genie.py tests/test_gpu_maskgit.py -m gpu -q --no-header -rf --timeout 900 -p no:cacheprovider > $OUT/pytest.log 2>&1
grep -E "^(FAILED|ERROR)|passed|failed" $OUT/pytest.log | head -20
for MB in 0 40 72 140; do
  GENIE_GN_CHUNK_MB=$MB python scripts/microbench.py hbm --iters 20 --out $OUT/mb_gn_$MB.json > /dev/null 2>&1
  python - <<PY
import json
r = json.load(open('$OUT/mb_gn_$MB.json'))['results']
print('GN chunk $MB MB:', [(x['name'][:28], x['ms']) for x in r if 'GroupNorm' in x.get('name','')])
PY
done
for MB in 0 72; do
  GENIE_GN_CHUNK_MB=$MB python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-events > $OUT/bench_gn_$MB.log 2>&1
  python - <<PY
import json
l=[x for x in open('$OUT/bench_gn_$MB.log') if x.startswith('{')]
d=json.loads(l[-1]); print('bench GN chunk $MB:', d['ms_per_step'], d['value'])
PY
done
for B in 4 16; do
  python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-kernel-events --batch $B > $OUT/bench_b$B.log 2>&1
  python - <<PY
import json
l=[x for x in open('$OUT/bench_b$B.log') if x.startswith('{')]
d=json.loads(l[-1]) if l else None; print('bench batch $B:', d and (d['ms_per_step'], d['value']))
PY
done
