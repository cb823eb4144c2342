#!/bin/bash
# round 3, GPU call 12: igemm3w (256 x 256 kw-triple kernel) -- staggered DMA issue (GENIE_TRI_VAR=1) vs production, plus timing ablations
# (2: no DMA in the loop, 3: no MFMA, 4: DMA + barriers only; results of 2..4 are wrong by construction)
set -u
OUT=gpurun_out/r3j; mkdir -p $OUT
GENIE_TRI_VAR=1 timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "triple_wide" 2>&1 | tail -3 > $OUT/pytest_var1.log
tail -2 $OUT/pytest_var1.log
export MB_BATCH=64 MB_NO_WGRAD=1
for rep in 1 2; do
for v in 0 1; do
  GENIE_TRI_VAR=$v MB_FILTER="res 256->256 k3 @16x32x32" timeout 300 python scripts/microbench.py conv --iters 20 --out $OUT/mb_v${v}_$rep.json 2>&1 | grep -E "fwd|dgrad" | sed "s/^/var=$v /" | cut -c1-200
done
done
for v in 2 3 4; do
  GENIE_TRI_VAR=$v MB_FILTER="res 256->256 k3 @16x32x32" timeout 300 python scripts/microbench.py conv --iters 20 --out $OUT/mb_v${v}.json 2>&1 | grep -E "fwd|dgrad" | sed "s/^/var=$v /" | cut -c1-200
done
for v in 0 1; do
  GENIE_TRI_VAR=$v timeout 600 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-in-order-pass 2>/dev/null | tail -1 > $OUT/bench_v$v.json
  python - <<PY
import json
d=json.loads(open('$OUT/bench_v$v.json').read())
print('bench var=$v', d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['frac'])
PY
done
