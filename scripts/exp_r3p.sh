#!/bin/bash
# round 3, GPU call: wgrad3l skips the chunks of time-padding frames (GENIE_W3_TRIM=0/1): tests, microbench A/B, bench A/B
set -u
OUT=gpurun_out/r3p; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_transpose.py tests/test_gpu_properties.py -q -m gpu -x 2>&1 | tail -3 | cut -c1-300
timeout 900 python -m pytest tests/test_gpu_tokenizer.py -q -m gpu -x -k "full or layer" 2>&1 | tail -3 | cut -c1-300
export MB_BATCH=64
for v in 0 1 0 1; do
for f in "res 256->256 k3 @16x32x32" "res 128->128 k3 @16x64x64"; do
  GENIE_W3_TRIM=$v MB_FILTER="$f" timeout 300 python scripts/microbench.py conv --iters 20 --out $OUT/mb_$v.json 2>&1 | grep -E "wgrad" | sed "s/^/w3trim=$v /" | cut -c1-200
done
done
for v in 0 1 0 1; do
  GENIE_W3_TRIM=$v timeout 600 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-in-order-pass 2>/dev/null | tail -1 > $OUT/bench_$v.json
  python - <<PY
import json
d=json.loads(open('$OUT/bench_$v.json').read())
print('bench w3trim=$v', d['value'], d['ms_per_step'], d['roofline']['frac'], d.get('power_cap'))
PY
done
