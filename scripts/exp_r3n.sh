#!/bin/bash
# round 3, GPU call 16: zero-frame skipping A/B (GENIE_TRI_TRIM=0/1) on one box
set -u
OUT=gpurun_out/r3n; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "triple or conv_forward or conv_dgrad" 2>&1 | tail -2
export MB_BATCH=64 MB_NO_WGRAD=1
for rep in 1 2; do
for v in 0 1; do
for f in "res 256->256 k3 @16x32x32" "res 128->128 k3 @16x64x64" "res 256->256 k3 @8x16x16"; do
  GENIE_TRI_TRIM=$v MB_FILTER="$f" timeout 300 python scripts/microbench.py conv --iters 20 --out $OUT/mb_$v.json 2>&1 | grep -E "fwd|dgrad" | sed "s/^/trim=$v /" | cut -c1-200
done
done
done
for v in 0 1 0 1; do
  GENIE_TRI_TRIM=$v timeout 600 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-in-order-pass 2>/dev/null | tail -1 > $OUT/bench_$v.json
  python - <<PY
import json
d=json.loads(open('$OUT/bench_$v.json').read())
print('bench trim=$v', d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['frac'], {k:(v['ms_per_step'],v['tflops']) for k,v in d['conv_kernels'].items() if v['ms_per_step']>5})
PY
done
