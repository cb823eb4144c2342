#!/bin/bash
# round 3: low-resolution 512-channel layers at 64 clips (M = 16384 rows): 128-row tiles (igemm3_kernel<128>, 512 blocks) vs 256-row tiles (one block per CU)
set -u
OUT=gpurun_out/r3q; mkdir -p $OUT
export MB_BATCH=64 MB_NO_WGRAD=1
for v in 512 256 512 256; do
  GENIE_TRI_BM256_MIN=$v MB_FILTER="res 512->512 k3 @4x8x8" timeout 300 python scripts/microbench.py conv --iters 30 --out $OUT/mb_$v.json 2>&1 | grep -E "fwd|dgrad" | sed "s/^/bm256_min=$v /" | cut -c1-200
done
GENIE_TRI_BM256_MIN=256 timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "triple" 2>&1 | tail -2
for v in 512 256; do
  GENIE_TRI_BM256_MIN=$v timeout 600 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-in-order-pass 2>/dev/null | tail -1 > $OUT/bench_$v.json
  python - <<PY
import json
d=json.loads(open('$OUT/bench_$v.json').read())
print('bench bm256_min=$v', d['value'], d['ms_per_step'], {k:(v['ms_per_step'],v['tflops']) for k,v in d['conv_kernels'].items() if v['ms_per_step']>1})
PY
done
