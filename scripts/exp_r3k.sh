#!/bin/bash
# round 3, GPU call 13: fragment reads in first-use order, interleaved with the MFMAs of the k-step (GENIE_TRI_VAR=1) in igemm3w (256 x 256)
# and igemm3h (256 x 128, k32) vs the compiler's order (the pre-read of the next half-tile was sunk behind the MFMAs, in front of the barrier)
set -u
OUT=gpurun_out/r3k; mkdir -p $OUT
GENIE_TRI_VAR=1 timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "triple_wide or triple_k32" 2>&1 | tail -3 > $OUT/pytest_var1.log
tail -2 $OUT/pytest_var1.log
export MB_BATCH=64 MB_NO_WGRAD=1
for rep in 1 2; do
for v in 0 1; do
  for f in "res 256->256 k3 @16x32x32" "res 128->128 k3 @16x64x64"; do
  GENIE_TRI_VAR=$v MB_FILTER="$f" timeout 300 python scripts/microbench.py conv --iters 20 --out $OUT/mb_v${v}_$rep.json 2>&1 | grep -E "fwd|dgrad" | sed "s/^/var=$v /" | cut -c1-200
  done
done
done
for v in 0 1 0 1; do
  GENIE_TRI_VAR=$v timeout 600 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-in-order-pass 2>/dev/null | tail -1 > $OUT/bench_v$v.json
  python - <<PY
import json
d=json.loads(open('$OUT/bench_v$v.json').read())
print('bench var=$v', d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['frac'], {k:(v['ms_per_step'],v['tflops']) for k,v in d['conv_kernels'].items() if v['ms_per_step']>5})
PY
done
