#!/bin/bash
# round 3, GPU call 5: lean (buffer-addressed) staging in the forward / backward-data kw-triple kernels: correctness, A/B, step
set -u
OUT=gpurun_out/r3d; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_properties.py tests/test_gpu_transpose.py -q -m gpu -k "conv or linear or upsample or causal" 2>&1 | tail -30 > $OUT/pytest_conv.log
tail -3 $OUT/pytest_conv.log
for lean in 1 0 1 0; do
  for f in "res 128->128 k3 @16x64x64" "res 256->256 k3 @16x32x32" "res 128->256 k3 @16x32x32"; do
    GENIE_TRI_LEAN=$lean MB_NO_WGRAD=1 MB_BATCH=64 MB_FILTER="$f" timeout 300 python scripts/microbench.py conv --iters 10 --out $OUT/mb_lean${lean}.json 2>&1 | grep -E "fwd|dgrad" | sed "s/^/lean=$lean /" >> $OUT/mb.log
  done
done
cut -c1-220 $OUT/mb.log
for lean in 1 0; do
  GENIE_TRI_LEAN=$lean timeout 600 python bench.py --no-cpu-baseline --steps 6 --warmup 2 > $OUT/bench_lean${lean}.json 2> $OUT/bench_lean${lean}.err
  python - <<PY
import json
d = json.loads(open('$OUT/bench_lean${lean}.json').read().strip().splitlines()[-1])
print('lean=$lean', d['value'], d['ms_per_step'], d['roofline']['frac'], {k: (v['ms_per_step'], v['tflops']) for k, v in d['conv_kernels'].items()})
PY
done
