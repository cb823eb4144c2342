#!/bin/bash
# round 3, GPU call 6: 512 x 128 kw-triple tile (igemm3t): correctness, A/B, step; new transposed-conv modules
set -u
OUT=gpurun_out/r3e; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_transpose.py tests/test_gpu_properties.py -q -m gpu -k "t512 or transpose or upsample or causal_pad or linear or forward_dgrad" 2>&1 | tail -30 > $OUT/pytest.log
tail -4 $OUT/pytest.log
for t in 1 0 1 0; do
  for f in "res 128->128 k3 @16x64x64"; do
    GENIE_TRI_T512=$t MB_NO_WGRAD=1 MB_BATCH=64 MB_FILTER="$f" timeout 300 python scripts/microbench.py conv --iters 10 --out $OUT/mb_t${t}.json 2>&1 | grep -E "fwd|dgrad" | sed "s/^/t512=$t /" >> $OUT/mb.log
  done
done
cut -c1-200 $OUT/mb.log
for t in 1 0; do
  GENIE_TRI_T512=$t timeout 600 python bench.py --no-cpu-baseline --steps 6 --warmup 2 > $OUT/bench_t${t}.json 2> $OUT/bench_t${t}.err
  python - <<PY
import json
d = json.loads(open('$OUT/bench_t${t}.json').read().strip().splitlines()[-1])
print('t512=$t', d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'], {k: (v['ms_per_step'], v['tflops']) for k, v in d['conv_kernels'].items()})
PY
done
