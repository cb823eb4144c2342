#!/bin/bash
# round 3, GPU call 15: zero-frame skipping in the kw-triple forward / backward-data kernels (step table sorted by dt, rows of padding frames
# trimmed per row tile): conv tests, model parity tests, microbench and bench
set -u
OUT=gpurun_out/r3m; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_transpose.py -q -m gpu -x 2>&1 | tail -3 > $OUT/pytest_kernels.log
tail -2 $OUT/pytest_kernels.log
timeout 900 python -m pytest tests/test_gpu_tokenizer.py -q -m gpu -x 2>&1 | tail -3 > $OUT/pytest_tok.log
tail -2 $OUT/pytest_tok.log
export MB_BATCH=64 MB_NO_WGRAD=1
for f in "res 256->256 k3 @16x32x32" "res 128->128 k3 @16x64x64" "res 256->256 k3 @8x16x16"; do
  MB_FILTER="$f" timeout 300 python scripts/microbench.py conv --iters 20 --out $OUT/mb.json 2>&1 | grep -E "fwd|dgrad" | cut -c1-200
done
for v in 0 1; do
  timeout 600 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-in-order-pass 2>/dev/null | tail -1 > $OUT/bench_$v.json
  python - <<PY
import json
d=json.loads(open('$OUT/bench_$v.json').read())
print('bench', d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['frac'], {k:(v['ms_per_step'],v['tflops']) for k,v in d['conv_kernels'].items() if v['ms_per_step']>5})
PY
done
