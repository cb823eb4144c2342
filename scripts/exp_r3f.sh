#!/bin/bash
# round 3, GPU call 6: 512 x 128 kw-triple tile (igemm3t): correctness, A/B, step; new transposed-conv modules
set -u
OUT=gpurun_out/r3f; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_trainer.py tests/test_gpu_properties.py -q -m gpu -k "k32 or linear or forward_dgrad or loopback" 2>&1 | tail -30 > $OUT/pytest.log
tail -4 $OUT/pytest.log
for t in 1 0 1 0; do
  for f in "res 128->128 k3 @16x64x64"; do
    GENIE_TRI_H=$t MB_NO_WGRAD=1 MB_BATCH=64 MB_FILTER="$f" timeout 300 python scripts/microbench.py conv --iters 10 --out $OUT/mb_t${t}.json 2>&1 | grep -E "fwd|dgrad" | sed "s/^/k32=$t /" >> $OUT/mb.log
  done
done
cut -c1-200 $OUT/mb.log
for t in 1 0; do
  GENIE_TRI_H=$t timeout 600 python bench.py --no-cpu-baseline --steps 6 --warmup 2 > $OUT/bench_t${t}.json 2> $OUT/bench_t${t}.err
  python - <<PY
import json
d = json.loads(open('$OUT/bench_t${t}.json').read().strip().splitlines()[-1])
print('k32=$t', d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'], {k: (v['ms_per_step'], v['tflops']) for k, v in d['conv_kernels'].items()})
PY
done
timeout 300 python bench.py --dp-loopback --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-events > $OUT/bench_loopback.json 2> $OUT/bench_loopback.err; python - <<PY
import json
d = json.loads(open('$OUT/bench_loopback.json').read().strip().splitlines()[-1])
print('loopback', d['value'], d['ms_per_step'], json.dumps(d.get('comm'))[:900])
PY
