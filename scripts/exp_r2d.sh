#!/bin/bash
# round-2 experiment batch D: wgrad side stream, clips-per-GPU sweep
set -u
OUT=gpurun_out/r2d; mkdir -p $OUT
python -m pytest tests/test_gpu_gan.py tests/test_gpu_trainer.py -m gpu -q --no-header -rf --timeout 900 -p no:cacheprovider > $OUT/pytest.log 2>&1
grep -E "^(FAILED|ERROR)|passed|failed" $OUT/pytest.log | head -20
run() { # name, args...
  local name=$1; shift
  python bench.py --no-cpu-baseline --no-kernel-events "$@" > $OUT/bench_$name.log 2>&1
  python - <<PY
import json
l=[x for x in open('$OUT/bench_$name.log') if x.startswith('{')]
d=json.loads(l[-1]) if l else None; print('bench $name:', d and (d['ms_per_step'], d['value']))
PY
}
run b8_sync --steps 10 --warmup 3 --batch 8 --async-wgrad 0
run b8_async --steps 10 --warmup 3 --batch 8 --async-wgrad 1
run b16_async --steps 6 --warmup 2 --batch 16 --async-wgrad 1
run b24_async --steps 5 --warmup 2 --batch 24 --async-wgrad 1
run b32_async --steps 4 --warmup 2 --batch 32 --async-wgrad 1
run b32_sync --steps 4 --warmup 2 --batch 32 --async-wgrad 0
