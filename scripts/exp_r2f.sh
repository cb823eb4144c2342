#!/bin/bash
# round-2 batch F: gated vs unordered wgrad side stream, XCD-column tile mapping of the kw-triple kernel
set -u
OUT=gpurun_out/r2f; mkdir -p $OUT
python -m pytest tests/test_gpu_kernels.py -k "triple" tests/test_gpu_trainer.py -m gpu -q --no-header -rf --timeout 900 -p no:cacheprovider > $OUT/pytest.log 2>&1
GENIE_TRI_XCDCOL=1 python -m pytest tests/test_gpu_kernels.py -k "triple" -m gpu -q --no-header -rf --timeout 900 -p no:cacheprovider > $OUT/pytest_xcdcol.log 2>&1
grep -E "^(FAILED|ERROR)|passed|failed" $OUT/pytest.log $OUT/pytest_xcdcol.log | head -20
run() { # name, env..., -- args
  local name=$1; shift
  env "$@" python bench.py --no-cpu-baseline --steps 5 --warmup 2 > $OUT/bench_$name.log 2>&1
  python - <<PY
import json
l=[x for x in open('$OUT/bench_$name.log') if x.startswith('{')]
d=json.loads(l[-1]) if l else None
print('bench $name:', d and (d['ms_per_step'], d['value'], d.get('roofline',{}).get('frac'), d.get('roofline',{}).get('avg_launch_ms')))
PY
}
run async0 GENIE_ASYNC_WGRAD=0
run async1 GENIE_ASYNC_WGRAD=1
run async2 GENIE_ASYNC_WGRAD=2
run async1_xcdcol GENIE_ASYNC_WGRAD=1 GENIE_TRI_XCDCOL=1
run async0_xcdcol GENIE_ASYNC_WGRAD=0 GENIE_TRI_XCDCOL=1
