#!/bin/bash
set -u
TAG=${1:-w3}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "wgrad" > $OUT/pytest.log 2>&1
echo "pytest exit $?"; tail -15 $OUT/pytest.log
for w in 0 1; do
  echo "== GENIE_TRI_WGRAD=$w"
  GENIE_TRI_WGRAD=$w MB_FILTER="${MB_FILTER:-}" timeout 300 python scripts/microbench.py conv --iters 20 --out $OUT/mb_w$w.json 2>&1 | grep wgrad | python -c "
import sys, json
for l in sys.stdin:
    try: r = json.loads(l)
    except Exception: print(l.rstrip()); continue
    print(f\"{r['name']:44s} {r['ms']:8.4f} ms {r.get('tflops',0):8.1f} TF  {r.get('kernel')}\")
"
done
