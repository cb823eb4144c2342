#!/bin/bash
# One GPU-box session (via gpurun): parity tests, headline bench (+ per-layer dump), kernel microbenchmarks.
#   gpurun --timeout 1500 -- 'bash scripts/gpu_session.sh r01b'
set -u
TAG=${1:-r01}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
export PYTHONUNBUFFERED=1
echo "== pytest -m gpu" | tee $OUT/progress.log
( time timeout 900 python -m pytest tests -m gpu -x -q --durations=15 ) > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?" | tee -a $OUT/progress.log
tail -5 $OUT/pytest_gpu.log
echo "== bench" | tee -a $OUT/progress.log
timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --dump $OUT/conv_by_label.json > $OUT/bench.json 2> $OUT/bench.err
echo "bench exit $?" | tee -a $OUT/progress.log
cat $OUT/bench.json
echo "== microbench" | tee -a $OUT/progress.log
timeout 600 python scripts/microbench.py ${MB_SECTIONS:-attn hbm conv} --out $OUT/microbench.json > $OUT/microbench.log 2>&1
echo "microbench exit $?" | tee -a $OUT/progress.log
tail -80 $OUT/microbench.log
