#!/bin/bash
# triple-kernel bring-up: parity tests, then A/B timing of the conv shapes under the kernel-selection knobs
set -u
TAG=${1:-tri}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "triple or conv_forward or residual or shuffle" > $OUT/pytest.log 2>&1
echo "pytest exit $?"; tail -15 $OUT/pytest.log
IFS=";" read -ra CFG_LIST <<< "${CFGS:--1 0;128 0;256 0;256 2;0 0}"
for cfg in "${CFG_LIST[@]}"; do
  set -- $cfg
  echo "== GENIE_TRI=$1 GENIE_TRI_FLAGS=$2"
  GENIE_TRI=$1 GENIE_TRI_FLAGS=$2 MB_FILTER="${MB_FILTER:-res }" MB_NO_WGRAD=1 timeout 300 python scripts/microbench.py conv --iters 20 --out $OUT/mb_tri$1_f$2.json 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    try: r = json.loads(l)
    except Exception: print(l.rstrip()); continue
    print(f\"{r['name']:44s} {r['ms']:8.4f} ms {r.get('tflops',0):8.1f} TF  {r.get('kernel')}\")
"
done
