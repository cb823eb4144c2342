#!/bin/bash
# round 3: one-pass GroupNorm forward (GENIE_GN_FUSED=0/1): tests, then kernel-level and step-level A/B
set -u
OUT=gpurun_out/r3s; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "groupnorm" 2>&1 | tail -4 | cut -c1-250
for v in 0 1; do
  for b in 8 64; do
    GENIE_GN_FUSED=$v MB_BATCH=$b timeout 300 python scripts/microbench.py hbm --iters 20 --out $OUT/mb_${v}_$b.json 2>&1 | grep -E "GroupNorm\+SiLU fwd" | sed "s/^/fused=$v /" | cut -c1-170
  done
done
for v in 0 1 0 1; do
  GENIE_GN_FUSED=$v GENIE_BENCH_NO_PROBE=1 timeout 600 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-in-order-pass 2>/dev/null | tail -1 > $OUT/bench_$v.json
  python - <<PY
import json
d=json.loads(open('$OUT/bench_$v.json').read())
print('bench gn_fused=$v', d['value'], d['ms_per_step'], d['roofline']['frac'], d['config']['final_loss'])
PY
done
