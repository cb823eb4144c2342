#!/bin/bash
# round 3: one-pass GroupNorm forward, GENIE_GN_FUSED = 0 (two-pass) / 1 (two register sets) / 2 (one set, two clips in flight de-phased)
set -u
OUT=gpurun_out/r3t; mkdir -p $OUT
for v in 0 2; do
  for b in 8 64; do
    GENIE_GN_FUSED=$v MB_BATCH=$b timeout 300 python scripts/microbench.py hbm --iters 20 --out $OUT/mb_${v}_$b.json 2>&1 | grep -E "GroupNorm\+SiLU fwd C=(128|256).*G=1" | sed "s/^/fused=$v /" | cut -c1-170
  done
done
