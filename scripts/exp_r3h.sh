#!/bin/bash
# round 3, GPU call 9: stem conv with the next tile's image prefetched: correctness + HBM microbench at 8 and 64 clips
set -u
OUT=gpurun_out/r3h; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "narrow" 2>&1 | tail -4 > $OUT/pytest.log
tail -2 $OUT/pytest.log
for b in 8 64; do
  MB_BATCH=$b timeout 300 python scripts/microbench.py hbm --iters 20 --out $OUT/mb_hbm_b$b.json 2>&1 | grep -E "CausalConv3d|yardstick|GroupNorm\+SiLU (fwd|bwd) C=128 16x64x64 G=1" | cut -c1-230
done
