#!/bin/bash
# round 3, GPU call 4: pipelined attention forward (correctness + A/B), non-temporal stores in the stem conv (A/B)
set -u
OUT=gpurun_out/r3c; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_attention.py tests/test_gpu_golden.py -q -m gpu 2>&1 | tail -8 > $OUT/pytest_attn.log
tail -3 $OUT/pytest_attn.log
for pipe in 1 0; do
  GENIE_ATTN_PIPE=$pipe timeout 300 python scripts/microbench.py attn --iters 20 --out $OUT/mb_attn_pipe${pipe}.json 2>&1 | grep -E "attention" | sed "s/^/pipe=$pipe /" >> $OUT/attn.log
done
cat $OUT/attn.log | cut -c1-260
for nt in 0 1; do
  GENIE_NARROW_NT=$nt timeout 300 python scripts/microbench.py hbm --iters 20 --out $OUT/mb_hbm_nt${nt}.json 2>&1 | grep -E "CausalConv3d" | sed "s/^/nt=$nt /" >> $OUT/hbm.log
done
cat $OUT/hbm.log | cut -c1-260
