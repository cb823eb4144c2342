#!/bin/bash
# round 3, GPU call 2: lean wgrad3 correctness + A/B timing at the bench's batch, the repaired parity tests
set -u
OUT=gpurun_out/r3a; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "wgrad" 2>&1 | tail -15 > $OUT/pytest_wgrad.log
for lean in 1 0; do
  for f in "res 128->128 k3 @16x64x64" "res 256->256 k3 @16x32x32" "res 512->512 k3 @4x8x8" "res 256->256 k3 @8x16x16"; do
    GENIE_W3_LEAN=$lean MB_BATCH=64 MB_FILTER="$f" timeout 300 python scripts/microbench.py conv --iters 10 --out $OUT/mb_lean${lean}.json 2>&1 | grep wgrad | sed "s/^/lean=$lean /" >> $OUT/mb.log
  done
done
cat $OUT/mb.log
timeout 1200 python -m pytest tests/test_gpu_properties.py tests/test_gpu_genie.py::test_genie_compute_loss_matches_its_parts_and_oracle tests/test_gpu_tokenizer.py::test_magvit2_full_training_step_parity tests/test_gpu_tokenizer.py::test_tokenizer_training_step_parity -q -m gpu -s 2>&1 | tail -60 > $OUT/pytest_parity.log
tail -5 $OUT/pytest_wgrad.log $OUT/pytest_parity.log
