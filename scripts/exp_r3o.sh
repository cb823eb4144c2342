#!/bin/bash
# round 3, GPU call 17: hipGraph capture of the training step -- tests, then eager vs replay at small batches
set -u
OUT=gpurun_out/r3o; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_graph.py tests/test_gpu_trainer.py -q -m gpu -x 2>&1 | tail -25 > $OUT/pytest.log
tail -25 $OUT/pytest.log
timeout 900 python scripts/bench_models.py smallbatch 2>&1 | tail -12 | cut -c1-400 > $OUT/smallbatch.log
cat $OUT/smallbatch.log
