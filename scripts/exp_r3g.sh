#!/bin/bash
# round 3, GPU call 8: profile of the default bench (kernel stats + FETCH / WRITE passes) and the other BASELINE configs
set -u
bash scripts/profile_bench.sh r03 64 > gpurun_out/prof_r03.log 2>&1
tail -5 gpurun_out/prof_r03.log
timeout 900 python scripts/bench_models.py lam dyn genie4 --cpu-baseline > gpurun_out/bench_models_r03.log 2>&1
cut -c1-700 gpurun_out/bench_models_r03.log | tail -6
bash scripts/profile_models.sh r03 > gpurun_out/prof_models_r03.log 2>&1
tail -40 gpurun_out/prof_models_r03.log
