#!/bin/bash
# round 3, final evidence run: full -m gpu suite, default bench line, rocprofv3 profile of the bench, the other BASELINE configs, micro-benchmarks
set -u
OUT=gpurun_out/r3final; mkdir -p $OUT
rm -f gpurun_out/parity_report.jsonl
timeout 1500 python -m pytest tests -q -m gpu --durations=8 2>&1 | tail -25 > $OUT/pytest_full.log
grep -E "passed|failed" $OUT/pytest_full.log | tail -2
timeout 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
tail -c 1500 $OUT/bench_default.json | head -c 1500; echo
bash scripts/profile_bench.sh r03 64 > $OUT/prof.log 2>&1
tail -3 $OUT/prof.log
timeout 900 python scripts/bench_models.py lam dyn repr genie4 --cpu-baseline > $OUT/bench_models.log 2>&1
grep -c '^{' $OUT/bench_models.log
timeout 600 python scripts/bench_models.py smallbatch > $OUT/smallbatch.log 2>&1
grep '^{' $OUT/smallbatch.log | cut -c1-330
bash scripts/profile_models.sh r03 > $OUT/prof_models.log 2>&1
tail -3 $OUT/prof_models.log
timeout 600 python scripts/microbench.py attn hbm conv --iters 20 --out $OUT/microbench.json > $OUT/microbench.log 2>&1
MB_BATCH=64 timeout 600 python scripts/microbench.py hbm conv --iters 10 --out $OUT/microbench_b64.json > $OUT/microbench_b64.log 2>&1
tail -3 $OUT/microbench_b64.log | cut -c1-200
