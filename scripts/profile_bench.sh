#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace stats of bench.py, then HBM traffic counters in their own passes.
#   gpurun --timeout 1200 -- 'bash scripts/profile_bench.sh r01 8'
set -u
TAG=${1:-r01}
BATCH=${2:-64}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export GENIE_BENCH_NO_PROBE=1     # bench.py's power-cap probe launches the dominant kernel itself: keep it out of that kernel's averages
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- python $ROOT/bench.py --steps 4 --warmup 2 --batch $BATCH --no-cpu-baseline --no-in-order-pass > $OUT/bench_under_rocprof.json 2> $OUT/stats.log
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o fetch -- python $ROOT/bench.py --steps 1 --warmup 1 --batch $BATCH --no-cpu-baseline --no-kernel-events > /dev/null 2> $OUT/fetch.log
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -o write -- python $ROOT/bench.py --steps 1 --warmup 1 --batch $BATCH --no-cpu-baseline --no-kernel-events > /dev/null 2> $OUT/write.log
cd $ROOT
python scripts/summarize_profile.py $OUT $TAG $BATCH
ls -R $OUT | head -40
