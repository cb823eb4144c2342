#!/bin/bash
# round 3: does zero-frame skipping change the fabric traffic (FETCH_SIZE) of the conv kernels?  Same box, three configurations.
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r3r; mkdir -p $OUT
export GENIE_BENCH_NO_PROBE=1
cd /tmp && export TMPDIR=/tmp
for cfg in "0 0" "1 0" "1 1"; do
  set -- $cfg
  tag=t$1w$2
  GENIE_TRI_TRIM=$1 GENIE_W3_TRIM=$2 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/$tag -o f -- python $ROOT/bench.py --steps 1 --warmup 1 --batch 64 --no-cpu-baseline --no-kernel-events > /dev/null 2> $OUT/$tag.log
done
cd $ROOT
python - <<'PY'
import csv, glob, collections
for tag in ('t0w0', 't1w0', 't1w1'):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for f in glob.glob(f'gpurun_out/r3r/{tag}/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name']
            name = 'igemm3w' if 'igemm3w' in k else 'igemm3h' if 'igemm3h' in k else 'wgrad3l<6>' if 'wgrad3l_kernel<6>' in k else 'wgrad3l<5>' if 'wgrad3l_kernel<5>' in k else None
            if name and r['Counter_Name'] == 'FETCH_SIZE':
                agg[name][0] += 1; agg[name][1] += float(r['Counter_Value'])
    print(tag, {k: (v[0], round(v[1] / v[0] * 1024 * 2 / 1e9, 3)) for k, v in sorted(agg.items())}, '(launches, GB fetched per launch: FETCH_SIZE KB x 2)')
PY
