#!/bin/bash
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/models; mkdir -p $OUT
timeout 600 python scripts/bench_models.py "$@" 2>&1 | tail -12
cd /tmp && export TMPDIR=/tmp
for m in "$@"; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$m -o $m -- python $ROOT/scripts/bench_models.py $m > $OUT/$m.log 2>&1
  f=$(find $OUT/$m -name "*kernel_stats.csv" | head -1)
  echo "== $m top kernels"; python - "$f" <<'PY'
import csv, sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:14]:
    print(f"{float(r['TotalDurationNs'])/tot*100:5.1f}%  {int(r['Calls']):6d} calls {float(r['AverageNs'])/1e3:9.1f} us  {r['Name'][:80]}")
PY
done
