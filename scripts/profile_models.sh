#!/bin/bash
# rocprofv3 kernel tables of one training step of BASELINE configs[2], [3], [4] (scripts/bench_models.py), condensed for profiles/.
#   gpurun --timeout 900 -- 'bash scripts/profile_models.sh r03'
set -u
TAG=${1:-r03}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_models_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for m in lam dyn genie4; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$m -o $m -- python $ROOT/scripts/bench_models.py $m > $OUT/${m}_bench.json 2> $OUT/${m}.log
done
cd $ROOT
python - <<PY
import csv, glob, os
out = '$OUT'
for m in ('lam', 'dyn', 'genie4'):
    hits = glob.glob(os.path.join(out, m, '**', '*kernel_stats.csv'), recursive=True)
    if not hits:
        print(m, 'no kernel stats'); continue
    rows = list(csv.DictReader(open(hits[0])))
    with open(os.path.join(out, f'${TAG}_models_{m}_kernel_stats.csv'), 'w', newline='') as f:
        w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
        w.writeheader()
        for r in rows[:30]:
            r = dict(r); r['Name'] = r['Name'][:140]; w.writerow(r)
    tot = sum(float(r['TotalDurationNs']) for r in rows)
    print(m, 'top kernels:')
    for r in rows[:8]:
        print(f"   {r['Name'][:80]:80s} {float(r['TotalDurationNs']) / tot * 100:5.1f} %  calls {r['Calls']}")
PY
