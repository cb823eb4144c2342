#!/bin/bash
# round 3, GPU call 14: is the 256 x 256 kernel held back by the power cap?  Same launch on all-zero operands (MB_ZERO=1) and the effective
# shader clock (GRBM_GUI_ACTIVE / wall time) of both; then the GroupNorm-in-epilogue switches on the current build.
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r3l; mkdir -p $OUT
export MB_BATCH=64 MB_NO_WGRAD=1
for z in 0 1; do
  for f in "res 256->256 k3 @16x32x32" "res 128->128 k3 @16x64x64"; do
    MB_ZERO=$z MB_FILTER="$f" timeout 300 python scripts/microbench.py conv --iters 20 --out $OUT/mb_z${z}.json 2>&1 | grep -E "fwd|dgrad" | sed "s/^/zero=$z /" | cut -c1-200
  done
done
cd /tmp && export TMPDIR=/tmp
for z in 0 1; do
  MB_ZERO=$z MB_FILTER="res 256->256 k3 @16x32x32" timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $OUT/pmc_z$z -o p -- python $ROOT/scripts/microbench.py conv --iters 5 --out $OUT/mb_pmc_z$z.json > $OUT/pmc_z$z.log 2>&1
done
cd $ROOT
python - <<'PY'
import csv, glob, collections
for z in (0, 1):
    trace = {}
    for f in glob.glob(f'gpurun_out/r3l/pmc_z{z}/**/*kernel_trace.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            trace[r['Dispatch_Id']] = (r['Kernel_Name'], int(r['End_Timestamp']) - int(r['Start_Timestamp']))
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f'gpurun_out/r3l/pmc_z{z}/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if 'igemm3w' not in r['Kernel_Name']: continue
            agg[r['Dispatch_Id']][r['Counter_Name']].append(float(r['Counter_Value']))
    rows = []
    for d, c in agg.items():
        if d not in trace: continue
        ns = trace[d][1]
        gui = sum(c.get('GRBM_GUI_ACTIVE', [0]))
        rows.append((ns, gui, sum(c.get('SQ_BUSY_CYCLES', [0])), sum(c.get('SQ_VALU_MFMA_BUSY_CYCLES', [0]))))
    rows = rows[2:]
    if rows:
        ns = sum(r[0] for r in rows) / len(rows); gui = sum(r[1] for r in rows) / len(rows)
        print(f'zero={z}: {len(rows)} igemm3w launches, avg {ns/1e3:.1f} us, GRBM_GUI_ACTIVE {gui:.3e} -> {gui/ns:.3f} GHz (if one instance), MFMA busy cycles {sum(r[3] for r in rows)/len(rows):.3e}, SQ busy {sum(r[2] for r in rows)/len(rows):.3e}')
PY
for g in 0 1 2; do
  GENIE_GN_FUSE=$g timeout 600 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-in-order-pass 2>/dev/null | tail -1 > $OUT/bench_gn$g.json
  python - <<PY
import json
d=json.loads(open('$OUT/bench_gn$g.json').read())
print('bench GN_FUSE=$g', d['value'], d['ms_per_step'], d['roofline']['frac'])
PY
done
