#!/bin/bash
# per-kernel durations of the GroupNorm passes under rocprofv3, per (sweep order, tiny-block) setting: "K:T" pairs in CONFIGS
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for cfg in ${CONFIGS:-0:0 8:0 8:1 4:1}; do
  k=${cfg%%:*}; t=${cfg##*:}
  d=$R/gpurun_out/gn_prof_${k}_$t; rm -rf $d
  GENIE_GN_SWEEP=$k GENIE_GN_TINY=$t MB_BATCH=64 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $d -o p -- python $R/scripts/microbench.py hbm --out $d/mb.json > $d.log 2>&1
  f=$(find $d -name "*kernel_trace.csv" | head -1)
  python - "$f" "$cfg" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.OrderedDict()
for r in rows:
    n = r['Kernel_Name']
    if 'gn_' not in n or 'finalize' in n: continue
    key = (n.split('(')[0][:40], r['Grid_Size_X'], r['Grid_Size_Y'])
    agg.setdefault(key, []).append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, v in agg.items():
    v = sorted(v)
    if v[len(v)//2] > 50: print(f"cfg {sys.argv[2]} {k[0]:40s} grid {k[1]:>8s} x {k[2]:>3s} n={len(v):3d} median {v[len(v)//2]:8.1f} us  min {v[0]:8.1f}")
PY
done
