#!/bin/bash
# GroupNorm passes one at a time under rocprofv3 (median / min duration per launch), for GENIE_GN_SWEEP = 0 (contiguous range per block, rounds 1-3)
# and 1 (the shipped sweeps): gpurun -- 'bash scripts/exp_gn_prof.sh'.  profiles/r04_groupnorm_sweep_ab.log was taken with the development form of this
# script (K and one-chunk-per-block as separate switches: commit "GroupNorm passes: chunk-interleaved ..." holds the result, its parent the switches).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for k in ${SWEEPS:-0 1}; do
  d=$R/gpurun_out/gn_prof_$k; rm -rf $d
  GENIE_GN_SWEEP=$k MB_BATCH=64 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $d -o p -- python $R/scripts/microbench.py hbm --out $d/mb.json > $d.log 2>&1
  f=$(find $d -name "*kernel_trace.csv" | head -1)
  python - "$f" "$k" <<'PY'
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
    if v[len(v)//2] > 50: print(f"sweep {sys.argv[2]} {k[0]:40s} grid {k[1]:>8s} x {k[2]:>3s} n={len(v):3d} median {v[len(v)//2]:8.1f} us  min {v[0]:8.1f}")
PY
done
