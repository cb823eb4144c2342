#!/bin/bash
# Step-table order inside a dt of the kw-triple conv kernels: (dh, channel block) [GENIE_TRI_DH_INNER=0, rounds 2-4] vs (channel block, dh) [=1]:
# kernel time and FETCH_SIZE / WRITE_SIZE of the two dominant layer shapes at 64 clips.  -> gpurun_out/ab_tri_order/summary.txt
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/ab_tri_order; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for shape in "256 256 16,32,32" "128 128 16,64,64"; do
  set -- $shape
  for v in 0 1 0 1; do
    GENIE_TRI_DH_INNER=$v python $ROOT/scripts/conv_once.py --cin $1 --cout $2 --size $3 --batch 64 --iters 8 --dgrad 1 2>/dev/null | grep '^{' >> $OUT/times.jsonl
  done
  for v in 0 1; do
    for c in FETCH_SIZE WRITE_SIZE; do
      GENIE_TRI_DH_INNER=$v timeout 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/p_${1}_${v}_$c -o p -- python $ROOT/scripts/conv_once.py --cin $1 --cout $2 --size $3 --batch 64 --iters 2 > /dev/null 2>&1
    done
  done
done
cd $ROOT
python - <<'PY' > $OUT/summary.txt
import csv, glob, json, collections
out='gpurun_out/ab_tri_order'
for l in open(out+'/times.jsonl'): print(l.strip())
for d in sorted(glob.glob(out+'/p_*')):
    agg=collections.defaultdict(lambda:[0,0.0])
    for f in glob.glob(d+'/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if 'igemm3' not in r['Kernel_Name']: continue
            a=agg[(r['Kernel_Name'][:40], r['Counter_Name'])]; a[0]+=1; a[1]+=float(r['Counter_Value'])
    for (k,c),(n,s) in agg.items():
        # FETCH_SIZE / WRITE_SIZE are in KB; gfx950: FETCH_SIZE counts half of a wide coalesced read (MI355X_MICROARCH.md, HBM section) -> x2
        mb = s/n/1024*(2 if c=='FETCH_SIZE' else 1)
        print(d.split('/')[-1], k, c, f'launches={n} MB_per_launch={mb:.1f}' + (' (x2 gfx950 correction applied)' if c=='FETCH_SIZE' else ''))
PY
cat $OUT/summary.txt
