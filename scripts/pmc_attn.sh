#!/bin/bash
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/pmc_attn; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_SCA" \
           "GRBM_GUI_ACTIVE SQ_WAVES"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -o p$i -- python $ROOT/scripts/exp_attn_one.py > $OUT/p$i.log 2>&1
done
cd $ROOT
python - <<'PY'
import csv, glob, collections
out='gpurun_out/pmc_attn'
agg=collections.defaultdict(lambda: collections.defaultdict(lambda:[0,0.0]))
for f in glob.glob(out+'/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'][:44]
        if 'attn' not in k: continue
        a=agg[k][r['Counter_Name']]; a[0]+=1; a[1]+=float(r['Counter_Value'])
for k,v in agg.items():
    print('==',k)
    for c,(n,s) in sorted(v.items()):
        print(f'   {c:28s} n={n:4d} mean={s/n:14.1f}')
PY
