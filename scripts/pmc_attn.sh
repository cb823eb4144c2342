#!/bin/bash
# SQ counters of the MFMA attention kernels (forward / dQ / dK-dV, general and register-lean family) on the LatentAction shape
# (S = 4096, 4 x 64): separate rocprofv3 passes, --kernel-trace + --pmc only.  Output: gpurun_out/pmc_attn/summary.txt
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/pmc_attn; mkdir -p $OUT
export AB_ONLY="${AB_ONLY:-spatial S=4096}" AB_MODES="${AB_MODES:-0,7}" AB_REPS=1 AB_ITERS=2
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU" \
           "SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_WAVES SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 150 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -o p$i -- python $ROOT/scripts/exp_attn_lean.py > $OUT/p$i.log 2>&1
done
cd $ROOT
python - <<'PY' > gpurun_out/pmc_attn/summary.txt
import csv, glob, collections
out='gpurun_out/pmc_attn'
agg=collections.defaultdict(lambda: collections.defaultdict(lambda:[0,0.0]))
for f in glob.glob(out+'/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'][:44]
        if 'attn_' not in k or 'prep' in k: continue
        a=agg[k][r['Counter_Name']]; a[0]+=1; a[1]+=float(r['Counter_Value'])
dur=collections.defaultdict(lambda:[0,0.0])
for f in glob.glob(out+'/p*/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'][:44]
        if 'attn_' not in k or 'prep' in k: continue
        d=dur[k]; d[0]+=1; d[1]+=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))
for k,v in sorted(agg.items()):
    n,t=dur[k]
    print('==',k, f'launches={n} mean_us={t/max(n,1)/1e3:.1f}')
    c={name:s/cnt for name,(cnt,s) in v.items()}
    for name in sorted(c): print(f'   {name:28s} {c[name]:16.1f}')
    if 'SQ_INSTS_MFMA' in c and c['SQ_INSTS_MFMA']>0:
        print(f"   -> VALU per MFMA {c.get('SQ_INSTS_VALU',0)/c['SQ_INSTS_MFMA']:.2f}; MFMA-busy / busy cycles {c.get('SQ_VALU_MFMA_BUSY_CYCLES',0)/max(c.get('SQ_BUSY_CYCLES',1),1):.3f}")
    if 'GRBM_GUI_ACTIVE' in c and n:
        print(f"   -> effective clock {c['GRBM_GUI_ACTIVE']/(t/n):.3f} GHz")
PY
cat gpurun_out/pmc_attn/summary.txt
