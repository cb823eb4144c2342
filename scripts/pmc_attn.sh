#!/bin/bash
# SQ counters of the MFMA attention kernels (forward / dQ / dK-dV, general and register-lean family) on the LatentAction shape
# (S = 4096, 4 x 64): separate rocprofv3 passes, --kernel-trace + --pmc only.  Output: gpurun_out/pmc_attn/summary.txt
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/pmc_attn; mkdir -p $OUT
export AB_ONLY="${AB_ONLY:-spatial S=4096}" AB_MODES="${AB_MODES:-0,23}" AB_REPS=1 AB_ITERS=2
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
    n,t=dur[k]      # launches, total ns
    print('==',k, f'launches={n} mean_us={t/max(n,1)/1e3:.1f}')
    c={name:s/cnt for name,(cnt,s) in v.items()}
    for name in sorted(c): print(f'   {name:28s} {c[name]:16.1f}')
    # derived figures.  GRBM_GUI_ACTIVE is summed over the 8 XCDs; SQ_ACTIVE_INST_* / SQ_WAVE_CYCLES count quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES
    # cycles (32 per v_mfma_f32_32x32x16_bf16), all summed over the chip's 1024 SIMDs (MI355X_MICROARCH.md, per-instruction constants)
    if 'GRBM_GUI_ACTIVE' in c and n:
        cyc = c['GRBM_GUI_ACTIVE'] / 8.0
        print(f"   -> kernel cycles {cyc:.0f}; effective clock {cyc / (t / n):.3f} GHz")
        if 'SQ_VALU_MFMA_BUSY_CYCLES' in c:
            print(f"   -> matrix pipe busy {c['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * cyc):.3f} of the kernel's cycles")
        if 'SQ_ACTIVE_INST_VALU' in c:
            print(f"   -> VALU active {4 * c['SQ_ACTIVE_INST_VALU'] / (1024 * cyc):.3f} of the kernel's cycles")
        if 'SQ_WAVE_CYCLES' in c:
            print(f"   -> resident waves per SIMD, kernel average {4 * c['SQ_WAVE_CYCLES'] / (1024 * cyc):.2f}")
    if 'SQ_INSTS_MFMA' in c and c['SQ_INSTS_MFMA'] > 0:
        print(f"   -> VALU instructions per MFMA {c.get('SQ_INSTS_VALU', 0) / c['SQ_INSTS_MFMA']:.2f}; LDS instructions per MFMA {c.get('SQ_INSTS_LDS', 0) / c['SQ_INSTS_MFMA']:.2f}; "
              f"SALU per MFMA {c.get('SQ_INSTS_SALU', 0) / c['SQ_INSTS_MFMA']:.2f}")
PY
cat gpurun_out/pmc_attn/summary.txt
