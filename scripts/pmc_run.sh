#!/bin/bash
# SQ counters + kernel durations of the kernels whose name contains FILTER, for any command: two separate rocprofv3 passes
# (--kernel-trace + --pmc only).   bash scripts/pmc_run.sh <outname> <filter> <command...>   -> gpurun_out/<outname>/summary.txt
# PMC_HBM=1 adds one pass each for FETCH_SIZE and WRITE_SIZE (reported in MB per launch: counters are KB; FETCH x 2 = the gfx950 correction).
set -u
NAME=$1; FILTER=$2; shift 2
ROOT=$(pwd); OUT=$ROOT/gpurun_out/$NAME; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU" \
           "SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_WAVES SQ_WAIT_ANY GRBM_GUI_ACTIVE" \
           ${PMC_HBM:+"FETCH_SIZE" "WRITE_SIZE"}; do
  i=$((i+1))
  (cd $ROOT && timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -o p$i -- "$@" > $OUT/p$i.log 2>&1)
done
cd $ROOT
python - "$OUT" "$FILTER" <<'PY' > $OUT/summary.txt
import csv, glob, collections, sys
out, filt = sys.argv[1], sys.argv[2]
agg=collections.defaultdict(lambda: collections.defaultdict(lambda:[0,0.0]))
for f in glob.glob(out+'/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'][:70]
        if filt not in k: continue
        a=agg[k][r['Counter_Name']]; a[0]+=1; a[1]+=float(r['Counter_Value'])
dur=collections.defaultdict(lambda:[0,0.0])
for f in glob.glob(out+'/p*/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'][:70]
        if filt not in k: continue
        d=dur[k]; d[0]+=1; d[1]+=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))
for k,v in sorted(agg.items()):
    n,t=dur[k]
    print('==',k, f'launches={n} mean_us={t/max(n,1)/1e3:.1f}')
    c={name:s/cnt for name,(cnt,s) in v.items()}
    for name in sorted(c): print(f'   {name:28s} {c[name]:16.1f}')
    if 'GRBM_GUI_ACTIVE' in c and n:
        cyc = c['GRBM_GUI_ACTIVE'] / 8.0
        print(f"   -> kernel cycles {cyc:.0f}; effective clock {cyc / (t / n):.3f} GHz")
        if 'SQ_VALU_MFMA_BUSY_CYCLES' in c: print(f"   -> matrix pipe busy {c['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * cyc):.3f} of the kernel's cycles")
        if 'SQ_ACTIVE_INST_VALU' in c: print(f"   -> VALU active {4 * c['SQ_ACTIVE_INST_VALU'] / (1024 * cyc):.3f} of the kernel's cycles")
        if 'SQ_ACTIVE_INST_LDS' in c: print(f"   -> LDS instruction issue active {4 * c['SQ_ACTIVE_INST_LDS'] / (1024 * cyc):.3f} of the kernel's cycles")
        if 'SQ_WAVE_CYCLES' in c: print(f"   -> resident waves per SIMD, kernel average {4 * c['SQ_WAVE_CYCLES'] / (1024 * cyc):.2f}")
        if 'SQ_WAIT_ANY' in c and 'SQ_WAVE_CYCLES' in c: print(f"   -> waves parked (s_waitcnt / barrier) {c['SQ_WAIT_ANY'] / c['SQ_WAVE_CYCLES']:.3f}, issue-stalled {c.get('SQ_WAIT_INST_ANY', 0) / c['SQ_WAVE_CYCLES']:.3f} of wave cycles")
    if 'FETCH_SIZE' in c: print(f"   -> HBM fetch {c['FETCH_SIZE'] * 1024 * 2 / 1e6:.1f} MB per launch (x2-corrected), write {c.get('WRITE_SIZE', 0) * 1024 / 1e6:.1f} MB")
    if 'SQ_INSTS_MFMA' in c and c['SQ_INSTS_MFMA'] > 0:
        print(f"   -> VALU instructions per MFMA {c.get('SQ_INSTS_VALU', 0) / c['SQ_INSTS_MFMA']:.2f}; LDS instructions per MFMA {c.get('SQ_INSTS_LDS', 0) / c['SQ_INSTS_MFMA']:.2f}; "
              f"SALU per MFMA {c.get('SQ_INSTS_SALU', 0) / c['SQ_INSTS_MFMA']:.2f}; LDS bank-conflict cycles per LDS instruction {c.get('SQ_LDS_BANK_CONFLICT', 0) / max(c.get('SQ_INSTS_LDS', 1), 1):.2f}")
PY
cat $OUT/summary.txt
