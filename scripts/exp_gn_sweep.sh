#!/bin/bash
# A/B of the GroupNorm sweep order (GENIE_GN_SWEEP = 0 contiguous range per block / 4 / 8 accesses in flight over interleaved chunks) on one box:
# the HBM-bound section of the microbench at 64 clips per value, then the GroupNorm tests on the default.
mkdir -p gpurun_out/gn_sweep
for k in 0 4 8 0 4 8; do
  GENIE_GN_SWEEP=$k MB_BATCH=64 timeout 300 python scripts/microbench.py hbm --out gpurun_out/gn_sweep/mb_$k.json > gpurun_out/gn_sweep/mb_$k.log 2>&1
  python - $k <<'PY'
import json, sys
k = sys.argv[1]
d = json.load(open(f'gpurun_out/gn_sweep/mb_{k}.json'))
for r in d['results']:
    if 'GroupNorm' in r['name']:
        print(f"sweep {k}: {r['name'][:60]:60s} {r['ms']:.4f} ms {r['gbps']:.0f} GB/s")
PY
done
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -p no:cacheprovider -k "groupnorm or gn_ or norm" 2>&1 | tail -5
