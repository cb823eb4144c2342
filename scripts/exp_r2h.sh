#!/bin/bash
# round-2 batch H: 256 x 256 kw-triple tile (igemm3w_kernel) -- tests, per-layer microbench, bench A/B
set -u
OUT=gpurun_out/r2h; mkdir -p $OUT
python -m pytest tests/test_gpu_kernels.py -k "wide" -m gpu -q --no-header -rf --timeout 900 -p no:cacheprovider > $OUT/pytest.log 2>&1
grep -E "^(FAILED|ERROR)|passed|failed" $OUT/pytest.log | head; grep -n "^E  " $OUT/pytest.log | head -10
for WIDE in 0 1; do
  GENIE_TRI_WIDE=$WIDE python scripts/microbench.py conv --iters 10 --out $OUT/mb_conv_w$WIDE.json > /dev/null 2>&1
done
python - <<PY
import json
a={x['name']:x for x in json.load(open('$OUT/mb_conv_w0.json'))['results'] if 'name' in x}
b={x['name']:x for x in json.load(open('$OUT/mb_conv_w1.json'))['results'] if 'name' in x}
for k in a:
    if k in b and 'tflops' in a[k]: print(f"{k[:58]:58s} {a[k]['ms']:8.4f} ms {a[k]['tflops']:7.1f} TF | wide {b[k]['ms']:8.4f} ms {b[k]['tflops']:7.1f} TF")
PY
run() { local name=$1; shift
  env "$@" python bench.py --no-cpu-baseline --steps 5 --warmup 2 > $OUT/bench_$name.log 2>&1
  python - <<PY
import json
l=[x for x in open('$OUT/bench_$name.log') if x.startswith('{')]
d=json.loads(l[-1]) if l else None
print('bench $name:', d and (d['ms_per_step'], d['value'], d.get('roofline',{}).get('kernel'), d.get('roofline',{}).get('frac'), {k:(v['ms_per_step'],v['tflops']) for k,v in d['conv_kernels'].items()}))
PY
}
run a0_w0 GENIE_ASYNC_WGRAD=0 GENIE_TRI_WIDE=0
run a0_w1 GENIE_ASYNC_WGRAD=0 GENIE_TRI_WIDE=1
run a2_w1 GENIE_ASYNC_WGRAD=2 GENIE_TRI_WIDE=1
