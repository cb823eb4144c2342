#!/bin/bash
# Evidence of one round on one MI355X box (run through gpurun), every stage writing files named for profiles/ under gpurun_out/evidence_<tag>/:
#
#   gpurun --timeout 1500 -- 'bash scripts/evidence.sh r04 tests bench profile'
#   gpurun --timeout 1200 -- 'bash scripts/evidence.sh r04 micro models pmc'
#
# stages
#   tests    full `pytest -m gpu` (no -x)                                   -> <tag>_pytest.log, <tag>_parity_report.jsonl
#   bench    the default `python bench.py` line                             -> <tag>_bench_default.json
#   profile  rocprofv3 --kernel-trace --stats of bench.py + FETCH_SIZE / WRITE_SIZE passes (scripts/profile_bench.sh)
#                                                                          -> <tag>_kernel_stats.csv, <tag>_summary.json, <tag>_bench_under_rocprof.json
#   micro    scripts/microbench.py at 8 and at 64 clips                     -> <tag>_microbench.json, <tag>_microbench_b64.json
#   models   scripts/bench_models.py (BASELINE configs[2], [3], [4], REPR tokenizer) + their rocprofv3 kernel tables
#                                                                          -> <tag>_bench_models.json, <tag>_models_{lam,dyn,genie4}_kernel_stats.csv
#   variants the bench on the same box with (a) the RCCL bucket all-reduces on a single-rank group, fp32 and bf16 payload, (b) the weight
#            gradients on a side stream -- the side measurements DESIGN.md quotes next to the default line
#                                                                          -> <tag>_bench_variants.jsonl
#   guard    kernel-level suite under guard-page allocations, two file orders + the deliberate-overrun regression -> <tag>_guard_*.log
#   multigpu bench.py --gpus 2 (both all-reduce algorithms, bf16 payload) + the two-rank RCCL test, where two GPUs exist -> <tag>_bench_gpus2.jsonl
#   pmc      SQ counters of the attention families and of the dominant conv layer (separate rocprofv3 passes)
#                                                                          -> <tag>_pmc_attn.txt, <tag>_pmc_conv.txt
# (Rounds 1-3 used one-off scripts/exp_r*.sh files for the same jobs; they are in the history up to commit 7bfe1bc.)
set -u
TAG=${1:-r04}; shift || true
STAGES=${*:-tests bench profile micro models pmc}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/evidence_$TAG; mkdir -p $OUT
for st in $STAGES; do
  case $st in
    tests)
      rm -f gpurun_out/parity_report.jsonl
      timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider --durations=5 2>&1 | tail -30 > $OUT/${TAG}_pytest.log
      cp gpurun_out/parity_report.jsonl $OUT/${TAG}_parity_report.jsonl 2>/dev/null
      grep -E "passed|failed" $OUT/${TAG}_pytest.log | tail -2 ;;
    bench)
      timeout 900 python bench.py > $OUT/bench_default.out 2> $OUT/bench_default.err
      grep '^{' $OUT/bench_default.out | tail -1 > $OUT/${TAG}_bench_default.json
      python - "$OUT/${TAG}_bench_default.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print('bench:', d['value'], d['unit'], d['ms_per_step'], 'ms/step; roofline', d['roofline'].get('kernel'), d['roofline'].get('frac'),
      '; st_attention', {k: v for k, v in d.get('st_attention', {}).items() if 'frac' in k})
PY
      ;;
    profile)
      bash scripts/profile_bench.sh $TAG 64 > $OUT/profile.log 2>&1
      cp gpurun_out/prof_$TAG/${TAG}_kernel_stats.csv gpurun_out/prof_$TAG/${TAG}_summary.json $OUT/ 2>/dev/null
      grep '^{' gpurun_out/prof_$TAG/bench_under_rocprof.json | tail -1 > $OUT/${TAG}_bench_under_rocprof.json
      python -c "import json; s=json.load(open('$OUT/${TAG}_summary.json')); print('profile:', s.get('meta'), s.get('duration_agreement'))" ;;
    micro)
      timeout 600 python scripts/microbench.py attn hbm conv --out $OUT/${TAG}_microbench.json > $OUT/micro_b8.log 2>&1
      MB_BATCH=64 timeout 600 python scripts/microbench.py hbm conv --out $OUT/${TAG}_microbench_b64.json > $OUT/micro_b64.log 2>&1
      grep -c '"section"' $OUT/micro_b8.log $OUT/micro_b64.log ;;
    models)
      timeout 900 python scripts/bench_models.py lam dyn repr genie4 --cpu-baseline > $OUT/bench_models.out 2> $OUT/bench_models.err
      python - "$OUT" "$TAG" <<'PY'
import json, sys
out, tag = sys.argv[1], sys.argv[2]
rows = [json.loads(l) for l in open(f'{out}/bench_models.out') if l.startswith('{')]
json.dump(rows, open(f'{out}/{tag}_bench_models.json', 'w'), indent=1)
for r in rows:
    print('models:', r['model'][:60], r['ms_per_step'], 'ms;', r.get('roofline', {}).get('kernel'), r.get('roofline', {}).get('frac'), r.get('roofline', {}).get('share_of_step_time'))
PY
      bash scripts/profile_models.sh $TAG > $OUT/profile_models.log 2>&1
      cp gpurun_out/prof_models_$TAG/${TAG}_models_*_kernel_stats.csv $OUT/ 2>/dev/null ;;
    variants)
      : > $OUT/${TAG}_bench_variants.jsonl
      for v in "" "--dp-loopback" "--dp-loopback --grad-compress bf16" "--dp-loopback --allreduce rs_ag" "--async-wgrad 1"; do
        timeout 600 python bench.py --no-cpu-baseline $v 2> $OUT/bench_variant.err | grep '^{' | tail -1 > $OUT/bench_variant.json
        python - "$OUT/bench_variant.json" "$v" >> $OUT/${TAG}_bench_variants.jsonl <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
keep = {k: d.get(k) for k in ('value', 'unit', 'ms_per_step', 'steps', 'warmup', 'n_gpus')}
keep['flags'] = sys.argv[2] or '(default)'
keep['roofline'] = {k: d['roofline'].get(k) for k in ('kernel', 'frac', 'achieved')}
for k in ('comm', 'roofline_in_order', 'wgrad_side_stream'):
    if k in d: keep[k] = d[k]
keep['config'] = {k: d['config'].get(k) for k in ('clips_per_gpu', 'wgrad_stream', 'grad_allreduce', 'peak_mem_GB')}
print(json.dumps(keep))
PY
      done
      python -c "import json; [print('variant:', r['flags'], r['value'], r['ms_per_step'], r['roofline']['frac']) for r in map(json.loads, open('$OUT/${TAG}_bench_variants.jsonl'))]" ;;
    pmc)
      timeout 400 bash scripts/pmc_attn.sh > $OUT/pmc_attn.log 2>&1; cp gpurun_out/pmc_attn/summary.txt $OUT/${TAG}_pmc_attn.txt 2>/dev/null
      MB_BATCH=64 timeout 500 bash scripts/pmc_conv.sh > $OUT/${TAG}_pmc_conv.txt 2>&1 ;;
    guard)
      # kernel-level suite with every Python-side allocation behind guard pages (tests/guard.py), in TWO file orders (round 4's overrun showed
      # in one order only), then the regression: the library with that overrun compiled back in must die under the harness
      F1="tests/test_gpu_kernels.py tests/test_gpu_golden.py tests/test_gpu_attention.py tests/test_gpu_attention_dropout.py tests/test_gpu_maskgit.py tests/test_gpu_linear_ce.py"
      F2="tests/test_gpu_linear_ce.py tests/test_gpu_maskgit.py tests/test_gpu_golden.py tests/test_gpu_attention_dropout.py tests/test_gpu_attention.py tests/test_gpu_kernels.py"
      timeout 1500 python scripts/guard_run.py $F1 -q -m gpu -p no:cacheprovider -x 2>&1 | tail -6 > $OUT/${TAG}_guard_order1.log
      timeout 1500 python scripts/guard_run.py $F2 -q -m gpu -p no:cacheprovider -x 2>&1 | tail -6 > $OUT/${TAG}_guard_order2.log
      make -C open-genie_amd probe -j 8 > $OUT/make_probe.log 2>&1      # incremental: rebuilt whenever the sources (or the ABI) moved since the last probe build
      timeout 900 python scripts/guard_run.py tests/test_gpu_random_geometry.py -q -m gpu -p no:cacheprovider 2>&1 | tail -4 > $OUT/${TAG}_guard_random_geometry.log
      GENIE_GUARD_REGRESSION=1 timeout 600 python -m pytest tests/test_gpu_guard.py -q -m gpu -p no:cacheprovider 2>&1 | tail -4 > $OUT/${TAG}_guard_regression.log
      for f in $OUT/${TAG}_guard_order1.log $OUT/${TAG}_guard_order2.log $OUT/${TAG}_guard_random_geometry.log $OUT/${TAG}_guard_regression.log; do tail -n 2 $f; done ;;
    multigpu)
      # first contact with more than one GPU (no gpurun box has had two so far): the plain launcher branch of bench.py, both all-reduce
      # algorithms, fp32 and bf16 payload -> <tag>_bench_gpus2.jsonl with the `comm` object of each; skipped on a one-GPU box
      NG=$(python -c "import torch; print(torch.cuda.device_count())")
      if [ "$NG" -ge 2 ]; then
        : > $OUT/${TAG}_bench_gpus2.jsonl
        for v in "" "--allreduce rs_ag" "--grad-compress bf16"; do
          timeout 900 python bench.py --gpus 2 --steps 6 --warmup 2 $v 2> $OUT/bench_gpus2.err | grep '^{' | tail -1 >> $OUT/${TAG}_bench_gpus2.jsonl
        done
        timeout 900 python -m pytest tests/test_gpu_trainer.py -q -m gpu -p no:cacheprovider -k two_rank 2>&1 | tail -3 > $OUT/${TAG}_two_rank_test.log
        python -c "import json; [print('gpus2:', r['value'], r['ms_per_step'], r.get('comm', {}).get('algorithm'), r.get('comm', {}).get('exposed_ms_per_step')) for r in map(json.loads, open('$OUT/${TAG}_bench_gpus2.jsonl'))]"
      else
        echo "multigpu: $NG GPU visible, skipped"
      fi ;;
    *) echo "unknown stage $st" ;;
  esac
done
ls $OUT
