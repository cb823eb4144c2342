#!/bin/bash
# round-4 GPU call 2: SQ counters of the attention kernel families, the other BASELINE configs on the lean kernels, the LAM parity test
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 330 bash scripts/pmc_attn.sh > gpurun_out/r04_pmc_attn.log 2>&1
timeout 240 python scripts/bench_models.py lam dyn genie4 > gpurun_out/r04_bench_models_call2.log 2>&1
(timeout 300 python -m pytest tests/test_gpu_lam.py -q -x -p no:cacheprovider 2>&1 | tail -15) > gpurun_out/r04_call2_lam.log 2>&1
cat gpurun_out/pmc_attn/summary.txt
grep -o '"model": "[^"]*"\|"ms_per_step": [0-9.]*\|"roofline": {[^}]*}' gpurun_out/r04_bench_models_call2.log
tail -5 gpurun_out/r04_call2_lam.log
grep end_to_end gpurun_out/parity_report.jsonl | tail -3
