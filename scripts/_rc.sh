cd /root/repo
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "narrow_out" 2>&1 | tail -3
for i in 1 2; do for c in 1 2; do GENIE_NARROW_OUT_CUT=$c AB_TAG=cut$c timeout 300 python scripts/ab_narrow.py head_fwd 2>&1 | grep "^{" ; done; done
