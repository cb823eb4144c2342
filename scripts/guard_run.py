#!/usr/bin/env python
"""pytest, in-process, with every Python-side CUDA allocation in its own guard-page mapping (tests/guard.py): an out-of-bounds access of any
kernel is a page fault -- the process aborts with "Memory access fault by GPU" -- on every run.

    python scripts/guard_run.py tests/test_gpu_kernels.py tests/test_gpu_attention.py -q -m gpu
    python scripts/guard_run.py --selftest          (allocations are tail-aligned; prints the mapping granularity)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'open-genie_amd'), os.path.join(ROOT, 'tests')):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch   # noqa: E402

import guard   # noqa: E402


def selftest():
    with guard.AllocationGuard():
        x = torch.randn(3, 5, 7).cuda()
        y = torch.empty(1000, dtype=torch.bfloat16, device='cuda')
        z = torch.zeros((4, 8, 2, 3, 3), dtype=torch.bfloat16, device='cuda').permute(0, 4, 1, 2, 3)
        w = torch.randn(2, 6, device='cuda')[:, :4]
        for t in (x, y):
            end = t.data_ptr() + guard._span(t) * t.element_size()
            assert end % 4096 <= 15 or (4096 - end % 4096) % 4096 <= 15, (hex(t.data_ptr()), hex(end))
        assert torch.equal(x.cpu(), x.cpu()) and y.numel() == 1000 and z.shape == (4, 3, 8, 2, 3)
        print('guard selftest ok:', guard.STATS, 'x ends', hex(x.data_ptr() + x.numel() * 4), flush=True)


if __name__ == '__main__':
    if '--selftest' in sys.argv:
        selftest()
        sys.exit(0)
    import pytest
    with guard.AllocationGuard():
        rc = pytest.main(sys.argv[1:])
    print('guard stats:', guard.STATS, flush=True)
    sys.exit(int(rc))
