"""The guard-page harness itself (tests/guard.py, csrc/guard.hip): allocations end on the last mapped byte, the patched allocation paths
hand out such tensors, a kernel-level call pattern with slice views runs clean under it -- and (opt-in, it ends in a deliberate GPU page
fault) the library with round 4's chan_sum_kernel overrun compiled back in (lib/libgenie_hip_oobprobe.so) dies under it on every run."""
import os
import subprocess
import sys

import pytest
import torch

from util import ROOT

pytestmark = pytest.mark.gpu


def test_guard_allocations_end_on_the_mapping():
    import guard
    with guard.AllocationGuard():
        a = torch.randn(3, 5, 7).cuda()
        b = torch.empty(4096, dtype=torch.bfloat16, device='cuda')
        c = torch.zeros((2, 8, 2, 3, 16), dtype=torch.bfloat16, device='cuda')
        d = torch.randn(4, 16, device='cuda')
        for t in (a, b, c, d):
            assert t.is_cuda and t.data_ptr() % 16 == 0
            end = t.data_ptr() + guard._span(t) * t.element_size()
            assert (-end) % 4096 <= 15, (hex(t.data_ptr()), hex(end))        # at most the 16-byte alignment slack before the unmapped page
        b.fill_(1.0)
        assert b.float().sum().item() == 4096 and c.abs().sum().item() == 0
        assert torch.equal(a.cpu(), a.clone().cpu())
    assert guard.STATS['blocks'] >= 4
    assert not hasattr(torch.empty, '__wrapped__')                          # patches removed


def _run_case(lib=None):
    env = dict(os.environ)
    if lib:
        env['GENIE_HIP_LIB'] = lib
    return subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'guard_case_blur.py')], capture_output=True, text=True, timeout=300, env=env)


def test_grouped_blur_runs_clean_under_the_guard():
    r = _run_case()
    assert r.returncode == 0 and 'guard case ok' in r.stdout, (r.returncode, r.stdout[-400:], r.stderr[-800:])


@pytest.mark.skipif(os.environ.get('GENIE_GUARD_REGRESSION') != '1', reason='deliberate GPU page fault: opt in with GENIE_GUARD_REGRESSION=1 (scripts/evidence.sh guard)')
def test_guard_catches_the_round4_chan_sum_overrun():
    """lib/libgenie_hip_oobprobe.so = this library with chan_sum_kernel walking the PITCH of a channel-slice view again (-DGENIE_REINTRODUCE_CHANSUM_OOB):
    16 bytes read behind the last pixel of the last group.  Without the harness that read hit whatever the caching allocator had placed there;
    under it the process dies -- deterministically."""
    lib = os.path.join(ROOT, 'open-genie_amd', 'lib', 'libgenie_hip_oobprobe.so')
    assert os.path.exists(lib), 'make -C open-genie_amd probe builds it (here, before gpurun ships the tree)'
    for _ in range(2):
        r = _run_case(lib)
        assert r.returncode != 0 and 'guard case ok' not in r.stdout, (r.returncode, r.stdout[-300:])
        assert 'fault' in (r.stderr + r.stdout).lower() or r.returncode < 0, r.stderr[-600:]
