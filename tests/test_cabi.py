"""The C-ABI shared library loads WITHOUT a GPU and exports every symbol include/genie_hip.h declares
(no compute is launched here)."""
import ctypes
import os
import re

import pytest

from util import ROOT

HEADER = os.path.join(ROOT, 'include', 'genie_hip.h')


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(genie_[a-z0-9_]+)\s*\(', src)))


def test_library_exports_every_declared_symbol():
    from genie import _hip
    if not os.path.exists(_hip.LIB_PATH):
        pytest.fail(f'{_hip.LIB_PATH} missing: run `python -c "import __graft_entry__ as g; g.build()"`')
    lib = ctypes.CDLL(_hip.LIB_PATH)
    names = declared_functions()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f'libgenie_hip.so does not export {n}'
    # the python binding covers the same set
    assert set(_hip.SIGNATURES) == set(names), set(_hip.SIGNATURES) ^ set(names)


def test_abi_version_and_error_channel():
    from genie import _hip
    lib = _hip.load_library()
    assert lib.genie_abi_version() == _hip.ABI_VERSION
    # argument validation happens before any device work: a null descriptor is rejected with a message
    rc = lib.genie_conv_igemm(None, None)
    assert rc == -1
    assert b'null descriptor' in lib.genie_last_error()
    rc = lib.genie_lfq_quantize(1, 0, 4, 1, 99, 128, None, 1, None)
    assert rc == -1 and b'codebook_dim' in lib.genie_last_error()


def test_struct_layouts_match_header():
    """ctypes mirrors of GenieTap / GenieConvDesc / GenieWgradDesc have the sizes the C compiler gives them."""
    import subprocess
    import tempfile
    from genie import _hip
    src = '#include <stdio.h>\n#include "genie_hip.h"\nint main(){printf("%zu %zu %zu\\n", sizeof(GenieTap), sizeof(GenieConvDesc), sizeof(GenieWgradDesc));return 0;}\n'
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, 't.c')
        open(c, 'w').write(src)
        exe = os.path.join(d, 't')
        subprocess.run(['gcc', '-I', os.path.join(ROOT, 'include'), c, '-o', exe], check=True)
        sizes = [int(v) for v in subprocess.run([exe], capture_output=True, text=True, check=True).stdout.split()]
    assert sizes == [ctypes.sizeof(_hip.GenieTap), ctypes.sizeof(_hip.GenieConvDesc), ctypes.sizeof(_hip.GenieWgradDesc)]


def test_roctx_ranges_behind_env_switch():
    """GENIE_ROCTX=1 brackets every enqueueing C-ABI call with a roctx range named after the entry point (tracing):
    the proxy loads a roctx library, forwards arguments and return codes unchanged, and leaves the query entry points alone.  (In a child
    process: the switch is read when the library is first loaded.)"""
    import subprocess
    import sys
    code = ('import sys; sys.path[:0] = [%r, %r]\n'
            'from genie import _hip\n'
            'lib = _hip.load_library()\n'
            'assert type(lib).__name__ == "_RoctxProxy" and lib.genie_abi_version() == _hip.ABI_VERSION\n'
            'assert lib.genie_conv_igemm(None, None) == -1 and b"null descriptor" in lib.genie_last_error()\n'
            'assert lib.genie_attention_lean_mode(-1) >= 0\n'
            'print("ok")\n') % (ROOT, os.path.join(ROOT, 'open-genie_amd'))
    r = subprocess.run([sys.executable, '-c', code], env=dict(os.environ, GENIE_ROCTX='1'), capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.strip().endswith('ok'), r.stderr[-2000:]
