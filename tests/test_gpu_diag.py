"""The diagnostics / A-B kernels (trace twin 70, streaming kernels 83 / 84) are not in the product library; they are
built into lib/libagxntt_diag.so by `make diag`.  A process binds ONE library, so these checks run in a child
process with AGX_NTT_LIB pointing at the diag build (tests/diag_child.py)."""
import ctypes
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_product_library_has_no_debug_hook(agx):
    raw = ctypes.CDLL(agx.LIB_PATH)
    assert not hasattr(raw, "agx_ntt_debug_set_trace_buffer")
    with pytest.raises(RuntimeError):
        agx.debug_set_trace_buffer(0, 0)


@pytest.mark.gpu
def test_diag_library_kernels_against_oracle(agx):
    if not os.path.exists(agx.DIAG_LIB_PATH):
        agx.build_diag()
    env = dict(os.environ, AGX_NTT_LIB=agx.DIAG_LIB_PATH)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "diag_child.py")], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0 and "DIAG OK" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]
