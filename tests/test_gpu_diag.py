"""The diagnostics / A-B kernels (trace twin 70, the A/B twins 147 / 161 / 115 / 160 / 114) are not in the product library; they are
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


@pytest.mark.gpu
def test_ab_registry_entries_under_the_diag_library(agx):
    """the A/B entries of the kernel registry are not in the product library; the id-parametrised parity tests of
    tests/test_gpu_parity.py skip them there.  Here the same tests run once more in ONE child process bound to
    lib/libagxntt_diag.so (AGX_NTT_LIB), where no id is skipped."""
    if not os.path.exists(agx.DIAG_LIB_PATH):
        agx.build_diag()
    env = dict(os.environ, AGX_NTT_LIB=agx.DIAG_LIB_PATH)
    pick = ("test_n4096_kernel_registry_variants or test_every_registry_entry_at_its_own_size or test_loop_kernels_more_frames_than_workgroups "
            "or test_dynamic_loop_kernels_on_two_streams_of_one_plan or test_large_frames_on_many_rounds_of_workgroups "
            "or test_one_launch_product_with_aliasing_and_no_scratch")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_parity.py"), "-m", "gpu", "-q", "-x", "-k", pick,
                        "-p", "no:cacheprovider"], capture_output=True, text=True, timeout=1500, env=env, cwd=ROOT)
    tail = r.stdout[-3000:] + r.stderr[-2000:]
    assert r.returncode == 0, tail
    assert " passed" in r.stdout and "skipped" not in r.stdout.splitlines()[-1], tail
    # the child's own count, into this test's captured output (-s / -rP / the junit log show it) and into a file beside the other records
    summary = r.stdout.splitlines()[-1]
    print(f"child pytest under lib/libagxntt_diag.so: {summary}")
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "diag_child_pytest_summary.txt"), "w") as f:
            f.write(summary + "\n")
    except OSError:
        pass
