import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def orc():
    """CPU oracle (test infrastructure only)."""
    from oracle import oracle

    oracle.build()
    return oracle


@pytest.fixture(scope="session")
def agx():
    """Product library binding; building it is part of the session setup."""
    import agilex_ntt_amd as a

    if not os.path.exists(a.LIB_PATH):
        a.build()
    a.lib()
    return a


@pytest.fixture(scope="session")
def dev(agx):
    """torch device helpers for the -m gpu tests; the HIP library must see a GPU."""
    import torch

    assert torch.cuda.is_available(), "GPU tests need a visible MI355X"
    assert agx.device_count() >= 1, "libagxntt.so sees no HIP device"
    from gpu_util import DeviceHelper

    return DeviceHelper(torch)
