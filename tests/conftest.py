import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "cuda-volpath_amd"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import oracle_lib
    oracle_lib.build()
    oracle_lib.lib()
    return oracle_lib


@pytest.fixture(scope="session")
def vp():
    """The HIP product through its C ABI. Fails loudly when the library or the GPU is missing."""
    import volpath
    volpath.lib()
    if volpath.device_count() < 1:
        raise RuntimeError("gpu-marked test without a visible HIP device")
    volpath.set_device(0)
    return volpath
