import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import oracle_py
    oracle_py.build()
    oracle_py.lib()
    return oracle_py


@pytest.fixture(scope="session")
def hip():
    """The product library through its C ABI.  Fails loudly when it is not built or sees no GPU."""
    import lcqpow_amd
    lcqpow_amd.lib()
    if lcqpow_amd.device_count() < 1:
        pytest.fail("liblcqpow_hip.so loaded but no GPU is visible: GPU tests must run on the HIP path")
    return lcqpow_amd
