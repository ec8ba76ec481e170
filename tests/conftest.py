import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import oracle_lib
    oracle_lib.lib()
    return oracle_lib


@pytest.fixture(scope="session")
def zra():
    """The product library through its C ABI. Fails loudly if the HIP extension is not built."""
    import zra_amd
    zra_amd.load()
    return zra_amd


@pytest.fixture(scope="session")
def gpu_engine(zra):
    if zra.load().ZraHipDeviceCount() < 1:
        pytest.fail("gpu test selected but no HIP device is visible")
    return zra.Engine(0)
