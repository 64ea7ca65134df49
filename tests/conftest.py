import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import pyoracle
    pyoracle.lib()
    return pyoracle


@pytest.fixture(scope="session")
def hip():
    """The product library on a real device; fails loudly when either is missing."""
    from soillib_amd import _abi
    lib = _abi.lib()
    assert lib.soil_device_count() > 0, "no HIP device visible: GPU tests cannot run"
    return lib
