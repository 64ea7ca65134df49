import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# the library's bounded waits on its RCCL wire (include/soil_slab.h), tightened for the tests: a
# transfer that does not complete fails its test within seconds, with the operation named
os.environ.setdefault("SOIL_RCCL_TIMEOUT_S", "20")
os.environ.setdefault("SOIL_RCCL_INIT_TIMEOUT_S", "45")


# Spent debris walkers (include/soil_hip.h: soil_set_debris_retire) are WATCHED throughout the suite — marked,
# walked to the end as the reference walks them (so the step counts equal the oracle's) and every deposit of a
# marked walker that is not an exact zero counted; the fixture below wants that count to be zero after every
# GPU test.  The product's default (they end their walks) is what tests/test_debris_retire.py runs.  In the
# environment, so that the compiled C++ hosts and spawned workers of the tests inherit it.
os.environ.setdefault("SOIL_DEBRIS_RETIRE", "2")


@pytest.fixture(autouse=True)
def _no_spent_walker_ever_adds_anything(request):
    yield
    if request.node.get_closest_marker("gpu") is None:
        return
    from soillib_amd import _abi, soil
    if _abi.lib().soil_device_count() > 0:
        assert soil.debris_retire_violations(reset=True) == 0, "a debris walker marked as spent went on to add something"


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


# Parity first.  `pytest -x` stops at the first failure, so the order of the session decides what a
# failure hides: the oracle-parity files of the hot path run before the infrastructure tests (compiled
# C++ hosts, spawned bench processes) and the RCCL / multi-process tests run last (one
# hung RCCL test in front of the suite left 381 parity tests unreached).
_ORDER = ["test_gpu_parity", "test_debris_retire", "test_golden", "test_oracle_kat", "test_gpu_oracle_fullsize", "test_fast_particles",
          "test_fma_tolerance", "test_gpu_window_shapes", "test_gpu_fullsize", "test_gpu_api_surface",
          "test_abi_symbols", "test_io_tiff", "test_knobs", "test_parallel_gloo", "test_cpp_api", "test_bench_configs",
          "test_bench_spawn", "test_gpu_parallel"]
_LAST_WITHIN = ("rccl", "eight_processes")   # within a file: the RCCL wire and the 8-process job at the very end


def pytest_collection_modifyitems(session, config, items):
    def key(item):
        name = os.path.splitext(os.path.basename(str(item.fspath)))[0]
        rank = _ORDER.index(name) if name in _ORDER else len(_ORDER) - 3   # unknown files: before the C++/spawn/RCCL tail
        late = any(w in item.name.lower() for w in _LAST_WITHIN)
        return (2 * len(_ORDER) if late else rank)
    items.sort(key=key)   # stable: the order inside a file is kept
    # and no single test may sit on the session: pytest-timeout (in the image) fails a test after 300 s
    # (the 8-process 16384^2 job: 600 s) and the session goes on
    if config.pluginmanager.hasplugin("timeout"):
        for item in items:
            if item.get_closest_marker("timeout") is None:
                item.add_marker(pytest.mark.timeout(600 if "eight_processes" in item.name else 300))
