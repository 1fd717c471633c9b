"""pytest config: registers the `gpu` marker and puts the repo root on sys.path."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "perf: wall-clock comparisons on a real MI355X (run with -m perf on an otherwise idle box; "
                                       "never part of the parity run: a noisy neighbour fails them without a code defect)")


def pytest_collection_modifyitems(config, items):
    """`perf` tests run only when asked for by name (-m perf): neither `-m gpu` nor `-m "not gpu"` selects them."""
    if "perf" in (config.getoption("-m") or ""):
        return
    keep, drop = [], []
    for it in items:
        (drop if it.get_closest_marker("perf") else keep).append(it)
    if drop:
        config.hook.pytest_deselected(items=drop)
        items[:] = keep


def _has_gpu() -> bool:
    try:
        from oscillink_amd import _native

        return _native.device_count() > 0
    except Exception:
        return False


@pytest.fixture(scope="session")
def gpu_available():
    return _has_gpu()
