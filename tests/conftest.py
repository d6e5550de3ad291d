import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "exhaustive: the remaining views of the eight-view configurations; only selected "
                                       "when the -m expression names it (-m \"gpu and exhaustive\" / -m \"gpu or exhaustive\")")


def pytest_collection_modifyitems(config, items):
    """`-m gpu` alone must stay inside the driver's step limit: tests marked `exhaustive` are deselected unless the
    marker expression mentions them."""
    if "exhaustive" in (config.getoption("-m") or ""):
        return
    keep, drop = [], []
    for it in items:
        (drop if it.get_closest_marker("exhaustive") else keep).append(it)
    if drop:
        config.hook.pytest_deselected(items=drop)
        items[:] = keep


@pytest.fixture(scope="session")
def hip_lib():
    """Builds (if stale) and loads libg4s_hip.so; GPU tests must go through it."""
    from g4splat_amd import _lib, build
    build.build()
    return _lib.load()


@pytest.fixture(scope="session")
def oracle_mod():
    from oracle import oracle
    oracle.build()
    return oracle
