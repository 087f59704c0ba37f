"""pytest configuration: markers, import path, shared helpers."""
import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


# Property tests (hypothesis) draw the SAME examples on every run unless BS_HYPOTHESIS_RANDOM=1: a suite that is green here is green in the driver's run;
# the random mode is for looking for new counter-examples (found ones are pinned with @example).
try:
    from hypothesis import settings as _hs
    _hs.register_profile("fixed", derandomize=True, database=None)
    _hs.register_profile("random", database=None)
    _hs.load_profile("random" if os.environ.get("BS_HYPOTHESIS_RANDOM") == "1" else "fixed")
except ImportError:                                   # (the tests that need it skip themselves)
    pass


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` through gpurun)")


@pytest.fixture(scope="session")
def bsa():
    """The product package (directory name has a hyphen, so import by string)."""
    return importlib.import_module("batch-scheduler_amd")


@pytest.fixture(scope="session")
def soa():
    return importlib.import_module("batch-scheduler_amd.soa")


@pytest.fixture(scope="session")
def orc():
    import orc as _orc
    _orc.build()
    return _orc


@pytest.fixture(scope="session")
def gpu_ctx_factory(bsa):
    """Factory for HIP contexts; fails loudly (no CPU fallback) when the extension or GPU is missing."""
    def make(scalar_lanes=0, eph_gate=1, timing=0):
        return bsa.Context(scalar_lanes=scalar_lanes, eph_gate=eph_gate, enable_timing=timing)
    return make
