import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def built():
    """Make sure the C/HIP library and the oracle exist (build() is idempotent)."""
    import __graft_entry__ as g
    g.build()
    return True


@pytest.fixture(scope="session")
def gpu(built):
    import libplacebo_amd as pl
    if pl.hip_device_count() < 1:
        pytest.fail("no HIP device visible: -m gpu tests need a real GPU "
                    "(there is no CPU fallback to silently pass on)")
    g = pl.HipGpu(0)
    yield g
    g.close()
