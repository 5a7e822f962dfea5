import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# Which polar kernel the suite runs by default. The library's default for an exact 2x upscale is
# k_polar_mx (the contraction on the f16 matrix pipe), whose parity statement is "+-1 code of 16
# bits": tests/test_gpu_polar_mfma.py and tests/test_gpu_metric.py switch it on explicitly and
# hold it to that, at the same geometries. Every other test pins k_polar_pp, the sequential-fma
# kernel that evaluates the taps in the reference's order and must match the oracle BIT FOR BIT.
# The whole suite also runs on the library's defaults (`PL_HIP_POLAR_MFMA=1 pytest -m gpu`,
# profiles/r04_*_gputests_mfma_forced.log): every comparison that involves a polar pass goes
# through util.assert_polar_equal, which is exact here and "at most one code / one f16 ulp" there.
os.environ.setdefault("PL_HIP_POLAR_MFMA", "0")


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
