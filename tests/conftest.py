import os
import sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def _orc_libm():
    import oracle
    return oracle.Oracle()


@pytest.fixture(scope="session")
def orc_sm():
    """the oracle built with the product's shared transcendental functions (oracle/liboracle_sm.so, csrc/ctl_fmath.h)"""
    import oracle
    return oracle.Oracle(shared_math=True)


@pytest.fixture
def orc(request, _orc_libm):
    """The oracle.  CPU tests get the glibc build — the reference's CPU path, pinned on the reference's own code.  Tests marked `gpu` get the shared-math build: the HIP
    kernels evaluate sin / cos / acos / atan2 / exp / log / pow with the same fp32 implementation, so GPU and checker differ only where a test says so.
    tests/test_fmath.py holds the two builds together."""
    if request.node.get_closest_marker("gpu") is not None:
        return request.getfixturevalue("orc_sm")
    return _orc_libm


@pytest.fixture(scope="session")
def ref():
    import oracle
    r = oracle.load_ref()
    if r is None:
        pytest.skip("oracle/_ref/libctlref.so not built (needs /root/reference; run `make -C oracle ref`)")
    return r


@pytest.fixture(scope="session")
def ctl():
    import cudatracerlib_amd
    return cudatracerlib_amd


@pytest.fixture(scope="session")
def gpu(ctl):
    if ctl.device_count() < 1:
        pytest.fail("-m gpu tests need a HIP device; none visible")
    return ctl
