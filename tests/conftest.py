import os
import sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def orc():
    import oracle
    return oracle.Oracle()


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
