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
    """The CPU oracle (test infrastructure, oracle/hz_oracle.c)."""
    from oracle import oracle
    oracle.build()
    return oracle


@pytest.fixture(scope="session")
def hip():
    """The product package; the HIP library must be present and a GPU visible."""
    import horayzon_amd
    from horayzon_amd import _lib
    _lib.lib()
    if _lib.device_count() < 1:
        pytest.fail("no HIP device visible: -m gpu tests need an MI355X")
    return horayzon_amd
