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
    # PyTorch bundles its own libamdhip64 (ROCm 7.0 here); the library links the system's (7.2).  Whichever loads first
    # serves both.  With the system's first, the first `import torch` in this process was seen to take 9 minutes on a fresh GPU
    # box (round 5, tests/test_gpu_parity.py::test_host_output_that_is_already_page_locked: 526 s) -- the suite ran in 5 minutes
    # whenever a torch-importing test module came first.  The order is pinned here: torch first, if it is installed.
    try:
        import torch  # noqa: F401
    except Exception:
        pass
    import horayzon_amd
    from horayzon_amd import _lib
    _lib.lib()
    if _lib.device_count() < 1:
        pytest.fail("no HIP device visible: -m gpu tests need an MI355X")
    # HZ_TEST_SCHEDULE="persist_grid=5,left_min=0x1818": run the whole session under another launch schedule (results never depend
    # on it: horayzon_amd.horizon.schedule_overrides) -- how the long fuzz sweeps exercise the block loop and the hand-over levels
    for item in filter(None, os.environ.get("HZ_TEST_SCHEDULE", "").split(",")):
        k, v = item.split("=")
        horayzon_amd.horizon.schedule_overrides[k.strip()] = int(v, 0)
    return horayzon_amd
