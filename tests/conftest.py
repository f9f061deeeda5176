import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def dev():
    """The HIP context.  GPU tests must FAIL (not skip) when the library or the device is missing."""
    from xclim_amd._capi import get_device

    return get_device(0)


@pytest.fixture
def rng():
    """Seeded generator; XH_TEST_SEED re-runs the whole suite on other draws (the parity tests hold for every seed)."""
    return np.random.default_rng(int(os.environ.get("XH_TEST_SEED", "20240925")))
