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
def dev(tmp_path_factory):
    """The HIP context.  GPU tests must FAIL (not skip) when the library or the device is missing.

    XH_TEST_DEVICE=hostsim (an AUDIT switch, never the default): the `-m gpu` tests run against the host simulation of
    tests/hostsim instead — `XH_TEST_DEVICE=hostsim pytest -m gpu -n 8 --timeout 300 --deselect tests/test_gpu_fullsize.py`
    shows which of them the simulation can serve (all but those that need rocPRIM, RCCL or the real device's input cache; a
    collective under divergent control flow would abort the worker)."""
    if os.environ.get("XH_TEST_DEVICE") == "hostsim":
        from tests.hostsim import simdevice

        lib = os.environ.get("HOSTSIM_LIB")   # (a library some caller built already: tests/test_hostsim_cpu.py)
        return simdevice.SimDevice(lib if lib and os.path.exists(lib) else simdevice.build(str(tmp_path_factory.mktemp("hostsim"))))
    from xclim_amd._capi import get_device

    return get_device(0)


@pytest.fixture
def rng():
    """Seeded generator; XH_TEST_SEED re-runs the whole suite on other draws (the parity tests hold for every seed)."""
    return np.random.default_rng(int(os.environ.get("XH_TEST_SEED", "20240925")))
