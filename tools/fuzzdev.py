"""The device a fuzzer runs on: the GPU (default), or — FUZZ_DEVICE=hostsim — the host simulation of tests/hostsim, where the
kernels run on the CPU (thread by thread / on fibers; the ISA-level ones with their few ISA statements rewritten to C++, see
tests/hostsim/simdevice.py).  Only what needs rocPRIM is missing there (xh_adapt_freq, QDM's exact-rank path beyond 32768 steps)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def get_fuzz_device():
    if os.environ.get("FUZZ_DEVICE") == "hostsim":
        import tempfile

        from tests.hostsim import simdevice

        return simdevice.SimDevice(simdevice.build(tempfile.mkdtemp(prefix="hostsim_")))
    from xclim_amd._capi import get_device

    return get_device()
