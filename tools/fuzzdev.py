"""The device a fuzzer runs on: the GPU (default), or — FUZZ_DEVICE=hostsim — the host simulation of tests/hostsim, where the
kernels that are not written at ISA level run on the CPU (thread by thread / on fibers).  On the simulation the fuzzers exercise
the GENERAL kernels (the ISA-level fast paths decline there), i.e. the fall-backs the GPU runs rarely reach."""
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
