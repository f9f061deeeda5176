"""xclim_amd — MI355X (gfx950) backend for xclim's index / run-length / percentile / quantile-mapping hot path.

Layers: ``csrc/`` hand-written HIP kernels behind the C ABI of ``include/xclim_hip.h`` ->
``_capi`` (ctypes) -> ``kernels`` (typed wrappers on device arrays) -> host mirrors of the reference modules
(``generic``, ``run_length``, ``calendar``, ``utils``, ``sdba``, ``indices``).  There is no CPU fallback.
"""

__version__ = "0.1.0"


def keep_inputs(device=None):
    """``with xclim_amd.keep_inputs(): ...`` — inside the block a large host field handed to several calls (``percentile_doy``
    then ``tx90p``; an index then its missing-value check) crosses PCIe ONCE (``Device.resident``).  The contract: no
    field is edited in place inside the block.  Outside such a block every call uploads its inputs."""
    from ._capi import get_device

    return (device or get_device()).keep_inputs()
