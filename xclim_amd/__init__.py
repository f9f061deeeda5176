"""xclim_amd — MI355X (gfx950) backend for xclim's index / run-length / percentile / quantile-mapping hot path.

Layers: ``csrc/`` hand-written HIP kernels behind the C ABI of ``include/xclim_hip.h`` ->
``_capi`` (ctypes) -> ``kernels`` (typed wrappers on device arrays) -> host mirrors of the reference modules
(``generic``, ``run_length``, ``calendar``, ``utils``, ``sdba``, ``indices``).  There is no CPU fallback.
"""

__version__ = "0.1.0"
