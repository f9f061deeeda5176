"""Host mirror of the NaN-aware quantile helpers of ``xclim.core.utils`` (reference: core/utils.py:279-367).

``calc_perc`` is the numpy callee that ``percentile_doy`` hands to ``xr.apply_ufunc`` (cal:469-479): the core dim
arrives LAST as a possibly strided view and the percentiles come back on a new last axis — same contract here.
"""

from __future__ import annotations

import numpy as np

from . import kernels as K
from ._capi import get_device


def nan_calc_percentiles(arr, percentiles=None, axis: int = -1, alpha: float = 1.0, beta: float = 1.0, copy: bool = True,
                         *, device=None) -> np.ndarray:
    """utl:326-367: quantile axis FIRST in the result, float64.  ``copy`` is a no-op (inputs are never mutated)."""
    pers = [50.0] if percentiles is None else list(percentiles)
    a = np.asarray(arr)
    if a.size == 0:
        return np.nan
    a = np.moveaxis(a, axis, -1)
    lead = a.shape[:-1]
    N = a.shape[-1]
    dev = device or get_device()
    # (C, N) sample-minor; float64 samples keep their dtype (xh_nan_quantile_f64: `diff` in float64, utl:486)
    flat = np.ascontiguousarray(a.reshape(-1, N), dtype=np.float64 if a.dtype == np.float64 else np.float32)
    q = np.array([p / 100.0 for p in pers])
    out = K.nan_quantile(dev, dev.to_device(flat), q, alpha, beta, sample_axis=1).get()  # (nq, C)
    return out.reshape((len(pers),) + lead)


def calc_perc(arr, percentiles=None, alpha: float = 1.0, beta: float = 1.0, copy: bool = True, *, device=None):
    """utl:279-323: percentiles along the last axis, percentile axis moved LAST."""
    return np.moveaxis(nan_calc_percentiles(arr, percentiles, -1, alpha, beta, copy, device=device), 0, -1)
