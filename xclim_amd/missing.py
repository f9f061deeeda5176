"""Host mirror of the missing-value check on the hot path (reference: core/missing.py:64-160 expected_count, :201-220
MissingBase.is_valid / __call__, :318-322 MissingAny.is_missing).

The index functions of :mod:`xclim_amd.indices` fuse this check into their kernels (a valid-count side output +
``xh_apply_missing_mask``); this module is the stand-alone form ``missing_any(da, freq)`` of ``xclim.core.missing``.
"""

from __future__ import annotations

import numpy as np

from . import kernels as K
from ._capi import get_device
from .calendar import _flatten
from .timeaxis import TimeAxis


def expected_count(time: TimeAxis, freq: str, **indexer) -> np.ndarray:
    """core/missing.py:64-160 for a daily source: days of every full period (the selected days with an indexer)."""
    return time.expected_count(freq, **indexer)


def missing_any(da, freq: str, time: TimeAxis, *, device=None, **indexer) -> np.ndarray:
    """core/missing.py:318-322: True where a period holds fewer valid (non-NaN) steps than expected.  With ``**indexer``
    the series is masked by select_time first and only the selected days are expected (core/missing.py:118-135)."""
    dev = device or get_device()
    x, cell_shape = _flatten(da, dev)
    if indexer and any(v is not None for k, v in indexer.items() if k != "include_bounds"):
        from .calendar import select_time

        x = select_time(x, time, device=dev, keep=True, **indexer)
    seg, _ = time.segments(freq)
    _, valid = K.resample_reduce(dev, x, "count", seg)
    exp = expected_count(time, freq, **indexer).reshape((-1,) + (1,) * len(cell_shape))
    v = valid.get().reshape((valid.shape[0],) + tuple(cell_shape))
    return v != exp
