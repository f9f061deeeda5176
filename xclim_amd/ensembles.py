"""Host mirror of ``xclim.ensembles.ensemble_percentiles`` (reference: src/xclim/ensembles/_base.py:213-372) — the
second caller of ``calc_perc``.

The ensemble is a numpy / device array with the REALIZATION axis first: ``(realization, ...)``; the percentiles of the
members are taken per remaining element with the NaN-aware Hyndman-Fan estimate of ``core/utils.py:370-557`` — the same
kernel as ``percentile_doy`` (``xh_nan_quantile``), samples on the slow axis so the members are read coalesced.
"""

from __future__ import annotations

import numpy as np

from . import kernels as K
from ._capi import DeviceArray, get_device, handle_float64

# _base.py:21-28
_quantile_params = {
    "interpolated_inverted_cdf": (0, 1),
    "hazen": (0.5, 0.5),
    "weibull": (0, 0),
    "linear": (1, 1),
    "median_unbiased": (1 / 3, 1 / 3),
    "normal_unbiased": (3 / 8, 3 / 8),
}


def ensemble_percentiles(ens, values=None, min_members: int | None = 1, weights=None, method: str = "linear", *,
                         device=None, keep: bool = False):
    """_base.py:213-372 for one variable.  Returns ``(..., percentiles)`` (percentile axis LAST like the reference's
    ``output_core_dims=[["percentiles"]]``) in float64, or the device array ``(nper, C)`` with ``keep=True``.
    Elements with fewer than ``min_members`` valid members are NaN (``None``: all members required)."""
    if values is None:
        values = [10, 50, 90]
    if method not in _quantile_params:
        raise KeyError(method)
    if weights is not None and method != "linear":
        raise ValueError("Only the 'linear' method is supported when using weights.")  # _base.py:347-348
    alpha, beta = _quantile_params[method]
    dev = device or get_device()
    if isinstance(ens, DeviceArray):
        x, lead = ens.reshape(ens.shape[0], -1), ens.shape[1:]
    else:
        a = np.asarray(ens)
        lead = a.shape[1:]
        if a.dtype == np.float64 and weights is None:  # float64 members: xh_nan_quantile_f64 (`diff` in float64, utl:486)
            x = dev.to_device(np.ascontiguousarray(a.reshape(a.shape[0], -1)))
        else:
            handle_float64(a, "ensemble_percentiles (weighted)")
            x = dev.to_device(np.ascontiguousarray(a.reshape(a.shape[0], -1), dtype=np.float32))
    R = x.shape[0]
    if min_members is None:
        min_members = R
    q = np.array([v / 100.0 for v in values], dtype=np.float64)
    if weights is not None:
        # _base.py:350-356: xarray's weighted quantile (parity unpinned: xarray is not available; see wquantile.hip)
        out = K.weighted_quantile(dev, x, weights, q)
    else:
        out = K.nan_quantile(dev, x, q, alpha, beta, sample_axis=0)  # (nper, C) float64
    if min_members != 1:
        seg = np.array([0, R], dtype=np.int64)
        nvalid, _ = K.resample_reduce(dev, x, "count", seg, want_valid=False)  # (1, C) int32
        enough = nvalid.get()[0] >= min_members
        o = out.get()
        o[:, ~enough] = np.nan
        if keep:
            return dev.to_device(o)
        return np.moveaxis(o.reshape((len(values),) + tuple(lead)), 0, -1)
    if keep:
        return out
    return np.moveaxis(out.get().reshape((len(values),) + tuple(lead)), 0, -1)
