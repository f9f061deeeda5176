"""Oracle: NaN-aware Hyndman-Fan quantiles (reference: src/xclim/core/utils.py:279-557).  TEST INFRASTRUCTURE ONLY."""

from __future__ import annotations

import numpy as np


def virtual_index(n, q, alpha, beta):
    """utl:370-395 `_compute_virtual_index`: n*q + (alpha + q*(1 - alpha - beta)) - 1, evaluated in that order."""
    return n * q + (alpha + q * (1 - alpha - beta)) - 1


def nan_quantile(arr: np.ndarray, quantiles, axis: int = 0, alpha: float = 1.0, beta: float = 1.0) -> np.ndarray:
    """utl:494-557 `_nan_quantile`.  Returns an array with the quantile axis FIRST, dtype float64 for fp32 input.

    Step by step (numbers = reference lines):
      506-510  empty axis -> NaN; length-1 axis -> the value broadcast over the quantiles
      515-523  n = count of non-NaN along the axis (float); n < 2 -> NaN
      527      vi = virtual_index(n, q)
      417-461  prev = floor(vi), next = prev + 1; vi >= n-1 -> both -1 (last sorted slot); vi < 0 -> both 0;
               vi NaN -> both -1
      538      sort ascending along the axis (NaN last)
      548-549  gamma = vi - prev (with the clipped prev); lerp (464-491): diff = right - left IN THE DATA DTYPE,
               left + diff*gamma, and right - diff*(1-gamma) where gamma >= 0.5
      552-554  NaN result -> nanmax of the slice
    """
    q = np.atleast_1d(np.asarray(quantiles, dtype=np.float64))
    arr = np.asarray(arr)
    L = arr.shape[axis]
    if L == 0:
        return np.nan
    a = np.moveaxis(arr, axis, 0)
    if L == 1:
        return np.broadcast_to(a[0], (q.size,) + a[0].shape).copy()
    a = np.array(a, copy=True)
    if a.ndim == 1:  # 1-D input: run the N-D code on a (L, 1) view and drop the dummy cell axis at the end
        return nan_quantile(a[:, None], q, 0, alpha, beta)[:, 0]
    n = (L - np.isnan(a).sum(axis=0)).astype(np.float64)
    n[n < 2] = np.nan
    n = n[..., None]
    vi = np.asarray(virtual_index(n, q, alpha, beta), dtype=np.float64)  # (..., nq)
    prev = np.floor(vi)
    nxt = prev + 1
    above = vi >= n - 1
    prev[above] = -1
    nxt[above] = -1
    below = vi < 0
    prev[below] = 0
    nxt[below] = 0
    isn = np.isnan(vi)
    prev[isn] = -1
    nxt[isn] = -1
    a.sort(axis=0)
    with np.errstate(invalid="ignore"):
        ip = prev.astype(np.intp)
        inx = nxt.astype(np.intp)
    a = a[..., None]
    left = np.take_along_axis(a, ip[None, ...], axis=0)[0]
    right = np.take_along_axis(a, inx[None, ...], axis=0)[0]
    gamma = vi - prev
    diff = np.subtract(right, left)  # data dtype (fp32 for fp32 input)
    out = np.asarray(left + diff * gamma)
    alt = right - diff * (1 - gamma)
    hi = gamma >= 0.5
    out[hi] = alt[hi]
    with np.errstate(all="ignore"), __import__("warnings").catch_warnings():
        __import__("warnings").simplefilter("ignore", RuntimeWarning)
        amax = np.nanmax(a, axis=0)
    out = np.where(np.isnan(out), amax, out)
    return np.moveaxis(out, -1, 0)


def nan_calc_percentiles(arr, percentiles=None, axis=-1, alpha=1.0, beta=1.0, copy=True):
    """utl:326-367: percentiles/100 -> `nan_quantile`; quantile axis first."""
    per = [50.0] if percentiles is None else percentiles
    q = np.array([p / 100.0 for p in per])
    return nan_quantile(np.asarray(arr), q, axis, alpha, beta)


def calc_perc(arr, percentiles=None, alpha=1.0, beta=1.0, copy=True):
    """utl:279-323: `nan_calc_percentiles` along the last axis with the percentile axis moved LAST."""
    return np.moveaxis(nan_calc_percentiles(arr, percentiles, -1, alpha, beta, copy), 0, -1)
