"""Oracle: ensemble_percentiles (reference: src/xclim/ensembles/_base.py:213-372).  TEST INFRASTRUCTURE ONLY."""

from __future__ import annotations

import numpy as np

from .quantile import calc_perc

_quantile_params = {"interpolated_inverted_cdf": (0, 1), "hazen": (0.5, 0.5), "weibull": (0, 0), "linear": (1, 1),
                    "median_unbiased": (1 / 3, 1 / 3), "normal_unbiased": (3 / 8, 3 / 8)}


def weighted_quantile_1d(data, weights, q):
    """xarray/core/weighted.py `_weighted_quantile_1d` (skipna=True, method "linear" = Hyndman-Fan type 7), the routine
    `ens.weighted(weights).quantile(q, dim="realization")` of _base.py:350-356 ends in.  PARITY UNPINNED: xarray is a
    third-party dependency that is neither under /root/reference nor installed; restated from its published form
    (Akinshin 2023, weighted quantile estimators with Kish's effective sample size)."""
    data = np.asarray(data, dtype=np.float64)
    weights = np.asarray(weights, dtype=np.float64)
    q = np.atleast_1d(np.asarray(q, dtype=np.float64))
    keep = ~np.isnan(data)
    data, weights = data[keep], weights[keep]
    nz = weights != 0
    data, weights = data[nz], weights[nz]
    if data.size == 0:
        return np.full(q.size, np.nan)
    nw = weights.sum() ** 2 / (weights**2).sum()
    sorter = np.argsort(data, kind="stable")
    data, weights = data[sorter], weights[sorter]
    weights = weights / weights.sum()
    weights_cum = np.append(0, weights.cumsum())
    qq = np.atleast_2d(q).T
    h = (nw - 1) * qq + 1
    u = np.maximum((h - 1) / nw, np.minimum(h / nw, weights_cum))
    v = u * nw - h + 1
    w = np.diff(v)
    return (data * w).sum(axis=1)


def ensemble_percentiles(ens, values=None, min_members=1, method="linear", weights=None):
    """_base.py:329-357: calc_perc over the realization axis (axis 0 here), where(valid members >= min_members); with
    `weights` xarray's weighted quantile per element (:346-356)."""
    ens = np.asarray(ens)
    if values is None:
        values = [10, 50, 90]
    if min_members is None:
        min_members = ens.shape[0]
    alpha, beta = _quantile_params[method]
    if weights is not None:
        flat = ens.reshape(ens.shape[0], -1)
        qt = np.array(values) / 100
        res = np.stack([weighted_quantile_1d(flat[:, c], weights, qt) for c in range(flat.shape[1])], axis=0)
        out = res.reshape(ens.shape[1:] + (len(values),))
    else:
        out = calc_perc(np.moveaxis(ens, 0, -1), percentiles=list(values), alpha=alpha, beta=beta)  # (..., nper)
    if min_members != 1:
        ok = (~np.isnan(ens)).sum(axis=0) >= min_members
        out = np.where(ok[..., None], out, np.nan)
    return out
