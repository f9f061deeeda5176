"""Oracle: ensemble_percentiles (reference: src/xclim/ensembles/_base.py:213-372).  TEST INFRASTRUCTURE ONLY."""

from __future__ import annotations

import numpy as np

from .quantile import calc_perc

_quantile_params = {"interpolated_inverted_cdf": (0, 1), "hazen": (0.5, 0.5), "weibull": (0, 0), "linear": (1, 1),
                    "median_unbiased": (1 / 3, 1 / 3), "normal_unbiased": (3 / 8, 3 / 8)}


def ensemble_percentiles(ens, values=None, min_members=1, method="linear"):
    """_base.py:329-357: calc_perc over the realization axis (axis 0 here), where(valid members >= min_members)."""
    ens = np.asarray(ens)
    if values is None:
        values = [10, 50, 90]
    if min_members is None:
        min_members = ens.shape[0]
    alpha, beta = _quantile_params[method]
    out = calc_perc(np.moveaxis(ens, 0, -1), percentiles=list(values), alpha=alpha, beta=beta)  # (..., nper)
    if min_members != 1:
        ok = (~np.isnan(ens)).sum(axis=0) >= min_members
        out = np.where(ok[..., None], out, np.nan)
    return out
