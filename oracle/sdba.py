"""Oracle: empirical quantile mapping, group="time".  TEST INFRASTRUCTURE ONLY — **parity unpinned**.

The algorithm lives in the third-party package ``xsdba`` (pinned ``>=0.4.0`` by the reference:
pyproject.toml:111, environment.yml:27; shim src/xclim/sdba.py:1-28).  Its source is not under /root/reference
and it is not installed here, so this is a *specified* restatement of the published algorithm:

* ``equally_spaced_nodes(n)``       q_i = (i + 1/2) / n                                  (xsdba.utils)
* ``nbutils.quantile``              NaN-aware Hyndman-Fan type 7 over the whole series  (same formula as
                                    src/xclim/core/utils.py:395 with alpha = beta = 1), output in the input dtype
* ``get_correction``                af = ref_q - hist_q ("+") | ref_q / hist_q ("*")
* ``interp_on_quantiles`` 1-D       scipy.interpolate.interp1d(hist_q, af, kind, bounds_error=False,
                                    fill_value=(af[0], af[-1]) | nan) over the non-NaN nodes, NaN sim -> NaN
* ``apply_correction``              scen = sim + af_t | sim * af_t

The only test in the reference that pins numbers at this boundary is tests/test_xsdba.py:113-155 (1 decimal).
"""

from __future__ import annotations

import numpy as np
from scipy.interpolate import interp1d

from .quantile import nan_quantile


def equally_spaced_nodes(n: int, eps=None) -> np.ndarray:
    """xsdba.utils.equally_spaced_nodes: linspace(1/2n, 1 - 1/2n, n), plus the end points eps / 1 - eps when given."""
    dq = 1.0 / n / 2.0
    q = np.linspace(dq, 1.0 - dq, n)
    return q if eps is None else np.insert(np.append(q, 1.0 - eps), 0, eps)


def quantile(da, q, axis=0):
    """nbutils.quantile: (nq, ...) in the dtype of `da`."""
    da = np.asarray(da)
    return nan_quantile(da, np.asarray(q, dtype=np.float64), axis=axis, alpha=1.0, beta=1.0).astype(da.dtype)


def eqm_train(ref, hist, nquantiles=20, kind="+"):
    q = equally_spaced_nodes(nquantiles) if np.isscalar(nquantiles) else np.asarray(nquantiles, dtype=np.float64)
    ref_q = quantile(ref, q)
    hist_q = quantile(hist, q)
    with np.errstate(all="ignore"):
        af = ref_q - hist_q if kind == "+" else ref_q / hist_q
    return af, hist_q


def _interp_1d(newx, oldx, oldy, method, extrap):
    mask_new = np.isnan(newx)
    mask_old = np.isnan(oldy) | np.isnan(oldx)
    out = np.full_like(newx, np.nan, dtype=oldy.dtype)
    if mask_new.all() or mask_old.all() or (~mask_old).sum() < 2:
        return out
    fill = (oldy[~mask_old][0], oldy[~mask_old][-1]) if extrap == "constant" else np.nan
    with np.errstate(all="ignore"):
        f = interp1d(oldx[~mask_old], oldy[~mask_old], kind=method, bounds_error=False, fill_value=fill)
        out[~mask_new] = f(newx[~mask_new])
    return out


def eqm_adjust(sim, af, hist_q, kind="+", interp="nearest", extrapolation="constant"):
    """sim (T, C); af, hist_q (nq, C).  Loops over cells like interp_on_quantiles' vectorize=True."""
    sim = np.asarray(sim)
    T = sim.shape[0]
    s2 = sim.reshape(T, -1)
    a2 = np.asarray(af).reshape(af.shape[0], -1)
    h2 = np.asarray(hist_q).reshape(hist_q.shape[0], -1)
    out = np.empty_like(s2)
    for c in range(s2.shape[1]):
        af_t = _interp_1d(s2[:, c], h2[:, c], a2[:, c], interp, extrapolation)
        with np.errstate(all="ignore"):
            out[:, c] = s2[:, c] + af_t if kind == "+" else s2[:, c] * af_t
    return out.reshape(sim.shape)
