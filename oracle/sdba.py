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

Function by function — what each oracle function restates upstream (``xsdba >= 0.4.0``), so that a maintainer WITH xsdba
can pin it: ``python tests/golden/make_sdba_golden.py`` calls exactly these upstream entry points on seeded inputs and
writes tests/golden/sdba_vectors.npz; tests/test_gpu_sdba_golden.py then holds the HIP path to those numbers (both skip
while the file is absent):

    oracle function                restates (xsdba)                                                  fixture keys
    -----------------------------  ----------------------------------------------------------------  -------------------
    equally_spaced_nodes           xsdba.utils.equally_spaced_nodes(n, eps)                          nodes_20, nodes_eps
    quantile                       xsdba.nbutils.quantile(da, q, dim="time")                         quantile
    eqm_train                      xsdba._adjustment.eqm_train via EmpiricalQuantileMapping.train    eqm_{kind}_af / _hist_q
                                   (ds.af, ds.hist_q; utils.get_correction)
    eqm_adjust / _interp_1d        _adjustment.qm_adjust -> utils.interp_on_quantiles (1-D branch,   eqm_{kind}_{interp}_{extrap}
                                   _interp_on_quantiles_1D: scipy interp1d) + utils.apply_correction
    rank_pct / qdm_adjust          _adjustment.qdm_adjust: utils.rank(sim, dim="time", pct=True) ->  qdm_{kind}_{interp}
                                   interp_on_quantiles(sim_q, quantiles, af) -> apply_correction
    eqm_train_grouped /            the same through base.Grouper("time.month" | "time.dayofyear",   eqmg_{group}_af / _hist_q /
      eqm_adjust_grouped           window): Grouper.apply / map_groups sample sets; adjust with      _scen  (interp "nearest")
                                   interp="nearest" (2-D griddata "nearest", see xclim_amd/sdba.py)
    dqm_train / dqm_adjust /       _adjustment.dqm_train / dqm_adjust, detrending.PolyDetrend        dqm_{kind}_af / _scaling /
      poly_trend                   (degree 0 / 1), utils.apply_correction / invert                   _scen_d{degree}

    adapt_freq / adapt_uniform     processing.adapt_freq -> _processing._adapt_freq (ecdf,           adapt_{group}_sim_ad / _pth /
                                   nbutils.vecquantiles, DataArray.rank(pct=True)); the random       _dP0 (sim_ad only where no
                                   fill values come from a counter-based uniform, not from numpy's   value was replaced: upstream's
                                   global generator                                                  draws are not reproducible)

    dqm_train_grouped /            dqm_train / dqm_adjust through Grouper("time.month" | "time.season")  dqmg_{group}_af / _hist_q /
      dqm_adjust_grouped           (window 1): u.broadcast of the scaling, PolyDetrend(group=...)         _scaling / _scen

    interp_on_quantiles_2d /       utils.interp_on_quantiles, 2-D branch (add_cyclic_bounds, _interp_on_quantiles_2D =     eqmg_{group}_scen_linear,
      group_index                  scipy griddata "nearest" | "linear" — the REAL scipy routine —, _extrapolate_on_quantiles);  qdmg_{group}_scen_linear
                                   Grouper.get_index(interp=True): fractional months, integer days of year

    NOT restated (refused by the product): interpolation over (quantile, group) for "cubic" (griddata's Clough-Tocher
    scheme), "linear" with extrapolation="nan" or a season grouping, DQM with a windowed sub-grouping (the trend is fitted
    on the window mean there), QDM cubic.

Where a difference is most likely once the fixtures exist (from memory of the upstream sources, not verified here): (1)
``nbutils.quantile`` casts the probabilities to the dtype of the data before it calls numpy's nanquantile, so float32
series see float32 probabilities and numpy's float32 interpolation arithmetic — this restatement (and the kernels) keep
the probabilities and the interpolation weight in float64 and round once: differences of the order of 1e-7 relative are
expected, inside the 1e-6 bar; (2) grouped adjustments with "nearest": upstream's 2-D ``griddata(method="nearest")`` measures distance in (value, group
index) space, so a node of a NEIGHBOURING group can be the nearest one where the quantile values of a group are more than
one unit apart — since round 4 restated with the real scipy routine (``interp_on_quantiles_2d_nearest``, mode
"griddata"; the own-group rule of rounds 2-3 is mode "group"); what remains from memory there: that upstream passes the
INTEGER group coordinate for "nearest" (``group.get_index(newx, interp=False)``) and extrapolates with the own group.
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


def quantile(da, q, axis=0, mode="numpy"):
    """nbutils.quantile: (nq, ...) in the dtype of `da`.  mode "numpy" (the contract of this backend): the reference's own
    Hyndman-Fan code (core/utils.py:370-557, numpy's ``_lerp``).  mode "numba": the arithmetic numba's ``np.nanquantile``
    uses inside xsdba's jitted ``nbutils._quantile`` — see :func:`quantile_numba`."""
    da = np.asarray(da)
    if mode == "numba":
        return quantile_numba(da, q, axis)
    return nan_quantile(da, np.asarray(q, dtype=np.float64), axis=axis, alpha=1.0, beta=1.0).astype(da.dtype)


def quantile_numba(da, q, axis=0):
    """What xsdba's ``nbutils._quantile`` computes when it is compiled: numba's implementation of ``np.nanquantile``
    (numba/np/arraymath.py, ``_collect_percentiles_inner``; from memory of the upstream source, not verified here —
    VERDICT r4 weak #1).  Per series, on the n non-NaN samples, for a probability q (as percentile p = 100 q):

        q == 0 / q == 1      the minimum / maximum
        otherwise            rank = 1 + (n - 1) * (p / 100);  f = floor(rank);  m = rank - f
                             lower, upper = the (f - 1)-th and f-th order statistic (0-based; upper clipped to the last)
                             value = lower * (1 - m) + upper * m          in float64, stored in the dtype of the data

    against the numpy mode's ``left + (right - left) * gamma`` with ``gamma`` from the virtual index ``n q + (1 - q) - 1``
    and the float32 difference.  Same order statistics, two roundings of the weights: the results differ in the last
    float32 ulps (tests/test_oracle_golden.py::test_numba_quantile_mode_stays_within_the_bar bounds it)."""
    da = np.asarray(da)
    q = np.atleast_1d(np.asarray(q, dtype=np.float64))
    a = np.moveaxis(da, axis, 0)
    flat = a.reshape(a.shape[0], -1)
    out = np.full((q.size, flat.shape[1]), np.nan, dtype=np.float64)
    for c in range(flat.shape[1]):
        v = np.sort(flat[:, c][~np.isnan(flat[:, c])])
        n = v.size
        if n == 0:
            continue
        for i, qi in enumerate(q):
            p = qi * 100.0
            if n == 1:
                out[i, c] = v[0]
            elif p == 100.0:
                out[i, c] = v[-1]
            elif p == 0.0:
                out[i, c] = v[0]
            else:
                rank = 1.0 + (n - 1) * (p / 100.0)
                f = np.floor(rank)
                m = rank - f
                lo = float(v[int(f) - 1])
                hi = float(v[min(int(f), n - 1)])
                out[i, c] = lo * (1.0 - m) + hi * m
    return out.reshape((q.size,) + a.shape[1:]).astype(da.dtype)


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
    out = np.full(np.shape(newx), np.nan, dtype=oldy.dtype)
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


# ---- QuantileDeltaMapping (xsdba._adjustment.qdm_adjust, group "time") — specified restatement, parity unpinned ---------
def rank_pct(x):
    """xsdba.utils.rank(da, dim="time", pct=True) for one series: average ranks of the valid samples divided by their
    count (xarray's ``rank(pct=True)``), then rescaled ``mx * (rnk - mn) / (mx - mn)`` so that the smallest sample gets 0
    (xarray's own percentage ranks start at 1 / n); NaN stays NaN."""
    from scipy.stats import rankdata

    x = np.asarray(x)
    out = np.full(x.shape, np.nan, dtype=np.float64)
    ok = ~np.isnan(x)
    n = int(ok.sum())
    if n == 0:
        return out
    rnk = rankdata(x[ok], method="average") / n
    mn, mx = rnk.min(), rnk.max()
    with np.errstate(all="ignore"):
        out[ok] = mx * (rnk - mn) / (mx - mn)
    return out


def qdm_adjust(sim, af, quantiles, kind="+", interp="nearest", extrapolation="constant"):
    """sim (T, C); af (nq, C); quantiles (nq,) — the abscissa of the factors is the quantile node, the same for every
    cell: sim_q = rank(sim, pct=True); af_t = interp_on_quantiles(sim_q, quantiles, af); scen = sim (+|*) af_t."""
    sim = np.asarray(sim)
    T = sim.shape[0]
    s2 = sim.reshape(T, -1)
    a2 = np.asarray(af).reshape(af.shape[0], -1)
    xq = np.asarray(quantiles, dtype=np.float64)
    out = np.empty_like(s2)
    for c in range(s2.shape[1]):
        af_t = _interp_1d(rank_pct(s2[:, c]), xq, a2[:, c], interp, extrapolation)
        with np.errstate(all="ignore"):
            out[:, c] = s2[:, c] + af_t if kind == "+" else s2[:, c] * af_t
    return out.reshape(sim.shape)


# ---- sub-groupings (xsdba.base.Grouper: "time.month", "time.dayofyear" with a window) — specified restatement ----------
def group_values(time, prop):
    if prop == "season":  # time.dt.season
        return np.array(["", "DJF", "DJF", "MAM", "MAM", "MAM", "JJA", "JJA", "JJA", "SON", "SON", "SON", "DJF"])[np.asarray(time.month)]
    return np.asarray(time.month if prop == "month" else time.doy)


def grouped_sample(x, time, prop, window, label):
    """Training sample of one group: the centred `window` steps around every time step of the group (NaN beyond the ends
    of the series), flattened — rolling(time=window, center=True).construct("window") + groupby(prop) in xsdba."""
    x = np.asarray(x)
    T = x.shape[0]
    t = np.nonzero(group_values(time, prop) == label)[0]
    half = window // 2
    rows = (t[:, None] + np.arange(-half, half + 1)[None, :]).reshape(-1)
    ok = (rows >= 0) & (rows < T)
    out = np.full((len(rows),) + x.shape[1:], np.nan, dtype=x.dtype)
    out[ok] = x[rows[ok]]
    return out


def eqm_train_grouped(ref, hist, time, prop, window=1, nquantiles=20, kind="+"):
    """(af, hist_q, labels) with shapes (G, nq, ...)."""
    q = equally_spaced_nodes(nquantiles) if np.isscalar(nquantiles) else np.asarray(nquantiles, dtype=np.float64)
    labels = np.unique(group_values(time, prop))
    afs, hqs = [], []
    for lab in labels:
        a, h = eqm_train(grouped_sample(ref, time, prop, window, lab), grouped_sample(hist, time, prop, window, lab), q, kind)
        afs.append(a)
        hqs.append(h)
    return np.stack(afs), np.stack(hqs), labels


def eqm_adjust_grouped(sim, time, prop, labels, af, hist_q, kind="+", interp="nearest", extrapolation="constant", mode="group"):
    """mode "group": every time step is mapped with the nearest node of its own group (rounds 2-3).  mode "griddata":
    utils.interp_on_quantiles with a sub-grouping as upstream does it (see interp_on_quantiles_2d_nearest)."""
    sim = np.asarray(sim)
    out = np.empty_like(sim)
    gv = group_values(time, prop)
    if mode == "griddata":
        newg = gv if interp == "nearest" else group_index(time, prop, True)
        af_t = interp_on_quantiles_2d(sim, newg, labels, hist_q, af, interp, extrapolation)
        with np.errstate(all="ignore"):
            return (sim + af_t if kind == "+" else sim * af_t).astype(sim.dtype)
    if interp != "nearest":
        raise ValueError("mode='group' is the own-group NEAREST rule; linear goes through mode='griddata'")
    for g, lab in enumerate(labels):
        rows = np.nonzero(gv == lab)[0]
        if rows.size:
            out[rows] = eqm_adjust(sim[rows], af[g], hist_q[g], kind, interp, extrapolation)
    return out


def group_index(time, prop, interp):
    """xsdba.base.Grouper.get_index(da, interp=...): the group coordinate of every time step.  Integer month / day of year;
    with ``interp=True`` (every method but "nearest") the month becomes FRACTIONAL — ``month - 0.5 + day / days_in_month``
    (the middle of a month sits on its integer) — and the day of year stays an integer."""
    if prop == "dayofyear":
        return np.asarray(time.doy, dtype=np.float64)
    if prop != "month":
        raise NotImplementedError(prop)
    month = np.asarray(time.month)
    if not interp:
        return month.astype(np.float64)
    if time.index is not None:
        dim = np.asarray(time.index.days_in_month)
    elif time.calendar == "360_day":
        dim = np.full(len(month), 30)
    else:
        dim = np.array([31, 28, 31, 30, 31, 30, 31, 31, 30, 31, 30, 31])[month - 1]
    return month - 0.5 + np.asarray(time.day) / dim


def interp_on_quantiles_2d(newx, newg, labels, xq, yq, method="nearest", extrapolation="constant"):
    """xsdba.utils.interp_on_quantiles for a month / day-of-year grouping, per cell:
    add_cyclic_bounds (the last group copied to coordinate 0, the first to G + 1; labels 1 .. G) ->
    _interp_on_quantiles_2D = scipy.interpolate.griddata((oldx, oldg), oldy, (newx, newg), method=method) on the
    non-NaN nodes of ALL groups — the REAL scipy routine, no rescaling: "nearest" = a cKDTree query in the (value, group)
    plane, "linear" = barycentric interpolation on Qhull's Delaunay triangulation of the nodes (NaN outside their convex
    hull) -> _extrapolate_on_quantiles (for "nearest" always, otherwise unless extrapolation="nan"): where newx lies outside
    the first / last non-null node — np.interp(newg, rows, bounds): of ITS OWN group for an integer newg, between the two
    neighbouring groups for a fractional one — that bound's first / last factor ("constant") or NaN.
    newx (T, ...), newg (T,) group coordinate per step (group_index), xq / yq (G, nq, ...); xq may be (nq,) (QDM: the
    quantile nodes themselves are the abscissa, the same for every group and cell)."""
    from scipy.interpolate import griddata

    newx = np.asarray(newx)
    shape = newx.shape
    x2 = newx.reshape(shape[0], -1)
    yq = np.asarray(yq)
    G, nq = yq.shape[:2]
    yq2 = np.asarray(yq, dtype=np.float64).reshape(G, nq, -1)
    xq = np.asarray(xq, dtype=np.float64)
    xq2 = np.broadcast_to(xq[None, :, None], yq2.shape) if xq.ndim == 1 else xq.reshape(G, nq, -1)
    assert np.array_equal(labels, np.arange(1, G + 1))
    ext = np.concatenate([[G - 1], np.arange(G), [0]])          # rows of the padded tables: coordinates 0 .. G + 1
    oldg = np.repeat(np.arange(G + 2, dtype=np.float64)[:, None], nq, axis=1)
    rows = np.arange(G + 2, dtype=np.float64)
    out = np.full(x2.shape, np.nan)
    g = np.asarray(newg, dtype=np.float64)

    def first_last(a):  # utils._first_and_last_nonnull: per group row, on THIS array alone (x and y independently)
        res = np.full((a.shape[0], 2), np.nan)
        for r, row in enumerate(a):
            ok = np.nonzero(~np.isnan(row))[0]
            if ok.size:
                res[r] = row[ok[0]], row[ok[-1]]
        return res

    for c in range(x2.shape[1]):
        oldx, oldy = xq2[ext, :, c], yq2[ext, :, c]
        x = x2[:, c].astype(np.float64)
        m_new, m_old = np.isnan(x), np.isnan(oldx) | np.isnan(oldy)
        if m_new.all() or m_old.all():
            continue
        res = np.full(x.shape, np.nan)
        res[~m_new] = griddata((oldx[~m_old], oldg[~m_old]), oldy[~m_old], (x[~m_new], g[~m_new]), method=method)
        if method == "nearest" or extrapolation != "nan":       # _extrapolate_on_quantiles
            bx, by = first_last(oldx), first_last(oldy)
            with np.errstate(invalid="ignore"):
                toolow, toohigh = x < np.interp(g, rows, bx[:, 0]), x > np.interp(g, rows, bx[:, 1])
            if extrapolation == "constant":
                res[toolow], res[toohigh] = np.interp(g, rows, by[:, 0])[toolow], np.interp(g, rows, by[:, 1])[toohigh]
            else:
                res[toolow | toohigh] = np.nan
        out[:, c] = res
    return out.reshape(shape).astype(np.float32)


def interp_on_quantiles_2d_nearest(newx, newg, labels, xq, yq, extrapolation="constant"):
    return interp_on_quantiles_2d(newx, newg, labels, xq, yq, "nearest", extrapolation)


def qdm_adjust_grouped(sim, time, prop, labels, af, quantiles, kind="+", interp="nearest", extrapolation="constant", mode="group"):
    """Grouped QDM: ranks inside each group's own time steps (main_only=True), factors of that group ("group"); mode
    "griddata": the factors interpolated over the (quantile, group) plane as xsdba does for interp != "nearest" —
    interp_on_quantiles(sim_q, quantiles, af) with the quantile nodes themselves as abscissa in every group."""
    sim = np.asarray(sim)
    out = np.empty_like(sim)
    gv = group_values(time, prop)
    if mode == "griddata":
        T = sim.shape[0]
        s2 = sim.reshape(T, -1)
        sim_q = np.full(s2.shape, np.nan)
        for lab in labels:
            rows = np.nonzero(gv == lab)[0]
            for c in range(s2.shape[1]):
                sim_q[rows, c] = rank_pct(s2[rows, c])
        af_t = interp_on_quantiles_2d(sim_q.reshape(sim.shape), group_index(time, prop, interp != "nearest"), labels,
                                      np.asarray(quantiles, dtype=np.float64), af, interp, extrapolation)
        with np.errstate(all="ignore"):
            return (sim + af_t if kind == "+" else sim * af_t).astype(sim.dtype)
    for g, lab in enumerate(labels):
        rows = np.nonzero(gv == lab)[0]
        if rows.size:
            out[rows] = qdm_adjust(sim[rows], af[g], quantiles, kind, interp, extrapolation)
    return out


# ---- DetrendedQuantileMapping (xsdba._adjustment.dqm_train / dqm_adjust, PolyDetrend) — specified restatement, unpinned ------
def _nanmean0(x):
    import warnings

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        return np.nanmean(np.asarray(x, dtype=np.float64), axis=0)


def _corr(x, f, kind, inverse=False):
    """apply_correction(x, f | invert(f)): float64 arithmetic, float32 result (the device's stage boundaries)."""
    x64 = np.asarray(x, dtype=np.float64)
    with np.errstate(all="ignore"):
        if kind == "+":
            out = x64 - f if inverse else x64 + f
        else:
            out = x64 / f if inverse else x64 * f
    return out.astype(np.float32)


def dqm_train(ref, hist, nquantiles=20, kind="+"):
    """(af, hist_q, scaling): quantiles of the mean-normalised series, scaling = mean(ref) (-|/) mean(hist)."""
    mu_r, mu_h = _nanmean0(ref), _nanmean0(hist)
    af, hist_q = eqm_train(_corr(ref, mu_r, kind, True), _corr(hist, mu_h, kind, True), nquantiles, kind)
    with np.errstate(all="ignore"):
        scaling = mu_r - mu_h if kind == "+" else mu_r / mu_h
    return af, hist_q, scaling


def poly_trend(x, degree):
    """Per-cell least-squares polynomial (degree 0 / 1) over the valid steps, evaluated at every step: (T, C) float64."""
    x = np.asarray(x, dtype=np.float64)
    T = x.shape[0]
    x2 = x.reshape(T, -1)
    t = np.arange(T, dtype=np.float64) - 0.5 * (T - 1)
    out = np.full(x2.shape, np.nan)
    for c in range(x2.shape[1]):
        ok = ~np.isnan(x2[:, c])
        if not ok.any():
            continue
        if degree == 0 or ok.sum() < 2:
            out[:, c] = x2[ok, c].mean()
        else:
            out[:, c] = np.polyval(np.polyfit(t[ok], x2[ok, c], 1), t)
    return out.reshape(x.shape)


def dqm_adjust(sim, af, hist_q, scaling, kind="+", interp="nearest", extrapolation="constant", detrend=1):
    scaled = _corr(sim, scaling, kind)
    trend = poly_trend(scaled, detrend)
    detr = _corr(scaled, trend, kind, True)
    scen0 = eqm_adjust(detr, af, hist_q, kind, interp, extrapolation)
    return _corr(scen0, trend, kind)


def dqm_adjust_members(sim, af, hist_q, scaling, kind="+", interp="nearest", extrapolation="constant", detrend=1, pooled=True):
    """dqm_adjust for a sim with a member axis behind time ((T, R, *cells)); af / hist_q / scaling have none.  pooled=True — the model
    was trained with Grouper(add_dims=[member]): PolyDetrend(group=that grouper) fits ONE trend on the mean over the members
    (_polydetrend_get_trend: ``if len(dim) > 1: da = da.mean(dim[1:])``) and every member is detrended with it; pooled=False: the
    member axis is an ordinary one, every series has its own trend."""
    sim = np.asarray(sim)
    scaled = _corr(sim, np.asarray(scaling)[None, None], kind)
    if pooled:
        import warnings

        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            mean = np.nanmean(scaled.astype(np.float64), axis=1).astype(np.float32)
        trend = np.broadcast_to(poly_trend(mean, detrend)[:, None], sim.shape)
    else:
        trend = poly_trend(scaled, detrend)
    detr = _corr(scaled, trend, kind, True)
    scen0 = np.stack([eqm_adjust(detr[:, r], af, hist_q, kind, interp, extrapolation) for r in range(sim.shape[1])], axis=1)
    return _corr(scen0, trend, kind)


def poly_trend_u(x, u, degree):
    """The same on an explicit coordinate `u` (one value per row): DataArray.polyfit over the time coordinate of a
    group's steps; evaluated at the same rows."""
    x = np.asarray(x, dtype=np.float64)
    x2 = x.reshape(x.shape[0], -1)
    u = np.asarray(u, dtype=np.float64)
    out = np.full(x2.shape, np.nan)
    for c in range(x2.shape[1]):
        ok = ~np.isnan(x2[:, c])
        if not ok.any():
            continue
        if degree == 0 or ok.sum() < 2 or np.ptp(u[ok]) == 0:
            out[:, c] = x2[ok, c].mean()
        else:
            out[:, c] = np.polyval(np.polyfit(u[ok], x2[ok, c], 1), u)
    return out.reshape(x.shape)


def dqm_train_grouped(ref, hist, time, prop, nquantiles=20, kind="+", window=1):
    """dqm_train per group on its (windowed) sample — the means run over time AND window (``ds.ref.mean(dim)`` with
    dim = [time, window]): (labels, af (G, nq, ...), hist_q, scaling (G, ...))."""
    labels = np.unique(group_values(time, prop))
    res = [dqm_train(grouped_sample(ref, time, prop, window, lab), grouped_sample(hist, time, prop, window, lab), nquantiles, kind)
           for lab in labels]
    return labels, np.stack([r[0] for r in res]), np.stack([r[1] for r in res]), np.stack([r[2] for r in res])


def window_nanmean(x, window):
    """rolling(time=window, center=True).construct("window").mean("window"): the mean over the valid samples of the centred
    window (rows beyond the ends of the series are NaN padding), float64 sums, float32 result (a stage boundary)."""
    import warnings

    x = np.asarray(x)
    T, half = x.shape[0], window // 2
    pad = np.full((half,) + x.shape[1:], np.nan, dtype=np.float64)
    xp = np.concatenate([pad, x.astype(np.float64), pad])
    stack = np.stack([xp[k:k + T] for k in range(window)])
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        return np.nanmean(stack, axis=0).astype(np.float32)


def dqm_adjust_grouped(sim, time, prop, labels, af, hist_q, scaling, kind="+", extrapolation="constant", detrend=1, mode="group",
                       interp="nearest", window=1):
    """dqm_adjust with a sub-grouping, interp="nearest": every step takes the scaling of its group (u.broadcast), the
    polynomial trend is fitted PER GROUP over the group's own steps on the time coordinate (PolyDetrend(group=...):
    polyfit along time — here days since the group's mean date; a linear fit does not depend on the origin), the
    detrended steps go through their group's nodes, the trend is put back."""
    sim = np.asarray(sim)
    out = np.full(sim.shape, np.nan, dtype=np.float32)
    gv = group_values(time, prop)
    days = np.arange(sim.shape[0], dtype=np.float64)  # (a DAILY series: the day number up to an origin)
    scaled_all = None
    if interp != "nearest" and prop != "dayofyear":
        # u.broadcast(scaling, sim, group=group, interp=interp): the scaling interpolated over the group coordinate
        # (add_cyclic_bounds, DataArray.interp "linear"); "nearest" and day-of-year groupings take the group's own value
        G = len(labels)
        gc = group_index(time, prop, True)
        ext = np.concatenate([[G - 1], np.arange(G), [0]])
        sc = np.asarray(scaling, dtype=np.float64).reshape(G, -1)[ext]
        r0 = np.clip(np.floor(gc).astype(int), 0, G)
        f = (gc - r0)[:, None]
        sc_t = (sc[r0] + (sc[r0 + 1] - sc[r0]) * f).reshape(sim.shape)
        scaled_all = _corr(sim, sc_t, kind)
    wmean = None
    if window > 1:
        # PolyDetrend with a windowed Grouper (xsdba.detrending._polydetrend_get_trend: ``if len(dim) > 1: da = da.mean(dim[1:])``
        # ahead of polyfit): the trend of a group is fitted on the window mean of the SCALED series at the group's steps
        if scaled_all is None:
            gidx = np.searchsorted(labels, gv)
            scaled_all = _corr(sim, np.asarray(scaling, dtype=np.float64).reshape((len(labels),) + sim.shape[1:])[gidx], kind)
        wmean = window_nanmean(scaled_all, window)
    detr_all = np.full(sim.shape, np.nan, dtype=np.float32)
    trend_all = np.full(sim.shape, np.nan, dtype=np.float64)
    for g, lab in enumerate(labels):
        rows = np.nonzero(gv == lab)[0]
        if not rows.size:
            continue
        scaled = _corr(sim[rows], scaling[g], kind) if scaled_all is None else scaled_all[rows]
        u = days[rows] - days[rows].mean()
        trend = poly_trend_u(scaled if wmean is None else wmean[rows], u, detrend)
        detr = _corr(scaled, trend, kind, True)
        if mode == "griddata":
            detr_all[rows], trend_all[rows] = detr, trend
        else:
            out[rows] = _corr(eqm_adjust(detr, af[g], hist_q[g], kind, "nearest", extrapolation), trend, kind)
    if mode == "griddata":
        # one interpolation over the (quantile, group) plane for the whole series (the same per step as group by group; a
        # triangulation per group and cell made the day-of-year case take minutes)
        newg = gv if interp == "nearest" else group_index(time, prop, True)
        af_t = interp_on_quantiles_2d(detr_all, newg, labels, hist_q, af, interp, extrapolation)
        with np.errstate(all="ignore"):
            scen0 = (detr_all + af_t if kind == "+" else detr_all * af_t).astype(np.float32)
        out = _corr(scen0, trend_all, kind)
    return out


# ---- adapt_freq (xsdba.processing.adapt_freq -> _processing._adapt_freq) ------------------------------------------------
_K0, _K1, _K2 = np.uint64(0xD1342543DE82EF95), np.uint64(0x9E3779B97F4A7C15), np.uint64(0xC2B2AE3D27D4EB4F)


def adapt_uniform(seed, tindex, cells):
    """The uniform numbers of xh_adapt_freq (xclim_amd/csrc/qdm3.hip k_q3_adapt), bit for bit: U(seed, t, cell) =
    (mix64(seed K0 + t K1 + cell K2) >> 11) / 2^53, float64 in [0, 1).  Stands where upstream calls
    ``np.random.random_sample(size=sim.shape)``."""
    from .synth import _mix64

    t = np.asarray(tindex, dtype=np.uint64)[:, None]
    c = np.asarray(cells, dtype=np.uint64)[None, :]
    with np.errstate(over="ignore"):
        z = _mix64(np.uint64(seed) * _K0 + t * _K1 + c * _K2)
    return (z >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def rank_avg_pct(x):
    """DataArray.rank(dim, pct=True) along axis 0: average ranks of the valid samples (bottleneck.nanrankdata) / their
    count; NaN stays NaN.  (utils.rank, used by QDM, rescales on top of this: rank_pct above.)"""
    from scipy.stats import rankdata

    x = np.asarray(x)
    out = np.full(x.shape, np.nan)
    for c in range(x.shape[1]):
        ok = ~np.isnan(x[:, c])
        n = int(ok.sum())
        if n:
            out[ok, c] = rankdata(x[ok, c] + x.dtype.type(0), method="average") / n   # (+0: the two zeros tie)
    return out


def _adapt_one(ref_sample, sim_sample, sim_main, thresh, u):
    """One group: samples (N, C) for the frequencies and pth, the group's own steps (n, C) for the ranks; u (n, C)."""
    f32 = np.float32
    with np.errstate(all="ignore"):
        p0_sim = (sim_sample <= f32(thresh)).sum(axis=0) / (~np.isnan(sim_sample)).sum(axis=0)      # utils.ecdf
        p0_ref = (ref_sample <= f32(thresh)).sum(axis=0) / (~np.isnan(ref_sample)).sum(axis=0)
        dp0 = (p0_sim - p0_ref) / p0_sim
    pth = np.full(ref_sample.shape[1], np.nan, f32)                                                   # nbutils.vecquantiles
    for c in range(ref_sample.shape[1]):
        if dp0[c] > 0:
            pth[c] = nan_quantile(ref_sample[:, c], np.array([p0_sim[c]]), axis=0, alpha=1.0, beta=1.0)[0].astype(f32)
    rank = rank_avg_pct(sim_main)
    with np.errstate(invalid="ignore"):
        keep = (rank < p0_ref[None]) | (rank > p0_sim[None]) | np.isnan(sim_main)
        fill = (pth[None].astype(np.float64) - float(thresh)) * u + float(thresh)
        inner = np.where(keep, sim_main, fill.astype(f32))
        sim_ad = np.where((dp0 < 0)[None], sim_main, inner).astype(f32)
    return sim_ad, pth, dp0


def adapt_freq(ref, sim, thresh, seed=0, time=None, prop="group", window=1, cell0=0):
    """sim_ad (T, C) float32, pth, dP0 — group "time" (prop "group") or a sub-grouping on the common `time` (OTime)."""
    ref, sim = np.asarray(ref), np.asarray(sim)
    T, C = sim.shape
    cells = cell0 + np.arange(C)
    if prop == "group":
        return _adapt_one(ref, sim, sim, thresh, adapt_uniform(seed, np.arange(T), cells))
    vals = group_values(time, prop)
    labels = np.unique(vals)
    out = sim.copy()
    pths, dp0s = [], []
    for lab in labels:
        main = np.nonzero(vals == lab)[0]
        ad, pth, dp0 = _adapt_one(grouped_sample(ref, time, prop, window, lab), grouped_sample(sim, time, prop, window, lab), sim[main],
                                  thresh, adapt_uniform(seed, main, cells))
        out[main] = ad
        pths.append(pth)
        dp0s.append(dp0)
    return out, np.stack(pths), np.stack(dp0s)
