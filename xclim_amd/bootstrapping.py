"""Host mirror of percentile bootstrapping (reference: src/xclim/core/bootstrapping.py:22-282, Zhang et al. 2005).

The reference deep-copies the base period n-1 times per in-base year (`build_bootstrap_year_da`, :235-282) and re-runs
`percentile_doy` on every copy.  Here the base period stays where it is in HBM: each replica is a *virtual time map*
(`vmap`, int32[T_base]) that redirects the rows of the replaced year to the rows of the source year (with the
reference's 365 <-> 366 length rules expressed as skipped / absent rows), and `xh_percentile_doy_mapped` reads the
samples through it.  The index of the year under study — ANY index decorated with :func:`percentile_bootstrap`: the
tx90p / tx10p family, warm / cold_spell_duration_index, days_over / fraction_over_precip_thresh, as in the reference
(indices/_multivariate.py:68, 1175, 1237, 1299, 1358, 1417, 1476, 1535, 1594, 1718) — is then evaluated against each
replica's percentile table and averaged over the replicas (:203); years outside the base period are evaluated against
the percentile the caller supplied.
"""

from __future__ import annotations

import numpy as np

from . import kernels as K
from ._capi import get_device
from .calendar import DoyPercentile, _flatten, doy_interp_tables
from .timeaxis import TimeAxis, parse_freq, MONTHS


def _get_bootstrap_freq(freq: str) -> str:
    """bootstrapping.py:214-223 for start-anchored offsets: yearly groups with the anchor of `freq`."""
    base, anchor = parse_freq(freq)
    return "YS" if (base == "M" or anchor == 1) else f"YS-{MONTHS[anchor - 1]}"


def _replica_map(nbase: int, bloc: np.ndarray, src: np.ndarray, tbase_axis: TimeAxis) -> np.ndarray:
    """vmap of the base series with rows `bloc` replaced by rows `src` (length rules of bootstrapping.py:253-279)."""
    vmap = np.arange(nbase, dtype=np.int32)
    nb, ns = len(bloc), len(src)
    if ns < 360 and ns < nb:
        return vmap
    if ns == nb:
        vmap[bloc] = src
    elif nb == 365:  # leap source: drop Feb 29 (convert_calendar("noleap"))
        keep = ~((tbase_axis.month[src] == 2) & (tbase_axis.day[src] == 29))
        vmap[bloc] = src[keep]
    elif nb == 366:  # non-leap source: Feb 29 absent (convert_calendar("366_day", missing=nan))
        feb29 = int(np.nonzero((tbase_axis.month[bloc] == 2) & (tbase_axis.day[bloc] == 29))[0][0])
        vals = np.full(366, -1, dtype=np.int32)
        vals[np.arange(366) != feb29] = src
        vmap[bloc] = vals
    elif nb < 365:
        vmap[bloc] = src[:nb]
    else:
        raise NotImplementedError
    return vmap


def _percentile_table(dev, x_base, tb, doys, tbase_axis, window, per, alpha, beta, vmap=None):
    """percentile_doy of the (virtual) base series incl. the drop-366 / re-interpolate step -> ((D, C) table, doys)."""
    p = K.percentile_doy(dev, x_base, tb, window, [per], alpha, beta, vmap=vmap)  # (1, ndoy, C)
    nd, C = p.shape[1], p.shape[2]
    if doys.max() == 366:
        nsrc = int((doys < 366).sum())
        max_t, min_t = int(tbase_axis.doy.max()), int(tbase_axis.doy.min())
        i0, i1, dxn, dxs = doy_interp_tables(nsrc, max_t, min_t)
        src = dev.wrap(p.ptr, (nsrc, C), np.float64)
        src._owner = p
        return K.doy_interp(dev, src, i0, i1, dxn, dxs, xsrc=doys[doys < 366]), np.arange(min_t, max_t + 1)
    return p.reshape(nd, C), doys


def _date_key(y, m, d):
    return np.asarray(y, dtype=np.int64) * 10000 + np.asarray(m, dtype=np.int64) * 100 + np.asarray(d, dtype=np.int64)


def _parse_bound(text, last: bool):
    """"YYYY", "YYYY-MM" or "YYYY-MM-DD" -> date key of the first (last) day it denotes (pandas slice semantics)."""
    parts = [int(v) for v in str(text)[:10].split("-")]
    y = parts[0]
    m = parts[1] if len(parts) > 1 else (12 if last else 1)
    d = parts[2] if len(parts) > 2 else (31 if last else 1)
    return y * 10000 + m * 100 + d


def bootstrap_func(compute_index_func, bound, da_key: str, per_key: str, time_key: str) -> np.ndarray:
    """core/bootstrapping.py:81-211 for ANY index with the signature pieces (da, per: DoyPercentile, time: TimeAxis, freq):

        for every year group g (frequency `_get_bootstrap_freq(freq)`) of `da`
            g inside the percentile base period (`climatology_bounds`, full dates):
                for every other base-period group s: percentile_doy of the base series with g's days replaced by s's
                (a virtual time map, no copy) -> the index of year g against that table; mean over the replicas (:203)
            else: the index of year g against the percentile the CALLER supplied (:196-199)

    The index function does the day-of-year alignment itself (`adjust_doy_calendar` + `resample_doy_index`: a table that
    does not cover a day of the year under study raises instead of reading a wrong row), applies its own time selection
    and MissingAny mask per period, and returns (P_g, *cells); the groups are concatenated along time.
    """
    args = dict(bound.arguments)
    da, per, time, freq = args[da_key], args[per_key], args[time_key], args.get("freq", "YS")
    if not isinstance(per, DoyPercentile) or "percentile_doy" not in per.attrs.get("history", ""):
        raise KeyError("`bootstrap` can only be used with percentiles computed using `percentile_doy`")  # :117-121
    for k in ("climatology_bounds", "window", "alpha", "beta"):
        if k not in per.attrs:
            raise KeyError(f"`bootstrap` can only be used with percentiles computed by percentile_doy (missing attr {k}).")
    if len(per.percentiles) != 1:
        raise ValueError("select one percentile first (DoyPercentile.sel)")
    dev = args.get("device") or get_device()
    x, cell_shape = _flatten(da, dev)
    T, C = x.shape
    if len(time) != T:
        raise ValueError("time axis length does not match the data")
    # overlap of the studied series with the percentile reference period: da.sel(time=slice(*clim)) (:156)
    b0, b1 = per.attrs["climatology_bounds"]
    key = _date_key(time.year, time.month, time.day)
    bidx = np.nonzero((key >= _parse_bound(b0, False)) & (key <= _parse_bound(b1, True)))[0]
    if len(bidx) == T:
        raise KeyError("`bootstrap` is unnecessary when all years are overlapping between reference "
                       "(percentiles period) and studied (index period) periods")
    if len(bidx) == 0:
        raise KeyError("`bootstrap` is unnecessary when no year overlap between reference "
                       "(percentiles period) and studied (index period) periods.")
    if not np.all(np.diff(bidx) == 1):
        raise ValueError("the base period must be a contiguous part of the time axis")
    b0i, nbase = int(bidx[0]), len(bidx)
    x_base = dev.wrap(x.ptr + b0i * C * 4, (nbase, C), np.float32)
    x_base._owner = x
    taxis_b = time.subset(slice(b0i, b0i + nbase))
    tb, years, doys = taxis_b.doy_table()
    bfreq = _get_bootstrap_freq(freq)
    seg_b, starts_b = taxis_b.segments(bfreq)
    seg_a, starts_a = time.segments(bfreq)
    base_labels = set(taxis_b.year.tolist())  # overlap_da.get_index("time").year (:176)
    window, alpha, beta = int(per.attrs["window"]), float(per.attrs["alpha"]), float(per.attrs["beta"])
    pval = float(per.percentiles[0])

    def index(t0, t1, p: DoyPercentile):
        sub = dev.wrap(x.ptr + t0 * C * 4, (t1 - t0, C), np.float32)
        sub._owner = x
        kw = dict(args)
        kw[da_key], kw[per_key], kw[time_key] = sub, p, time.subset(slice(t0, t1))
        out = np.asarray(compute_index_func(**kw), dtype=np.float64)
        return out.reshape(out.shape[0], -1)

    acc = []
    for g, (yg, _) in enumerate(starts_a):
        t0, t1 = int(seg_a[g]), int(seg_a[g + 1])
        if t1 == t0:
            continue
        if yg in base_labels:
            bloc = np.arange(max(t0, b0i), min(t1, b0i + nbase)) - b0i
            tot, nrep = None, 0
            for s_, (ys, _) in enumerate(starts_b):
                if ys == yg:
                    continue
                src = np.arange(int(seg_b[s_]), int(seg_b[s_ + 1]))
                vmap = _replica_map(nbase, bloc, src, taxis_b)
                table, tdoys = _percentile_table(dev, x_base, tb, doys, taxis_b, window, pval, alpha, beta, vmap=vmap)
                rep = DoyPercentile(table.reshape(1, table.shape[0], C), tdoys, [pval], (C,), per.attrs)
                v = index(t0, t1, rep)
                tot = v if tot is None else tot + v
                nrep += 1
            acc.append(tot / nrep)  # .mean(dim=BOOTSTRAP_DIM) (:203)
        else:
            acc.append(index(t0, t1, _flat_percentile(per, C)))
    out = np.concatenate(acc, axis=0)
    return out.reshape((out.shape[0],) + tuple(cell_shape))


def _flat_percentile(per: DoyPercentile, C: int) -> DoyPercentile:
    """The caller's percentile with a flat cell axis (the year views handed to the index are (t, C))."""
    if per.cell_shape == (C,):
        return per
    return DoyPercentile(per.data, per.dayofyear, per.percentiles, (C,), per.attrs)


def percentile_bootstrap(func):
    """core/bootstrapping.py:22-78: decorator adding ``bootstrap=True`` to an index function that takes the data first,
    a :class:`DoyPercentile`, a :class:`TimeAxis` and ``freq`` (in any position: arguments are bound by name, like the
    reference does with ``signature(func).bind``)."""
    import functools
    import inspect

    sig = inspect.signature(func)

    @functools.wraps(func)
    def wrapper(*args, bootstrap: bool = False, **kwargs):
        if not bootstrap:
            return func(*args, **kwargs)
        ba = sig.bind(*args, **kwargs)
        ba.apply_defaults()
        names = list(sig.parameters)
        da_key = names[0]
        time_key = next((n for n, v in ba.arguments.items() if isinstance(v, TimeAxis)), None)
        per_key = next((n for n, v in ba.arguments.items() if isinstance(v, DoyPercentile)), None)
        if per_key is None:
            # per may be a plain array on non day-of-year percentiles
            raise KeyError("`bootstrap` can only be used with percentiles computed using `percentile_doy`")
        if time_key is None:
            raise KeyError("the TimeAxis of the data must be provided")
        return bootstrap_func(func, ba, da_key, per_key, time_key)

    return wrapper
