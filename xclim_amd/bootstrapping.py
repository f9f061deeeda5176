"""Host mirror of percentile bootstrapping (reference: src/xclim/core/bootstrapping.py:22-282, Zhang et al. 2005).

The reference deep-copies the base period n-1 times per in-base year (`build_bootstrap_year_da`, :235-282) and re-runs
`percentile_doy` on every copy.  Here the base period stays where it is in HBM: each replica is a *virtual time map*
(`vmap`, int32[T_base]) that redirects the rows of the replaced year to the rows of the source year (with the
reference's 365 <-> 366 length rules expressed as skipped / absent rows), and `xh_percentile_doy_mapped` reads the
samples through it.  The exceedance count of the year under study is then taken against each replica's percentile
table and averaged over the replicas (:203).
"""

from __future__ import annotations

import numpy as np

from . import kernels as K
from ._capi import get_device
from .calendar import DoyPercentile, _flatten, doy_interp_tables, resample_doy_index
from .generic import get_op
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


def bootstrap_exceedance(da, time: TimeAxis, base_years: tuple[int, int], freq: str, op: str = ">", window: int = 5,
                         per: float = 90.0, alpha: float = 1.0 / 3.0, beta: float = 1.0 / 3.0, *, device=None, floor=None,
                         stat: str = "count") -> np.ndarray:
    """`tx90p(..., bootstrap=True)`-style exceedance count (percentile_bootstrap + bootstrap_func, :22-211).
    With `floor` (the wet-day threshold) the index is days_over_precip_thresh (stat "count") or
    fraction_over_precip_thresh (stat "frac") instead: the percentile is floored by `floor` before the compare.

    `base_years` = (first, last) year of the percentile reference period (the `climatology_bounds` of the reference).
    Returns float64 (P, *cells): averaged counts for in-base years, plain counts elsewhere.
    """
    sym = get_op(op)
    dev = device or get_device()
    x, cell_shape = _flatten(da, dev)
    T, C = x.shape
    y0, y1 = base_years
    in_base = (time.year >= y0) & (time.year <= y1)
    bidx = np.nonzero(in_base)[0]
    if len(bidx) == T:
        raise KeyError("`bootstrap` is unnecessary when all years are overlapping between reference "
                       "(percentiles period) and studied (index period) periods")
    if len(bidx) == 0:
        raise KeyError("`bootstrap` is unnecessary when no year overlap between reference "
                       "(percentiles period) and studied (index period) periods.")
    if not np.all(np.diff(bidx) == 1):
        raise ValueError("the base period must be a contiguous part of the time axis")
    b0, nbase = int(bidx[0]), len(bidx)
    x_base = dev.wrap(x.ptr + b0 * C * 4, (nbase, C), np.float32)
    x_base._owner = x
    taxis_b = time.subset(slice(b0, b0 + nbase))
    tb, years, doys = taxis_b.doy_table()
    bfreq = _get_bootstrap_freq(freq)
    seg_b, starts_b = taxis_b.segments(bfreq)
    seg_a, starts_a = time.segments(bfreq)
    base_labels = set(taxis_b.year.tolist())
    per_table, per_doys = _percentile_table(dev, x_base, tb, doys, taxis_b, window, per, alpha, beta)

    def count(t0, t1, table, tdoys):
        sub = dev.wrap(x.ptr + t0 * C * 4, (t1 - t0, C), np.float32)
        sub._owner = x
        tsub = time.subset(slice(t0, t1))
        seg, _ = tsub.segments(freq)
        pos = np.searchsorted(tdoys, tsub.doy).astype(np.int32)
        if floor is not None:
            cnt, frac, _ = K.precip_over_doy(dev, sub, sym, float(floor), table, pos, seg, want=(stat,), want_valid=False)
            return (cnt if stat == "count" else frac).get().astype(np.float64)
        cnt, _ = K.threshold_count(dev, sub, sym, seg, doy_table=table, tidx=pos, want_valid=False)
        return cnt.get().astype(np.float64)

    acc = []
    for g, (yg, _) in enumerate(starts_a):
        t0, t1 = int(seg_a[g]), int(seg_a[g + 1])
        if t1 == t0:
            continue
        if yg in base_labels:
            bloc = np.arange(max(t0, b0), min(t1, b0 + nbase)) - b0
            vals = []
            for s, (ys, _) in enumerate(starts_b):
                if ys == yg:
                    continue
                src = np.arange(int(seg_b[s]), int(seg_b[s + 1]))
                vmap = _replica_map(nbase, bloc, src, taxis_b)
                table, tdoys = _percentile_table(dev, x_base, tb, doys, taxis_b, window, per, alpha, beta, vmap=vmap)
                vals.append(count(t0, t1, table, tdoys))
            acc.append(np.mean(np.stack(vals, axis=0), axis=0))
        else:
            acc.append(count(t0, t1, per_table, per_doys))
    out = np.concatenate(acc, axis=0)
    return out.reshape((out.shape[0],) + tuple(cell_shape))


def bootstrap_func(compute_index_func, da, per: DoyPercentile, time: TimeAxis, freq: str = "YS", op: str | None = None, *,
                   device=None, thresh=None) -> np.ndarray:
    """bootstrapping.py:81-211 for the percentile-exceedance indices (tx90p / tn10p ... families): the percentile
    reference period, window, alpha and beta are read from the attributes percentile_doy stored (cal:487-494) and the
    index of every in-base year is averaged over the n-1 replicas in which that year is replaced."""
    for k in ("climatology_bounds", "window", "alpha", "beta"):
        if k not in per.attrs:
            raise KeyError(f"`bootstrap` can only be used with percentiles computed by percentile_doy (missing attr {k}).")
    if len(per.percentiles) != 1:
        raise ValueError("select one percentile first (DoyPercentile.sel)")
    b0, b1 = per.attrs["climatology_bounds"]
    if op is None:
        op = getattr(compute_index_func, "_default_op", ">")
    return bootstrap_exceedance(da, time, (int(str(b0)[:4]), int(str(b1)[:4])), freq, op, int(per.attrs["window"]),
                                float(per.percentiles[0]), float(per.attrs["alpha"]), float(per.attrs["beta"]), device=device,
                                floor=thresh, stat=getattr(compute_index_func, "_bootstrap_stat", "count"))


def percentile_bootstrap(func):
    """bootstrapping.py:22-78: decorator adding ``bootstrap=True`` support to an index function with the signature
    ``func(da, per, time, freq=..., op=..., device=...)``."""
    import functools

    @functools.wraps(func)
    def wrapper(da, per, time, freq="YS", *args, bootstrap: bool = False, **kwargs):
        if not bootstrap:
            return func(da, per, time, freq, *args, **kwargs)
        op = kwargs.get("op", args[0] if args else None)
        if not isinstance(per, DoyPercentile):
            raise KeyError("`bootstrap` can only be used with percentiles computed by percentile_doy")  # bootstrapping.py:117-121
        return bootstrap_func(func, da, per, time, freq, op, device=kwargs.get("device"), thresh=kwargs.get("thresh"))

    return wrapper
