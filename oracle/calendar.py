"""Oracle: day-of-year percentiles (reference: src/xclim/core/calendar.py:395-494, 690-790).  TEST INFRASTRUCTURE ONLY."""

from __future__ import annotations

import numpy as np
from scipy.interpolate import interp1d

from .quantile import calc_perc
from .timeutil import OTime

_MAX_DOY = {"standard": 366, "gregorian": 366, "proleptic_gregorian": 366, "noleap": 365, "365_day": 365,
            "360_day": 360, "all_leap": 366, "366_day": 366}


def rolling_construct_center(arr, window):
    """``arr.rolling(time=window, center=True, min_periods=1).construct("window")`` (cal:448): NaN padded windows
    [t - w//2, t + w - 1 - w//2] stacked on a new LAST axis."""
    arr = np.asarray(arr)
    T = arr.shape[0]
    start = window // 2
    pad_shape = (window - 1,) + arr.shape[1:]
    fl = arr.astype(np.result_type(arr.dtype, np.float32), copy=False)
    padded = np.concatenate([np.full((start,) + arr.shape[1:], np.nan, fl.dtype), fl,
                             np.full((pad_shape[0] - start,) + arr.shape[1:], np.nan, fl.dtype)], axis=0)
    return np.stack([padded[k : k + T] for k in range(window)], axis=-1)


def percentile_doy(arr, time: OTime, window=5, per=10.0, alpha=1.0 / 3.0, beta=1.0 / 3.0):
    """cal:395-494.  Returns (p, doys) with p of shape (ndoy, ..., nper) float64 (doy axis first here).

    rolling window -> (year, dayofyear) MultiIndex -> unstack("time") [missing (year, doy) pairs become NaN] ->
    stack(stack_dim=("year", "window")) -> calc_perc over stack_dim; then, when doy 366 exists, drop it and
    re-interpolate 365 -> 366 (cal:484-485).
    """
    arr = np.asarray(arr)
    rr = rolling_construct_center(arr, window)  # (T, ..., w)
    years = np.unique(time.year)
    doys = np.unique(time.doy)
    trailing = rr.shape[1:-1]
    stack = np.full((len(doys), len(years)) + trailing + (window,), np.nan, dtype=rr.dtype)
    yi = np.searchsorted(years, time.year)
    di = np.searchsorted(doys, time.doy)
    stack[di, yi] = rr
    # (doy, year, ..., w) -> (doy, ..., year*w)
    stack = np.moveaxis(stack, 1, -2)
    stack = stack.reshape(stack.shape[:-2] + (len(years) * window,))
    pers = [per] if np.isscalar(per) else list(per)
    p = calc_perc(stack, percentiles=pers, alpha=alpha, beta=beta)  # (doy, ..., nper)
    if doys.max() == 366:
        keep = doys < 366
        p, doys = adjust_doy_calendar(p[keep], doys[keep], time)
    return p, doys


def climatological_mean_doy(arr, time: OTime, window=5):
    """cal:907-931: rolling(center, min_periods=1).construct -> groupby dayofyear -> mean / std over (time, window)."""
    import warnings

    arr = np.asarray(arr)
    rr = rolling_construct_center(arr, window)  # (T, ..., w)
    doys = np.unique(time.doy)
    m = np.empty((len(doys),) + arr.shape[1:], dtype=arr.dtype)
    s = np.empty_like(m)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)
        for i, d in enumerate(doys):
            g = np.moveaxis(rr[time.doy == d], -1, 1)  # (nyears, w, ...)
            g = g.reshape((-1,) + arr.shape[1:]).astype(np.float64)
            m[i] = np.nanmean(g, axis=0)
            s[i] = np.nanstd(g, axis=0)
    return m, s, doys


def interpolate_doy_calendar(source, src_doys, doy_max, doy_min=1):
    """cal:690-726 `_interpolate_doy_calendar`: interpolate_na (linear in the doy coordinate, NaN beyond the outer
    valid points), relabel to linspace(doy_min, doy_max, n), then scipy interp1d (xarray's `interp` path for N-D
    linear) onto the integer doys."""
    source = np.asarray(source, dtype=np.float64)
    n = source.shape[0]
    filled = source.copy()
    flat = filled.reshape(n, -1)
    x = np.asarray(src_doys, dtype=np.float64)
    for j in range(flat.shape[1]):
        col = flat[:, j]
        nans = np.isnan(col)
        if nans.any() and not nans.all():
            col[nans] = np.interp(x[nans], x[~nans], col[~nans], left=np.nan, right=np.nan)
    newx = np.linspace(doy_min, doy_max, n)
    target = np.arange(doy_min, doy_max + 1)
    f = interp1d(newx, filled, kind="linear", axis=0, bounds_error=False, fill_value=np.nan, assume_sorted=True)
    return f(target.astype(np.float64)), target


def adjust_doy_calendar(source, src_doys, target_time: OTime):
    """cal:729-760.  `has_similar_doys` compares bound methods (always False, cal:756) -> only the full-calendar test."""
    max_t, min_t = int(target_time.doy.max()), int(target_time.doy.min())
    if src_doys.max() == _MAX_DOY[target_time.calendar]:
        return source, src_doys
    return interpolate_doy_calendar(source, src_doys, max_t, min_t)


def resample_doy(doy_arr, src_doys, time: OTime):
    """cal:763-790: thresh[t] = adjusted_doy[dayofyear(t)]; NaN where the doy is absent from the table (reindex)."""
    adoy, doys = adjust_doy_calendar(doy_arr, np.asarray(src_doys), time)
    pos = np.searchsorted(doys, time.doy)
    pos_c = np.clip(pos, 0, len(doys) - 1)
    ok = doys[pos_c] == time.doy
    out = np.asarray(adoy)[pos_c].astype(np.float64, copy=True)
    out[~ok] = np.nan
    return out


# ---- select_time (cal:1259-1378), the time selections of the `**indexer` arguments -----------------------------------
def select_time_mask(time: OTime, season=None, month=None, doy_bounds=None, date_bounds=None, include_bounds=(True, True)):
    """Restated with explicit calendar arithmetic: season / month from the month number, doy_bounds on dayofyear
    (wrapping over the year end, cal:1137-1163), date_bounds on the day of year of the all_leap calendar for
    non-uniform calendars (cal:1354-1371) or of the calendar itself for uniform ones."""
    if isinstance(include_bounds, bool):
        include_bounds = (include_bounds, include_bounds)
    mon, day = np.asarray(time.month), np.asarray(time.day)

    def get_doys(start, end):
        d = list(range(start, end + 1)) if start <= end else list(range(start, 367)) + list(range(0, end + 1))
        if not include_bounds[0]:
            d = d[1:]
        if not include_bounds[1]:
            d = d[:-1]
        return d

    if season is not None:
        names = {"DJF": (12, 1, 2), "MAM": (3, 4, 5), "JJA": (6, 7, 8), "SON": (9, 10, 11)}
        months = [m for s_ in ([season] if isinstance(season, str) else season) for m in names[s_]]
        return np.isin(mon, months)
    if month is not None:
        return np.isin(mon, [month] if np.isscalar(month) else list(month))
    if doy_bounds is not None:
        return np.isin(np.asarray(time.doy), get_doys(*doy_bounds))
    (ms, ds), (me, de) = (tuple(int(v) for v in b.split("-")) for b in date_bounds)
    if time.calendar == "360_day":
        cum = [30 * i for i in range(12)]
    elif time.calendar in ("noleap", "365_day"):
        cum = [0, 31, 59, 90, 120, 151, 181, 212, 243, 273, 304, 334]
    else:
        cum = [0, 31, 60, 91, 121, 152, 182, 213, 244, 274, 305, 335]
    cum = np.asarray(cum)
    return np.isin(cum[mon - 1] + day, get_doys(cum[ms - 1] + ds, cum[me - 1] + de))


def mask_between_doys_cells(time: OTime, start, end, cell_shape, include_bounds=(True, True)):
    """cal:1199-1257, bounds WITHOUT a time dimension: (T, *cells) mask; start > end: the span crosses the new year;
    NaN bounds default to 1 / 366."""
    if isinstance(include_bounds, bool):
        include_bounds = (include_bounds, include_bounds)
    s = np.broadcast_to(np.asarray(start, dtype=np.float64), cell_shape).copy()
    e = np.broadcast_to(np.asarray(end, dtype=np.float64), cell_shape).copy()
    if not include_bounds[0]:
        s += 1
    if not include_bounds[1]:
        e -= 1
    s = np.where(np.isnan(s), 1.0, s)
    e = np.where(np.isnan(e), 366.0, e)
    doys = np.asarray(time.doy, dtype=np.float64).reshape((-1,) + (1,) * len(cell_shape))
    return np.where(s <= e, (doys >= s) & (doys <= e), ~((doys > e) & (doys < s)))


def mask_between_doys_temporal(time: OTime, start, end, bounds_labels, freq, cell_shape, include_bounds=(True, True)):
    """cal:1211-1246, bounds WITH a time dimension: `start` / `end` (P_b, *cells), `bounds_labels` = list of (year, month, day)
    of their time coordinate (period starts).  For each period of time.resample(freq): its bounds become days since the
    period's first step (doy_to_days_since, cal:1004-1072: a doy below the base doy lies in the next year), NaN -> 0 /
    366, and a step is selected when start_d <= days <= end_d; periods without bounds are all False (cal:1239-1243)."""
    from .timeutil import groups

    if isinstance(include_bounds, bool):
        include_bounds = (include_bounds, include_bounds)
    nb = len(bounds_labels)
    s = np.broadcast_to(np.asarray(start, dtype=np.float64), (nb,) + tuple(cell_shape)).copy()
    e = np.broadcast_to(np.asarray(end, dtype=np.float64), (nb,) + tuple(cell_shape)).copy()
    if not include_bounds[0]:
        s += 1
    if not include_bounds[1]:
        e -= 1
    labels = [tuple(int(v) for v in lab) for lab in bounds_labels]
    mask = np.zeros((len(time),) + tuple(cell_shape), dtype=bool)

    def daynum(y, m, d):  # consecutive day number in the calendar of `time`
        if time.calendar == "360_day":
            return y * 360 + (m - 1) * 30 + d
        if time.calendar in ("noleap", "365_day"):
            return y * 365 + [0, 31, 59, 90, 120, 151, 181, 212, 243, 273, 304, 334][m - 1] + d
        import datetime

        return datetime.date(int(y), int(m), int(d)).toordinal()

    for label, idx in groups(time, freq):
        if idx.size == 0:
            continue
        lab = (label.year, label.month, label.day) if hasattr(label, "year") else (int(label[0]), int(label[1]), 1)
        if lab not in labels:
            continue
        i = labels.index(lab)
        ly = lab[0]
        base = float(daynum(*lab) - daynum(ly, 1, 1) + 1)          # dayofyear of the period label (base_time)
        if time.calendar == "360_day":
            ylen = 360.0
        elif time.calendar in ("noleap", "365_day"):
            ylen = 365.0
        else:
            ylen = 366.0 if (ly % 4 == 0 and ly % 100 != 0) or ly % 400 == 0 else 365.0
        with np.errstate(invalid="ignore"):
            sd = np.where(s[i] >= base, s[i], s[i] + ylen) - base
            ed = np.where(e[i] >= base, e[i], e[i] + ylen) - base
        sd = np.where(np.isnan(sd), 0.0, sd)
        ed = np.where(np.isnan(ed), 366.0, ed)
        days = np.array([daynum(int(time.year[t]), int(time.month[t]), int(time.day[t])) - daynum(*lab) for t in idx], dtype=np.float64)
        days = days.reshape((-1,) + (1,) * len(cell_shape))
        mask[idx] = (days >= sd) & (days <= ed)
    return mask


def select_time(da, time: OTime, drop=False, **indexer):
    """da.where(mask, drop=drop): NaN outside the selection, or only the selected steps (+ their time axis)."""
    da = np.asarray(da)
    db = indexer.get("doy_bounds")
    if db is not None and not all(isinstance(b, (int, np.integer)) for b in db):
        if drop:
            raise ValueError("Passing array-like doy bounds is incompatible with drop=True.")
        if indexer.get("bounds_labels") is not None:
            m = mask_between_doys_temporal(time, db[0], db[1], indexer["bounds_labels"], indexer["bounds_freq"], da.shape[1:],
                                           indexer.get("include_bounds", (True, True)))
        else:
            m = mask_between_doys_cells(time, db[0], db[1], da.shape[1:], indexer.get("include_bounds", (True, True)))
        out = da.astype(np.result_type(da.dtype, np.float32), copy=True)
        out[~m] = np.nan
        return out
    mask = select_time_mask(time, **indexer)
    if drop:
        idx = np.nonzero(mask)[0]
        return da[idx], time.isel(idx)
    out = da.astype(np.result_type(da.dtype, np.float32), copy=True)
    out[~mask] = np.nan
    return out
