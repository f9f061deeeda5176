"""Oracle: run-length algorithms (reference: src/xclim/indices/run_length.py).  TEST INFRASTRUCTURE ONLY.

numpy restatement with time on axis 0.  xarray semantics restated explicitly: ``where`` promotes small unsigned
ints to float32 (float64 from uint32 up), ``shift(fill_value=0)``, ``resample(time=freq).map``.
"""

from __future__ import annotations

import warnings

import numpy as np

from .timeutil import OTime, groups


def _smallest_uint(T: int):
    """rl:135-139."""
    for dt in (np.uint8, np.uint16, np.uint32, np.uint64):
        if np.iinfo(dt).max > T:
            return dt
    return np.uint64


def _promote(dtype):
    """xarray dtypes.maybe_promote for the dtypes that occur here."""
    dtype = np.dtype(dtype)
    if dtype.kind == "f":
        return dtype
    if dtype.kind == "b":
        return np.dtype(object)  # never needed: bool arrays are converted before `where`
    return np.dtype(np.float32) if dtype.itemsize <= 2 else np.dtype(np.float64)


def where_nan(a, cond):
    """``a.where(cond)``: NaN where cond is False, with xarray's dtype promotion."""
    a = np.asarray(a)
    out = a.astype(_promote(a.dtype), copy=True)
    out[~np.broadcast_to(cond, out.shape)] = np.nan
    return out


def shift0(a, n, fill):
    """``a.shift(time=n, fill_value=fill)`` along axis 0 (n > 0 moves data towards later times)."""
    a = np.asarray(a)
    out = np.full_like(a, fill)
    if n == 0:
        return a.copy()
    if abs(n) >= a.shape[0]:
        return out
    if n > 0:
        out[n:] = a[:-n]
    else:
        out[:n] = a[-n:]
    return out


def cumsum_reset_np(arr, index, one):
    """rl:143-151 `_cumsum_reset_np` with the core dim on axis 0 instead of -1.  100110111 -> 100120123."""
    T = arr.shape[0]
    it = range(1, T) if index == "last" else range(T - 2, -1, -1)
    step = 1 if index == "last" else -1
    for i in it:
        arr[i] *= arr[i - step] + one
    return arr


def cumsum_reset(da, index="last"):
    """rl:172-219 `_cumsum_reset` fast track: NaN -> 0, then the njit recurrence."""
    da = np.asarray(da)
    typ = _smallest_uint(da.shape[0])
    if da.dtype.kind == "f":
        a = np.where(np.isnan(da), da.dtype.type(0), da)  # fillna(typ(0)) keeps the float dtype
    elif da.dtype.kind == "b":
        a = da.astype(typ)  # where(notnull, bool, uint) -> uint
    else:
        a = da.astype(np.result_type(da.dtype, typ))
    return cumsum_reset_np(a.copy(), index, typ(1))


def cumsum_reset_xr(da, index, reset_on_zero):
    """rl:154-169 `_cumsum_reset_xr` (float capable; NaN handling differs from the fast track)."""
    da = np.asarray(da)
    if index == "first":
        da = da[::-1]
    with np.errstate(invalid="ignore"):
        cs = np.nancumsum(da, axis=0) if da.dtype.kind == "f" else np.cumsum(da, axis=0)
    if da.dtype.kind == "f":
        # xarray cumsum(skipna=True): NaN treated as 0 in the running sum (result not NaN)
        pass
    cond = (da == 0) if reset_on_zero else np.isnan(da)
    cs2 = where_nan(cs, cond)
    cs2[0] = 0
    # ffill along axis 0
    idx = np.where(~np.isnan(cs2), np.arange(cs2.shape[0]).reshape((-1,) + (1,) * (cs2.ndim - 1)), 0)
    idx = np.maximum.accumulate(idx, axis=0)
    cs2 = np.take_along_axis(cs2, idx, axis=0)
    out = cs - cs2
    if index == "first":
        out = out[::-1]
    return out


def rle(da, index="first"):
    """rl:223-272 `rle`."""
    da = np.asarray(da)
    if da.shape[0] == 0:
        return da.astype(np.float32)
    if index == "first":
        da = da[::-1]
    cs_s = cumsum_reset(da)
    # keep numbers with a 0 to the right (and the last number); NaN neighbours compare False (rl:264)
    with np.errstate(invalid="ignore"):
        nxt = shift0(da.astype(np.float64) if da.dtype.kind != "f" else da, -1, 0)
        cs_s = where_nan(cs_s, nxt == 0)
        pos = da > 0
    out = np.where(pos, cs_s, cs_s.dtype.type(0))  # .where(da > 0, 0)
    if index == "first":
        out = out[::-1]
    return out


def _rl_stat(d, window, reducer):
    """rl:311-327 `get_rl_stat` on one (sub-)array, time on axis 0."""
    with np.errstate(invalid="ignore"):
        ok = d >= window
    dw = where_nan(d, ok)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)
        if reducer == "max":
            stat = np.nanmax(dw, axis=0)
        elif reducer == "min":
            stat = np.nanmin(dw, axis=0)
        elif reducer == "sum":
            stat = np.nansum(dw, axis=0)
        elif reducer == "mean":
            stat = np.nanmean(dw, axis=0)
        elif reducer == "std":
            stat = np.nanstd(dw, axis=0)
        elif reducer == "count":
            stat = np.sum(~np.isnan(dw), axis=0)
        elif reducer.startswith("q") and reducer[1:].isdigit():
            stat = np.nanquantile(dw.astype(np.float64), float(f"0.{reducer[1:]}"), axis=0)
        else:
            raise ValueError(reducer)
    with np.errstate(invalid="ignore"):
        none = (np.isnan(d) | (d < window)).all(axis=0)
    return np.where(none, 0, stat)


def _map_groups(arr, time: OTime, freq, func):
    """``arr.resample(time=freq).map(func)`` with func reducing axis 0."""
    outs = []
    for _, idx in groups(time, freq):
        outs.append(func(arr[idx], time.isel(idx)))
    return np.stack(outs, axis=0)


NPTS_OPT = 9000  # rl:26


def use_ufunc(ufunc_1dim, da, freq=None, index="first"):
    """rl:33-78 with OPTIONS[RUN_LENGTH_UFUNC] = "auto" (the default): grids under 9000 cells take the 1-D ufunc path
    when the runs are indexed by their first step and no resampling follows."""
    da = np.asarray(da)
    if ufunc_1dim is True and freq is not None:
        raise ValueError("Resampling after run length operations is not implemented for 1d method")
    if ufunc_1dim in ("auto", "from_context"):
        ufunc_1dim = (da.size // max(da.shape[0], 1)) < NPTS_OPT
    return bool(index == "first" and ufunc_1dim and freq is None)


def _apply_1d(func, da, *args):
    """xr.apply_ufunc(..., vectorize=True) over the cells: `func` sees one series at a time (rl:1500-1618)."""
    da = np.asarray(da)
    flat = da.reshape(da.shape[0], -1)
    with np.errstate(invalid="ignore"):
        import warnings

        with warnings.catch_warnings():
            warnings.simplefilter("ignore", RuntimeWarning)
            out = np.array([func(flat[:, c], *args) for c in range(flat.shape[1])], dtype=np.float64)
    return out.reshape(da.shape[1:])


def rle_statistics(da, reducer, window, time: OTime | None = None, freq=None, index="first", ufunc_1dim="auto"):
    """rl:275-335: the 1-D ufunc path for small grids (rl:314-316), else the N-D path — resample AFTER the run-length
    encoding when freq is given."""
    if use_ufunc(ufunc_1dim, da, freq, index):
        return _apply_1d(statistics_run_1d, da, reducer, window)
    d = rle(da, index=index)
    if freq is None:
        return _rl_stat(d, window, reducer)
    return _map_groups(d, time, freq, lambda g, _t: _rl_stat(g, window, reducer))


def longest_run(da, time=None, freq=None, index="first", ufunc_1dim="auto"):
    """rl:338-378."""
    return rle_statistics(da, "max", 1, time, freq, index, ufunc_1dim)


def windowed_run_events(da, window, time=None, freq=None, index="first", ufunc_1dim="auto"):
    """rl:381-434."""
    da = np.asarray(da)
    if use_ufunc(ufunc_1dim, da, freq, index):
        return _apply_1d(windowed_run_events_1d, da, window)
    if window == 1:
        shift = 1 if index == "first" else -1
        with np.errstate(invalid="ignore"):
            d = np.where(shift0(da.astype(np.float64), shift, 0) == 0, 1, 0)
            d = np.where(da == 1, d, 0)
    else:
        d = rle(da, index=index)
        with np.errstate(invalid="ignore"):
            d = np.where(d >= window, 1, 0)
    if freq is not None:
        return _map_groups(d, time, freq, lambda g, _t: g.sum(axis=0))
    return d.sum(axis=0)


def windowed_run_count(da, window, time=None, freq=None, index="first", ufunc_1dim="auto"):
    """rl:437-488."""
    da = np.asarray(da)
    if use_ufunc(ufunc_1dim, da, freq, index):
        return _apply_1d(windowed_run_count_1d, da, window)
    if window == 1 and freq is None:
        return np.nansum(da.astype(np.float64) if da.dtype.kind != "f" else da, axis=0)
    d = rle(da, index=index)
    with np.errstate(invalid="ignore"):
        d = np.where(d >= window, d, 0)  # d.where(d >= window, 0): NaN >= window False -> 0
    if freq is not None:
        return _map_groups(d, time, freq, lambda g, _t: g.sum(axis=0))
    return d.sum(axis=0)


def windowed_max_run_sum(da, window, time=None, freq=None, index="first"):
    """rl:491-540."""
    da = np.asarray(da)
    if window == 1 and freq is None:
        return np.nanmax(cumsum_reset_xr(da, index, True), axis=0)
    d_rse = cumsum_reset_xr(da, index, True)
    d_rle = rle((da > 0), index=index)
    with np.errstate(invalid="ignore"):
        d = np.where(d_rle >= window, d_rse, 0)
    if freq is not None:
        return _map_groups(d, time, freq, lambda g, _t: np.nanmax(g, axis=0))
    return np.nanmax(d, axis=0)


def _find_boundary_run(runs, position):
    """rl:591-603 `find_boundary_run` (index output, coord=False)."""
    T = runs.shape[0]
    if position == "last":
        runs = runs[::-1]
    dmax = runs.argmax(axis=0)
    out = np.where(dmax != runs.argmin(axis=0), dmax.astype(np.float64), np.nan)
    if position == "last":
        out = T - out - 1
    return out


def boundary_run(da, window, time=None, freq=None, position="first"):
    """rl:543-640 `_boundary_run`, N-D path (ufunc_1dim=False), coord=False."""
    da = np.asarray(da)
    if da.dtype.kind == "f":
        da = np.where(np.isnan(da), da.dtype.type(0), da)
    if window == 1:
        if freq is not None:
            return _map_groups(da, time, freq, lambda g, _t: _find_boundary_run(g, position))
        return _find_boundary_run(da, position)
    d = cumsum_reset(da, index=position)
    d = np.where(d >= window, 1, 0)
    if freq is not None:
        return _map_groups(d, time, freq, lambda g, _t: _find_boundary_run(g, position))
    return _find_boundary_run(d, position)


def first_run(da, window, time=None, freq=None):
    """rl:643-690."""
    return boundary_run(da, window, time, freq, "first")


def last_run(da, window, time=None, freq=None):
    """rl:693-740."""
    return boundary_run(da, window, time, freq, "last")


def resample_and_rl(da, resample_before_rl, compute, *args, time: OTime, freq, **kwargs):
    """rl:87-132: resample before (cut the series per period, then compute with freq=None) or after."""
    if resample_before_rl:
        return _map_groups(np.asarray(da), time, freq, lambda g, t: compute(g, *args, time=t, freq=None, **kwargs))
    return compute(da, *args, time=time, freq=freq, **kwargs)


# ---- 1-D variants (rl:1334-1618), for small cases ----
def rle_1d(arr):
    """rl:1334-1340 `_rle_1d` + rl:1343-1405: (values, run lengths, start positions)."""
    ia = np.asarray(arr)
    n = len(ia)
    if n == 0:
        return None, None, None
    y = ia[1:] != ia[:-1]
    i = np.append(np.nonzero(y)[0], n - 1)
    rl_ = np.diff(np.append(-1, i))
    pos = np.cumsum(np.append(0, rl_))[:-1]
    return ia[i], rl_, pos


def statistics_run_1d(arr, reducer, window):
    """rl:1408-1437, verbatim logic: NaN steps form their own (value NaN) runs; `v * rl >= window` is False for them."""
    v, rls, _ = rle_1d(np.asarray(arr, dtype=np.float64))
    with np.errstate(invalid="ignore"):
        if not np.any(v) or np.all(v * rls < window):
            return 0
        if reducer == "count":
            return (v * rls >= window).sum()
        kwargs = {}
        if reducer.startswith("q") and reducer[1:].isdigit():
            kwargs["q"] = float(f"0.{reducer[1:]}")
            reducer = "quantile"
        return getattr(np, f"nan{reducer}")(np.where(v * rls >= window, rls, np.nan), **kwargs)


def windowed_run_count_1d(arr, window):
    """rl:1440-1458."""
    v, rls, _ = rle_1d(np.asarray(arr, dtype=np.float64))
    with np.errstate(invalid="ignore"):
        return np.where(v * rls >= window, rls, 0).sum()


def windowed_run_events_1d(arr, window):
    """rl:1461-1480."""
    v, rls, _ = rle_1d(np.asarray(arr, dtype=np.float64))
    with np.errstate(invalid="ignore"):
        return (v * rls >= window).sum()


def first_run_1d(arr, window):
    """rl:1483-1497 (NaN when no run)."""
    v, rls, pos = rle_1d(arr)
    ind = np.where(v * rls >= window, pos, np.inf).min()
    return np.nan if np.isinf(ind) else ind


# ---- hysteresis runs, longest run, seasons (rl:805-1145) ----
def _ffill0(a):
    """ffill along axis 0."""
    idx = np.where(~np.isnan(a), np.arange(a.shape[0]).reshape((-1,) + (1,) * (a.ndim - 1)), 0)
    idx = np.maximum.accumulate(idx, axis=0)
    return np.take_along_axis(a, idx, axis=0)


def runs_with_holes(da_start, window_start, da_stop, window_stop):
    """rl:844-888."""
    a = np.nan_to_num(np.asarray(da_start).astype(np.float64)).astype(int)
    b = np.nan_to_num(np.asarray(da_stop).astype(np.float64)).astype(int)
    start_runs = cumsum_reset(a.astype(bool), index="first")
    stop_runs = cumsum_reset(b.astype(bool), index="first")
    start_pos = np.where(start_runs >= window_start, 1.0, np.nan)
    stop_pos = np.where(stop_runs >= window_stop, 0.0, np.nan)
    runs = np.where(np.isnan(stop_pos), start_pos, stop_pos)  # stop_positions.combine_first(start_positions)
    runs = _ffill0(runs)
    return np.nan_to_num(runs, nan=0.0)


def keep_longest_run(da, time=None, freq=None):
    """rl:805-841."""
    da = np.asarray(da)
    rls = rle(da)

    def _get_out(_rls, _t=None):
        T = _rls.shape[0]
        with warnings.catch_warnings():
            warnings.simplefilter("ignore", RuntimeWarning)
            am = np.nanargmax(np.where(np.isnan(_rls).all(axis=0), 0, _rls), axis=0)
        ar = np.arange(T).reshape((-1,) + (1,) * (_rls.ndim - 1))
        out = np.where(ar == am, _rls + 1, _rls)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore", RuntimeWarning)
            return _ffill0(out) == np.nanmax(out, axis=0)

    if freq is not None:
        return np.concatenate([_get_out(rls[idx]) for _, idx in groups(time, freq)], axis=0)
    return _get_out(rls)


def _date_index(time: OTime, date):
    """rl:1621-1665 `index_of_date` for "MM-DD" strings (every year matches); returns an index array."""
    if date is None:
        return None
    m, d = (int(v) for v in date.split("-"))
    return np.nonzero((time.month == m) & (time.day == d))[0]


def first_run_before_date(da, window, date, time: OTime):
    """rl:1287-1331."""
    da = np.asarray(da, dtype=np.float64)
    if date is not None:
        idx = _date_index(time, date)
        if idx.size == 0:
            return np.full(da.shape[1:], np.nan)
        lim = idx[0] + window - 1
        da = da.copy()
        da[lim:] = np.nan  # da.where(t < t[mid + window - 1])
    return first_run(da, window)


def first_run_after_date(da, window, date, time: OTime):
    """rl:1204-1244."""
    da = np.asarray(da, dtype=np.float64)
    idx = _date_index(time, date)
    mid = 0 if idx is None else (idx[0] if idx.size else None)
    if mid is None:
        return np.full(da.shape[1:], np.nan)
    da = da.copy()
    da[:mid] = np.nan
    return first_run(da, window)


def last_run_before_date(da, window, date, time: OTime):
    """rl:1247-1284."""
    da = np.asarray(da, dtype=np.float64)
    idx = _date_index(time, date)
    if idx.size == 0:
        return np.full(da.shape[1:], np.nan)
    da = da.copy()
    da[idx[0] + 1 :] = np.nan  # da.where(t <= t[mid])
    return last_run(da, window)


def run_end_after_date(da, window, date, time: OTime):
    """rl:1148-1201 with coord=False."""
    da = np.asarray(da).astype(bool)
    T = da.shape[0]
    idx = _date_index(time, date)
    if idx.size == 0:
        return np.full(da.shape[1:], np.nan)
    mid = idx[0]
    nd = (~da).astype(np.float64)
    nd[:mid] = np.nan
    end = first_run(nd, window)
    d2 = da.astype(np.float64)
    d2[mid:] = np.nan
    beg = first_run(d2, window)
    with np.errstate(invalid="ignore"):
        end = np.where(np.isnan(end) & ~np.isnan(beg), T - 1, end)
    return np.where(np.isnan(beg), np.nan, end)


def map_groups_fn(fn, da, time: OTime, freq, *args):
    """resample(time=freq).map(fn): apply a one-group function to every period and stack."""
    return np.stack([fn(np.asarray(da)[idx], *args, time.isel(idx)) for _, idx in groups(time, freq)], axis=0)


def season(da, window, mid_date, time: OTime):
    """rl:998-1110 with coord=False: (start, end, length) index arrays for ONE group (time on axis 0)."""
    da = np.asarray(da).astype(bool)
    T = da.shape[0]
    beg = first_run_before_date(da, window, mid_date, time)
    ar = np.arange(T).reshape((-1,) + (1,) * (da.ndim - 1))
    not_da = np.where(ar >= np.nan_to_num(beg, nan=0.0), (~da).astype(np.float64), np.nan)
    end = first_run_after_date(not_da, window, mid_date, time)
    with np.errstate(invalid="ignore"):
        length = np.where(np.isnan(beg), 0, np.where(np.isnan(end), T - beg, end - beg))
        end = np.where(np.isnan(end) & ~np.isnan(beg), T - 1, end)
        end = np.where(np.isnan(beg), np.nan, end)
    return beg, end, length


def season_per_period(da, window, mid_date, time: OTime, freq):
    """gen:841-853 pattern: resample(time=freq).map(rl.season ...) with index outputs."""
    outs = [season(np.asarray(da)[idx], window, mid_date, time.isel(idx)) for _, idx in groups(time, freq)]
    return tuple(np.stack([o[k] for o in outs], axis=0) for k in range(3))


def run_bounds(mask):
    """rl:745-802 with coord=False: (2, events, *cells) start / end indices of the runs of True, NaN-padded."""
    mask = np.asarray(mask).astype(bool)
    mi = mask.astype(int)
    diff = np.concatenate((mi[:1], np.diff(mi, axis=0)), axis=0)
    nstarts = int((diff == 1).sum(axis=0).max()) if mask.size else 0

    def _get_indices(arr, N):
        out = np.full((N,), np.nan, dtype=float)
        inds = np.where(arr)[0]
        out[: len(inds)] = inds
        return out

    cells = mask.shape[1:]
    starts = np.full((nstarts,) + cells, np.nan)
    ends = np.full((nstarts,) + cells, np.nan)
    for ix in np.ndindex(*cells):
        sl = (slice(None),) + ix
        starts[sl] = _get_indices(diff[sl] == 1, nstarts)
        ends[sl] = _get_indices(diff[sl] == -1, nstarts)
    return np.stack((starts, ends))


def _find_events(da_start, da_stop, data, window_start, window_stop):
    """rl:1760-1842 on one period: dict of (event, *cells) arrays; event_start as a step index."""
    da_start = np.asarray(da_start).astype(bool)
    da_stop = np.asarray(da_stop).astype(bool)
    runs = runs_with_holes(da_start, window_start, da_stop, window_stop)
    event_length = np.nan_to_num(rle(runs).astype(np.float64), nan=0.0)
    eff = cumsum_reset_xr(np.where(runs == 1, da_start.astype(np.float32), np.float32(np.nan)), "first", False)
    eff = np.nan_to_num(eff, nan=0.0).astype(np.int16)
    T = da_start.shape[0]
    cells = da_start.shape[1:]
    nev = int(np.ceil(T / (window_start + window_stop)))
    out = {"event_length": np.full((nev,) + cells, np.nan), "event_effective_length": np.full((nev,) + cells, np.nan),
           "event_start": np.full((nev,) + cells, np.nan)}
    esum = None
    if data is not None:
        data = np.asarray(data)
        esum = cumsum_reset_xr(np.where(runs == 1, data, data.dtype.type(np.nan)), "first", False)
        out["event_sum"] = np.full((nev,) + cells, np.nan)
    tidx = np.arange(T)
    for ix in np.ndindex(*cells):
        sl = (slice(None),) + ix
        sel = event_length[sl] > 0
        k = int(sel.sum())
        out["event_length"][(slice(0, k),) + ix] = event_length[sl][sel]
        out["event_effective_length"][(slice(0, k),) + ix] = eff[sl][sel]
        out["event_start"][(slice(0, k),) + ix] = tidx[sel]
        if esum is not None:
            out["event_sum"][(slice(0, k),) + ix] = esum[sl][sel]
    return out


def find_events(condition, window, condition_stop=None, window_stop=1, data=None, time=None, freq=None):
    """rl:1846-1901."""
    condition = np.asarray(condition).astype(bool)
    if condition_stop is None:
        condition_stop = ~condition
    if freq is None:
        return _find_events(condition, condition_stop, data, window, window_stop)
    parts = [_find_events(condition[idx], np.asarray(condition_stop)[idx], None if data is None else np.asarray(data)[idx],
                          window, window_stop) for _, idx in groups(time, freq)]
    nev = max(p["event_length"].shape[0] for p in parts)

    def pad(a):
        o = np.full((nev,) + a.shape[1:], np.nan)
        o[: a.shape[0]] = a
        return o

    return {k: np.stack([pad(p[k]) for p in parts]) for k in parts[0]}


def suspicious_run_1d(arr, window=10, op=">", thresh=None):
    """rl:1668-1714."""
    import operator

    arr = np.asarray(arr)
    v, rl_, pos = rle_1d(arr)
    sus_runs = rl_ >= window
    if thresh is not None:
        f = {">": operator.gt, "gt": operator.gt, "<": operator.lt, "lt": operator.lt, "==": operator.eq, "eq": operator.eq,
             "!=": operator.ne, "ne": operator.ne, ">=": operator.ge, "ge": operator.ge, "gteq": operator.ge,
             "<=": operator.le, "le": operator.le, "lteq": operator.le}
        if op not in f:
            raise NotImplementedError(f"{op}")
        with np.errstate(invalid="ignore"):
            sus_runs = sus_runs & f[op](v, thresh)
    out = np.zeros_like(arr, dtype=bool)
    for st, length in zip(pos[sus_runs], rl_[sus_runs]):
        out[st : st + length] = True
    return out


def suspicious_run(arr, window=10, op=">", thresh=None):
    """rl:1717-1757: vectorised over the cells (time on axis 0)."""
    arr = np.asarray(arr)
    out = np.zeros(arr.shape, dtype=bool)
    for ix in np.ndindex(*arr.shape[1:]):
        sl = (slice(None),) + ix
        out[sl] = suspicious_run_1d(arr[sl], window, op, thresh)
    return out
