"""Host mirror of ``xclim.indices.run_length`` (reference: src/xclim/indices/run_length.py) over the HIP kernels.

Inputs are masks with TIME ON AXIS 0 (bool, or float32 with NaN where ``select_time`` masked values); results are
identical for the reference's N-D and 1-D (``ufunc_1dim``) code paths, so that option is accepted and ignored
except for the one error the reference raises (rl:67-68).
"""

from __future__ import annotations

import numpy as np

from . import kernels as K
from ._capi import DeviceArray, get_device
from .calendar import _flatten
from .timeaxis import TimeAxis

npts_opt = 9000  # rl:26 — kept for API parity; no dispatch depends on it here


def use_ufunc(ufunc_1dim, da=None, dim="time", freq=None, index="first") -> bool:
    """rl:33-78: only the argument validation matters on this backend."""
    if ufunc_1dim is True and freq is not None:
        raise ValueError("Resampling after run length operations is not implemented for 1d method")
    return False


def _mask(da, dev):
    if isinstance(da, DeviceArray):
        return da.reshape(da.shape[0], -1), da.shape[1:]
    a = np.asarray(da)
    if a.dtype != np.float32:
        a = a.astype(np.float32)
    return _flatten(a, dev)


def _whole(T):
    return np.array([0, T], dtype=np.int64)


def _run(da, stat, window, time, freq, index, device, keep, cut=False):
    dev = device or get_device()
    m, cell_shape = _mask(da, dev)
    if freq is None:
        seg = _whole(m.shape[0])
        cut = True
    else:
        seg, _ = time.segments(freq)
    out, _ = K.run_stats(dev, m, stat, window, seg, cut=cut, index=index, want_valid=False)
    if keep:
        return out
    o = out.get().reshape((out.shape[0],) + tuple(cell_shape))
    return o[0] if freq is None else o


def _cumsum_reset(da, dim="time", index="last", *, device=None, keep=False):
    """rl:172-219."""
    dev = device or get_device()
    m, cell_shape = _mask(da, dev)
    out = K.cumsum_reset(dev, m, index)
    return out if keep else out.get().reshape((m.shape[0],) + tuple(cell_shape))


def rle(da, dim="time", index="first", *, device=None, keep=False):
    """rl:223-272."""
    dev = device or get_device()
    m, cell_shape = _mask(da, dev)
    out = K.rle(dev, m, index)
    return out if keep else out.get().reshape((m.shape[0],) + tuple(cell_shape))


def rle_statistics(da, reducer: str, window: int, dim="time", freq=None, ufunc_1dim="from_context", index="first", *,
                   time: TimeAxis | None = None, device=None, keep=False):
    """rl:275-335.  ``freq`` given -> resample AFTER the run-length encoding."""
    use_ufunc(ufunc_1dim, freq=freq, index=index)
    if reducer.startswith("q"):
        raise NotImplementedError("quantile run statistics are not wired to the HIP path yet")
    return _run(da, reducer, window, time, freq, index, device, keep)


def longest_run(da, dim="time", freq=None, ufunc_1dim="from_context", index="first", *, time=None, device=None,
                keep=False):
    """rl:338-378."""
    return rle_statistics(da, "max", 1, dim, freq, ufunc_1dim, index, time=time, device=device, keep=keep)


def windowed_run_events(da, window: int, dim="time", freq=None, ufunc_1dim="from_context", index="first", *, time=None,
                        device=None, keep=False):
    """rl:381-434."""
    use_ufunc(ufunc_1dim, freq=freq, index=index)
    return _run(da, "count", window, time, freq, index, device, keep)


def windowed_run_count(da, window: int, dim="time", freq=None, ufunc_1dim="from_context", index="first", *, time=None,
                       device=None, keep=False):
    """rl:437-488 (window == 1 and freq None: plain sum, rl:478-479)."""
    use_ufunc(ufunc_1dim, freq=freq, index=index)
    stat = "plainsum" if (window == 1 and freq is None) else "sum"
    return _run(da, stat, window, time, freq, index, device, keep)


def first_run(da, window: int, dim="time", freq=None, coord=None, ufunc_1dim="from_context", *, time=None, device=None,
              keep=False):
    """rl:643-690 (index output; coord lookups are host work on the 1-D time axis)."""
    return _run(da, "first", window, time, freq, "first", device, keep, cut=(window == 1))


def last_run(da, window: int, dim="time", freq=None, coord=None, ufunc_1dim="from_context", *, time=None, device=None,
             keep=False):
    """rl:693-740."""
    return _run(da, "last", window, time, freq, "first", device, keep, cut=(window == 1))


def windowed_max_run_sum(da, window: int, dim="time", freq=None, index="first", *, time=None, device=None, keep=False):
    """rl:491-540 for non-negative NaN-free values (the documented use: precipitation sums over wet runs).  With
    `freq` the runs are cut at the period edges (resample-before semantics)."""
    dev = device or get_device()
    m, cell_shape = _mask(da, dev)
    seg = _whole(m.shape[0]) if freq is None else time.segments(freq)[0]
    out = K.max_run_sum(dev, m, window, seg)
    if keep:
        return out
    o = out.get().reshape((out.shape[0],) + tuple(cell_shape))
    return o[0] if freq is None else o


def runs_with_holes(da_start, window_start: int, da_stop, window_stop: int, dim="time", *, device=None, keep=False):
    """rl:844-888."""
    dev = device or get_device()
    a, cell_shape = _mask(da_start, dev)
    b, _ = _mask(da_stop, dev)
    out = K.runs_with_holes(dev, a, window_start, b, window_stop)
    return out if keep else out.get().reshape((a.shape[0],) + tuple(cell_shape))


def keep_longest_run(da, dim="time", freq=None, *, time=None, device=None, keep=False):
    """rl:805-841."""
    dev = device or get_device()
    m, cell_shape = _mask(da, dev)
    seg = _whole(m.shape[0]) if freq is None else time.segments(freq)[0]
    out = K.keep_longest_run(dev, m, seg)
    return out if keep else out.get().reshape((m.shape[0],) + tuple(cell_shape)).astype(bool)


def index_of_date(time: TimeAxis, date: str | None, max_idxs=None, default: int = 0) -> np.ndarray:
    """rl:1621-1665 for "MM-DD" and "YYYY-MM-DD" strings."""
    if date is None:
        return np.array([default])
    parts = date.split("-")
    if len(parts) == 2:
        m, d = int(parts[0]), int(parts[1])
        cond = (time.month == m) & (time.day == d)
    else:
        y, m, d = (int(p) for p in parts)
        cond = (time.year == y) & (time.month == m) & (time.day == d)
    idxs = np.where(cond)[0]
    if max_idxs is not None and idxs.size > max_idxs:
        raise ValueError(f"More than {max_idxs} instance of date {date} found in the coordinate array.")
    return idxs


def season(da, window: int, mid_date: str | None = None, dim="time", stat=None, coord=False, *, time: TimeAxis | None = None,
           freq: str | None = None, device=None):
    """rl:998-1110, mapped per period when `freq` is given (the gen.season pattern, gen:841-853).

    Returns {"start", "end", "length"} as numpy arrays; coord=False -> indices relative to the period start,
    coord="dayofyear" -> day of year of those steps (NaN preserved).
    """
    dev = device or get_device()
    m, cell_shape = _mask(da, dev)
    T = m.shape[0]
    seg = _whole(T) if freq is None else time.segments(freq)[0]
    P = len(seg) - 1
    mid = None
    if mid_date is not None:
        mid = np.full(P, -1, dtype=np.int32)
        for p in range(P):
            idx = index_of_date(time.subset(slice(int(seg[p]), int(seg[p + 1]))), mid_date, max_idxs=1)
            if idx.size:
                mid[p] = idx[0]
    s, e, ln = (a.get() for a in K.season(dev, m, window, seg, mid))
    if coord:
        if coord != "dayofyear":
            raise NotImplementedError("only coord='dayofyear' is supported")
        for arr in (s, e):
            for p in range(P):
                ok = ~np.isnan(arr[p])
                arr[p, ok] = time.doy[int(seg[p]) + arr[p, ok].astype(np.int64)]
    shp = (P,) + tuple(cell_shape)
    out = {"start": s.reshape(shp), "end": e.reshape(shp), "length": ln.reshape(shp)}
    if freq is None:
        out = {k: v[0] for k, v in out.items()}
    return out


def season_length(da, window: int, mid_date: str | None = None, dim="time", *, time=None, freq=None, device=None):
    """rl:1113-1145."""
    return season(da, window, mid_date, time=time, freq=freq, device=device)["length"]


def season_start(da, window: int, mid_date: str | None = None, dim="time", coord=False, *, time=None, freq=None, device=None):
    """rl:891-929 (= first_run_before_date)."""
    return season(da, window, mid_date, coord=coord, time=time, freq=freq, device=device)["start"]


def _date_bounded(da, window, date, time, freq, device, *, side, invert=False, last=False):
    """Shared body of the date-bounded run functions (rl:1148-1331): mask rows relative to `date`, then first/last run.

    side: "ge" keep t >= date ; "le" keep t <= date ; "lt" keep t < date ; "lt_w" keep t < date + window - 1.
    Returns ((P, C) float32 indices relative to the period start, seg, cell_shape)."""
    dev = device or get_device()
    m, cell_shape = _mask(da, dev)
    T = m.shape[0]
    seg = _whole(T) if freq is None else time.segments(freq)[0]
    P = len(seg) - 1
    lo = np.zeros(P, dtype=np.int32)
    hi = np.zeros(P, dtype=np.int32)
    absent = np.zeros(P, dtype=bool)
    for p in range(P):
        n = int(seg[p + 1] - seg[p])
        if date is None:
            lo[p], hi[p] = 0, n
            continue
        idx = index_of_date(time.subset(slice(int(seg[p]), int(seg[p + 1]))), date, max_idxs=None if side == "le" else 1,
                            default=-1 if side == "le" else 0)
        if idx.size == 0:
            absent[p] = True
            continue
        mid = int(idx[0])
        lo[p], hi[p] = {"ge": (mid, n), "le": (0, mid + 1), "lt": (0, mid), "lt_w": (0, min(n, mid + window - 1))}[side]
    masked = K.mask_rows(dev, m, seg, lo, hi, invert=invert)
    out, _ = K.run_stats(dev, masked, "last" if last else "first", window, seg, cut=True, want_valid=False)
    res = out.get()
    res[absent] = np.nan  # the date is not within the group (rl:1235-1237)
    return res, seg, cell_shape


def _to_coord(res, seg, coord, time):
    if coord:
        if coord != "dayofyear":
            raise NotImplementedError("only coord='dayofyear' is supported")
        for p in range(res.shape[0]):
            ok = ~np.isnan(res[p])
            res[p, ok] = time.doy[int(seg[p]) + res[p, ok].astype(np.int64)]
    return res


def _shape(res, cell_shape, freq):
    out = res.reshape((res.shape[0],) + tuple(cell_shape))
    return out[0] if freq is None else out


def first_run_after_date(da, window: int, date: str | None = "07-01", dim="time", coord=False, *, time=None, freq=None,
                         device=None):
    """rl:1204-1244."""
    res, seg, cs = _date_bounded(da, window, date, time, freq, device, side="ge")
    return _shape(_to_coord(res, seg, coord, time), cs, freq)


def last_run_before_date(da, window: int, date: str = "07-01", dim="time", coord=False, *, time=None, freq=None, device=None):
    """rl:1247-1284."""
    res, seg, cs = _date_bounded(da, window, date, time, freq, device, side="le", last=True)
    return _shape(_to_coord(res, seg, coord, time), cs, freq)


def first_run_before_date(da, window: int, date: str | None = "07-01", dim="time", coord=False, *, time=None, freq=None,
                          device=None):
    """rl:1287-1331."""
    res, seg, cs = _date_bounded(da, window, date, time, freq, device, side="lt_w")
    return _shape(_to_coord(res, seg, coord, time), cs, freq)


def run_end_after_date(da, window: int, date: str = "07-01", dim="time", coord=False, *, time=None, freq=None, device=None):
    """rl:1148-1201: end of the first run that started before `date` (first run of `window` False after the date)."""
    end, seg, cs = _date_bounded(da, window, date, time, freq, device, side="ge", invert=True)
    beg, _, _ = _date_bounded(da, window, date, time, freq, device, side="lt")
    lens = np.diff(seg).astype(np.float32)[:, None]
    with np.errstate(invalid="ignore"):
        end = np.where(np.isnan(end) & ~np.isnan(beg), lens - 1, end)
        end = np.where(np.isnan(beg), np.nan, end).astype(np.float32)
    return _shape(_to_coord(end, seg, coord, time), cs, freq)


_STATS = {"rle_statistics", "longest_run", "windowed_run_events", "windowed_run_count", "first_run", "last_run"}


def resample_and_rl(da, resample_before_rl: bool, compute, *args, freq: str, time: TimeAxis, dim="time", device=None,
                    keep=False, **kwargs):
    """rl:87-132: cut the series per period first (default of every index) or resample after."""
    name = compute.__name__
    if name not in _STATS:
        raise NotImplementedError(f"resample_and_rl: {name} is not available on the HIP path")
    if not resample_before_rl:
        return compute(da, *args, freq=freq, time=time, device=device, keep=keep, **kwargs)
    # resample before: same kernel with cut_at_segments = 1
    params = dict(kwargs)
    names = {"rle_statistics": ("reducer", "window"), "longest_run": (), "windowed_run_events": ("window",),
             "windowed_run_count": ("window",), "first_run": ("window",), "last_run": ("window",)}[name]
    params.update(dict(zip(names, args)))
    index = params.get("index", "first")
    window = params.get("window", 1)
    if name == "rle_statistics":
        stat = params["reducer"]
    elif name == "longest_run":
        stat = "max"
    elif name == "windowed_run_events":
        stat = "count"
    elif name == "windowed_run_count":
        stat = "plainsum" if window == 1 else "sum"
    else:
        stat = "first" if name == "first_run" else "last"
    dev = device or get_device()
    m, cell_shape = _mask(da, dev)
    seg, _ = time.segments(freq)
    out, _ = K.run_stats(dev, m, stat, window, seg, cut=True, index=index, want_valid=False)
    return out if keep else out.get().reshape((out.shape[0],) + tuple(cell_shape))
