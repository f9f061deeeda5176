"""Host mirror of ``xclim.indices.run_length`` (reference: src/xclim/indices/run_length.py) over the HIP kernels.

Inputs are masks with TIME ON AXIS 0 (bool, or float32 with NaN where ``select_time`` masked values).  The reference's
N-D and 1-D (``ufunc_1dim``) code paths are the same kernels here; the one place where they disagree — a run next to a
NaN step keeps its length in the 1-D path and loses it in the N-D path — follows the reference's own dispatch
(:func:`use_ufunc`: grids under 9000 cells without resampling take the 1-D semantics).
"""

from __future__ import annotations

import numpy as np

from . import kernels as K
from ._capi import DeviceArray, get_device
from .calendar import _flatten
from .timeaxis import TimeAxis

npts_opt = 9000  # rl:26 — kept for API parity; no dispatch depends on it here


RUN_LENGTH_UFUNC = "auto"  # the reference's option OPTIONS[RUN_LENGTH_UFUNC] (core/options.py), read for "from_context"


def use_ufunc(ufunc_1dim, da=None, dim="time", freq=None, index="first") -> bool:
    """rl:33-78.  The reference runs small grids (< `npts_opt` cells, "auto") through its 1-D ufunc path when no
    resampling follows and runs are indexed by their first step.  Both paths are the same kernels here; they differ in
    ONE thing the kernels reproduce (``xh_run_stats`` index mode 2): the 1-D path lets a run next to a NaN step keep its
    length, the N-D path (rle, rl:223-272) drops it."""
    if ufunc_1dim is True and freq is not None:
        raise ValueError("Resampling after run length operations is not implemented for 1d method")
    if ufunc_1dim == "from_context":
        ufunc_1dim = RUN_LENGTH_UFUNC
    if ufunc_1dim == "auto":
        if da is None:
            ufunc_1dim = False
        else:
            shape = tuple(da.shape) if isinstance(da, DeviceArray) else np.shape(da)
            ufunc_1dim = int(np.prod(shape[1:], dtype=np.int64)) < npts_opt
    return bool((index == "first") and ufunc_1dim and (freq is None))


def _mask(da, dev):
    if isinstance(da, DeviceArray):
        if da.dtype != np.float32:
            raise TypeError(f"device masks must be float32 (1 / 0 / NaN), got {np.dtype(da.dtype).name}: "
                            "compare(..., keep=True) returns one")
        return da.reshape(da.shape[0], -1), da.shape[1:]
    a = np.asarray(da)
    if a.dtype == np.bool_ or a.dtype == np.uint8:
        # boolean masks cross PCIe as bytes and become the float mask on the device (no host astype of the field)
        m8 = dev.to_device(np.ascontiguousarray(a).reshape(a.shape[0], -1).view(np.uint8))
        return K.mask_to_f32(dev, m8), a.shape[1:]
    if a.dtype != np.float32:
        a = a.astype(np.float32)
    return _flatten(a, dev)


def _whole(T):
    return np.array([0, T], dtype=np.int64)


def _run(da, stat, window, time, freq, index, device, keep, cut=False, ufunc_1dim="from_context", stat_form=False):
    dev = device or get_device()
    one_dim = use_ufunc(ufunc_1dim, da, freq=freq, index=index)
    if one_dim and stat_form:
        one_dim = "stat"  # rle_statistics goes through statistics_run_1d
    m, cell_shape = _mask(da, dev)
    if freq is None:
        seg = _whole(m.shape[0])
        cut = True
    else:
        seg, _ = time.segments(freq)
    out, _ = K.run_stats(dev, m, stat, window, seg, cut=cut, index=index, want_valid=False, one_dim=one_dim)
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
    if reducer.startswith("q") and reducer[1:].isdigit():
        return _run_quantile(da, float(f"0.{reducer[1:]}"), window, time, freq, index, device, keep,
                             use_ufunc(ufunc_1dim, da, freq=freq, index=index))
    return _run(da, reducer, window, time, freq, index, device, keep, ufunc_1dim=ufunc_1dim, stat_form=True)


def _run_quantile(da, q, window, time, freq, index, device, keep, one_dim=False):
    """rl:318-327 with reducer "qNN": d.where(d >= window).quantile(q) (linear / Hyndman-Fan type 7, NaN-skipping) of the
    run lengths of each period, 0 where the period holds no run of at least `window`.  `one_dim` (statistics_run_1d,
    rl:1408-1437): NaN steps only break runs, and a series with NaN steps but no qualifying run gives NaN."""
    dev = device or get_device()
    m, cell_shape = _mask(da, dev)
    T, C_ = m.shape
    src = K.compare_map(dev, m, ">", 0.0, "maskf") if one_dim else m  # NaN -> 0: the run next to it keeps its length
    d = K.compare_map(dev, K.rle(dev, src, index), ">=", float(window), "where")
    seg = _whole(T) if freq is None else time.segments(freq)[0]
    P = len(seg) - 1
    out = dev.empty((P, C_), np.float32)
    for p in range(P):
        t0, t1 = int(seg[p]), int(seg[p + 1])
        if t1 <= t0:
            continue
        view = DeviceArray(dev, d.ptr + t0 * C_ * 4, (t1 - t0, C_), np.float32, owner=False)
        row = DeviceArray(dev, out.ptr + p * C_ * 4, (1, C_), np.float32, owner=False)
        K.quantile_series(dev, view, [q], out=row)
    o = out.get()
    if one_dim:  # np.nanquantile of an empty selection is NaN — reached only when the series holds a NaN step
        _, valid = K.resample_reduce(dev, m, "count", seg)
        has_nan = valid.get() < np.diff(seg).reshape(-1, 1)
        o = np.where(np.isnan(o) & ~has_nan, 0.0, o).astype(np.float32)
    else:
        o = np.nan_to_num(o, nan=0.0)  # no qualifying run -> 0 (rl:326)
    if keep:
        return dev.to_device(o)
    o = o.reshape((P,) + tuple(cell_shape))
    return o[0] if freq is None else o


def longest_run(da, dim="time", freq=None, ufunc_1dim="from_context", index="first", *, time=None, device=None,
                keep=False):
    """rl:338-378."""
    return rle_statistics(da, "max", 1, dim, freq, ufunc_1dim, index, time=time, device=device, keep=keep)


def windowed_run_events(da, window: int, dim="time", freq=None, ufunc_1dim="from_context", index="first", *, time=None,
                        device=None, keep=False):
    """rl:381-434."""
    return _run(da, "count", window, time, freq, index, device, keep, ufunc_1dim=ufunc_1dim)


def windowed_run_count(da, window: int, dim="time", freq=None, ufunc_1dim="from_context", index="first", *, time=None,
                       device=None, keep=False):
    """rl:437-488 (window == 1 and freq None: plain sum, rl:478-479)."""
    stat = "plainsum" if (window == 1 and freq is None) else "sum"
    return _run(da, stat, window, time, freq, index, device, keep, ufunc_1dim=ufunc_1dim)


def first_run(da, window: int, dim="time", freq=None, coord=None, ufunc_1dim="from_context", *, time=None, device=None,
              keep=False):
    """rl:643-690.  ``coord`` ("dayofyear", "year", "month", "day") maps the index through the time axis (host work on the
    small result, rl:586-597); it needs ``time``."""
    return _boundary(da, "first", window, time, freq, coord, device, keep)


def last_run(da, window: int, dim="time", freq=None, coord=None, ufunc_1dim="from_context", *, time=None, device=None,
             keep=False):
    """rl:693-740 (``coord`` as in :func:`first_run`)."""
    return _boundary(da, "last", window, time, freq, coord, device, keep)


def _boundary(da, stat, window, time, freq, coord, device, keep):
    res = _run(da, stat, window, time, freq, "first", device, keep, cut=(window == 1))
    if not coord:
        return res
    if keep:
        raise ValueError("coord lookups are applied to host results: use keep=False")
    if time is None:
        raise ValueError("coord needs the time axis (time=TimeAxis)")
    seg = _whole(len(time)) if freq is None else time.segments(freq)[0]
    flat = np.array(res, dtype=np.float64).reshape(len(seg) - 1, -1)
    return _to_coord(flat, seg, coord, time).reshape(np.shape(res))


def windowed_max_run_sum(da, window: int, dim="time", freq=None, index="first", *, time=None, device=None, keep=False,
                         cut=False):
    """rl:491-540: largest sum of the values of a run (da > 0) of at least `window` steps.  With `freq` the reference
    resamples AFTER (run sums and lengths cross the period edges, a run counts for the period of its first step);
    ``cut=True`` cuts the runs at the period edges instead — what :func:`resample_and_rl` does with
    ``resample_before_rl=True`` (the default of hot_spell_max_magnitude)."""
    if index not in ("first", "last"):
        raise ValueError(f"index must be 'first' or 'last', got {index!r}")
    dev = device or get_device()
    m, cell_shape = _mask(da, dev)
    T = m.shape[0]
    seg = _whole(T) if freq is None else time.segments(freq)[0]
    if index == "last":
        # the reference flips the series (rle / _cumsum_reset with index="last"): the run sum is accumulated FORWARD and
        # sits on the run's last step.  The kernel marches backward and credits the first step: run it on the reversed
        # series with the reversed segments (same fp32 summation order), then put the periods back in order.
        m = K.select_rows(dev, m, np.arange(T - 1, -1, -1, dtype=np.int64))
        seg = (T - np.asarray(seg)[::-1]).astype(np.int64)
    out = K.max_run_sum(dev, m, window, seg, cut=(cut or freq is None))
    if index == "last" and out.shape[0] > 1:
        out = K.select_rows(dev, out, np.arange(out.shape[0] - 1, -1, -1, dtype=np.int64))
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
        s, e = _to_coord(s, seg, coord, time), _to_coord(e, seg, coord, time)
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


def _coord_table(time: TimeAxis, coord) -> np.ndarray:
    """The 1-D array that ``coord`` selects on the time axis (rl:690, 740: ``da[dim].dt.<coord>`` looked up at the run's
    index through utils.lazy_indexing).  The lookup itself is host work on (period, cell) results."""
    names = {"dayofyear": "doy", "year": "year", "month": "month", "day": "day"}
    if coord is True:
        return time.dates()  # rl:589-592: the coordinate itself (dates)
    if coord not in names:
        raise NotImplementedError(f"coord={coord!r}: supported are True (dates) and the datetime fields {sorted(names)}")
    return np.asarray(getattr(time, names[coord]))


def _to_coord(res, seg, coord, time):
    """Indices (relative to the period start, NaN = no run) -> coordinate values at those steps.  ``coord=True`` gives the
    dates themselves: datetime64[ns] with NaT where there is no run (strings / None for calendars without datetime64)."""
    if coord:
        table = _coord_table(time, coord)
        if coord is True:
            dates = np.full(res.shape, np.datetime64("NaT") if table.dtype.kind == "M" else None, dtype=table.dtype)
            for p in range(res.shape[0]):
                ok = ~np.isnan(res[p])
                dates[p, ok] = table[int(seg[p]) + res[p, ok].astype(np.int64)]
            return dates
        for p in range(res.shape[0]):
            ok = ~np.isnan(res[p])
            res[p, ok] = table[int(seg[p]) + res[p, ok].astype(np.int64)]
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


_STATS = {"rle_statistics", "longest_run", "windowed_run_events", "windowed_run_count", "first_run", "last_run",
          "windowed_max_run_sum"}


def resample_and_rl(da, resample_before_rl: bool, compute, *args, freq: str, time: TimeAxis, dim="time", device=None,
                    keep=False, **kwargs):
    """rl:87-132: cut the series per period first (default of every index) or resample after."""
    name = compute.__name__
    if name not in _STATS:
        raise NotImplementedError(f"resample_and_rl: {name} is not available on the HIP path")
    if not resample_before_rl:
        return compute(da, *args, freq=freq, time=time, device=device, keep=keep, **kwargs)
    # resample before: same kernel with cut_at_segments = 1
    if name == "windowed_max_run_sum":
        return compute(da, *args, freq=freq, time=time, device=device, keep=keep, cut=True, **kwargs)
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
    # every period is handed to `compute` with freq=None (rl:122-129): small grids take the 1-D NaN semantics
    one_dim = use_ufunc(params.get("ufunc_1dim", "from_context"), da, freq=None, index=index)
    if one_dim and name in ("rle_statistics", "longest_run"):
        one_dim = "stat"
    m, cell_shape = _mask(da, dev)
    seg, _ = time.segments(freq)
    out, _ = K.run_stats(dev, m, stat, window, seg, cut=True, index=index, want_valid=False, one_dim=one_dim)
    return out if keep else out.get().reshape((out.shape[0],) + tuple(cell_shape))


def season_end(da, window: int, mid_date: str | None = None, dim="time", coord=False, *, time: TimeAxis | None = None,
               device=None):
    """rl:932-995 (stand-alone form: a season with a start but no end ends on the last step)."""
    return season(da, window, mid_date, dim, coord=coord, time=time, device=device)["end"]


def run_bounds(mask, dim="time", coord=False, *, time: TimeAxis | None = None, device=None):
    """rl:745-802: (2, events, *cells) float array of [start, end) step indices of every run of True values, the events
    axis as long as the cell with most runs needs and NaN-padded.  ``end`` is the first step AFTER the run (the position
    of the -1 in the differenced mask), NaN for a run that reaches the end.  coord="dayofyear" maps the indices."""
    dev = device or get_device()
    m, cell_shape = _mask(mask, dev)
    T = m.shape[0]
    seg = _whole(T)
    nruns, _ = K.run_stats(dev, m, "count", 1, seg, cut=True, want_valid=False)
    n = int(np.nanmax(nruns.get())) if m.shape[1] else 0
    ev = K.run_events(dev, m, seg, n, want=("start", "end"))
    out = np.stack([ev["start"].get()[0], ev["end"].get()[0]]).astype(np.float64).reshape((2, n) + tuple(cell_shape))
    if coord:
        table = _coord_table(time, coord)
        ok = ~np.isnan(out)
        out[ok] = table[out[ok].astype(np.int64)]
    return out


def find_events(condition, window: int, condition_stop=None, window_stop: int = 1, data=None, freq: str | None = None, *,
                time: TimeAxis | None = None, device=None):
    """rl:1846-1901 / 1760-1842.  Returns a dict of (event, *cells) arrays (or (period, event, *cells) with ``freq``):
    event_length, event_effective_length, event_start (step index relative to the period start; the reference converts
    it to a date with the time coordinate) and event_sum when ``data`` is given; NaN past the last event.  The event
    axis has ceil(T / (window + window_stop)) entries like the reference's."""
    dev = device or get_device()
    a, cell_shape = _mask(condition, dev)
    T, C_ = a.shape
    if condition_stop is None:
        b = K.compare_map(dev, a, "==", 0.0, "maskf")  # ~condition
    else:
        b, _ = _mask(condition_stop, dev)
    dat = None
    if data is not None:
        dat, _ = _flatten(np.asarray(data, dtype=np.float32) if not isinstance(data, DeviceArray) else data, dev)
    seg = _whole(T) if freq is None else time.segments(freq)[0]
    P = len(seg) - 1
    lens = np.diff(seg)
    nev = int(np.ceil((int(lens.max()) if P else 0) / (window + window_stop)))
    # runs with holes are found independently in every period (resample(...).map in the reference)
    runs = dev.empty((T, C_), np.float32)
    for p in range(P):
        t0, t1 = int(seg[p]), int(seg[p + 1])
        if t1 <= t0:
            continue
        va = DeviceArray(dev, a.ptr + t0 * C_ * 4, (t1 - t0, C_), np.float32, owner=False)
        vb = DeviceArray(dev, b.ptr + t0 * C_ * 4, (t1 - t0, C_), np.float32, owner=False)
        r = K.runs_with_holes(dev, va, window, vb, window_stop)
        dev.copy_d2d(runs.ptr + t0 * C_ * 4, r.ptr, (t1 - t0) * C_ * 4)
    want = ("start", "len", "eff") + (("sum",) if dat is not None else ())
    ev = K.run_events(dev, runs, seg, nev, eff=a, data=dat, want=want)
    shp = (P, nev) + tuple(cell_shape)
    out = {"event_length": ev["len"].get().reshape(shp), "event_effective_length": ev["eff"].get().reshape(shp),
           "event_start": ev["start"].get().reshape(shp)}
    if dat is not None:
        out["event_sum"] = ev["sum"].get().reshape(shp)
    if freq is None:
        out = {k: v[0] for k, v in out.items()}
    return out


def suspicious_run(arr, dim="time", window: int = 10, op: str = ">", thresh=None, *, device=None, keep=False):
    """rl:1717-1757: True on the steps that belong to a run of at least ``window`` identical values (optionally only
    values satisfying ``op thresh``)."""
    from .generic import get_op

    dev = device or get_device()
    x, cell_shape = _flatten(np.asarray(arr, dtype=np.float32) if not isinstance(arr, DeviceArray) else arr, dev)
    out = K.suspicious_run(dev, x, window, get_op(op) if thresh is not None else None, thresh)
    return out if keep else out.get().reshape((x.shape[0],) + tuple(cell_shape)).astype(bool)


# ---- 1-D "ufunc" variants (rl:1334-1618): same results as the N-D functions on this backend -------------------------
def rle_1d(arr):
    """rl:1334-1389: (values, run lengths, start positions) of the runs of identical values of a 1-D array.  Pure host
    bookkeeping (variable-length output), restated with numpy."""
    ia = np.asarray(arr)
    n = len(ia)
    if n == 0:
        import warnings

        warnings.warn("run length array empty", stacklevel=2)
        return np.array(np.nan), 0, np.array(np.nan)
    y = ia[1:] != ia[:-1]
    i = np.append(np.nonzero(y)[0], n - 1)
    rl_ = np.diff(np.append(-1, i))
    pos = np.cumsum(np.append(0, rl_))[:-1]
    return ia[i], rl_, pos


def statistics_run_1d(arr, reducer: str, window: int, *, device=None):
    """rl:1436-1465."""
    return rle_statistics(np.asarray(arr)[:, None], reducer, window, device=device)[0]


def windowed_run_count_1d(arr, window: int, *, device=None):
    """rl:1468-1486."""
    return windowed_run_count(np.asarray(arr)[:, None], window, device=device)[0]


def windowed_run_events_1d(arr, window: int, *, device=None):
    """rl:1489-1507."""
    return windowed_run_events(np.asarray(arr)[:, None], window, device=device)[0]


def first_run_1d(arr, window: int, *, device=None):
    """rl:1392-1414."""
    return first_run(np.asarray(arr)[:, None], window, device=device)[0]


def statistics_run_ufunc(x, reducer: str, window: int, dim="time", *, device=None):
    """rl:1531-1562 (vectorised 1-D form == N-D form here)."""
    return rle_statistics(x, reducer, window, dim, device=device)


def windowed_run_count_ufunc(x, window: int, dim="time", *, device=None):
    """rl:1565-1590."""
    return windowed_run_count(x, window, dim, device=device)


def windowed_run_events_ufunc(x, window: int, dim="time", *, device=None):
    """rl:1510-1528."""
    return windowed_run_events(x, window, dim, device=device)


def first_run_ufunc(x, window: int, dim="time", *, device=None):
    """rl:1593-1618."""
    return first_run(x, window, dim, device=device)


_FULL_SHAPE = ("_cumsum_reset", "rle", "keep_longest_run", "runs_with_holes", "suspicious_run")  # results shaped like the input


def _time_dim_only(fn):
    """The kernels march along axis 0.  ``dim``: "time" (axis 0, the run dimension of every index function) or — the mirrors
    take plain arrays, which have no dimension names — an INTEGER axis (rl:223, 275, 338: any ``dim`` of the DataArray):
    that axis is moved first (a view; the upload copies it once), the same kernels run, full-shape results get their axis
    back.  Resampling (``freq`` / ``time``) belongs to the time axis, so it needs ``dim="time"``.  Any other name is refused
    rather than ignored."""
    import functools
    import inspect

    sig = inspect.signature(fn)

    @functools.wraps(fn)
    def wrapper(*args, **kwargs):
        ba = sig.bind_partial(*args, **kwargs)
        dim = ba.arguments.get("dim", "time")
        if isinstance(dim, (int, np.integer)) and not isinstance(dim, bool):
            if ba.arguments.get("freq") is not None or ba.arguments.get("time") is not None:
                raise ValueError(f"{fn.__name__}: freq / time resample the time axis: use dim='time' (axis 0)")
            if ba.arguments.get("keep") or ba.arguments.get("coord"):
                raise NotImplementedError(f"{fn.__name__}: keep= / coord= need the run axis first (dim='time')")
            ax = int(dim)
            for name in ("da", "arr", "x", "mask", "da_start", "da_stop"):
                v = ba.arguments.get(name)
                if v is not None:
                    if isinstance(v, DeviceArray):
                        raise NotImplementedError(f"{fn.__name__}: device arrays must have the run axis first (dim='time')")
                    ba.arguments[name] = np.moveaxis(np.asarray(v), ax, 0)
            ba.arguments["dim"] = "time"
            out = fn(*ba.args, **ba.kwargs)
            return np.moveaxis(out, 0, ax) if fn.__name__ in _FULL_SHAPE else out
        if dim != "time":
            raise NotImplementedError(f"{fn.__name__}: dim is 'time' (axis 0) or an integer axis on the HIP path, got {dim!r}")
        return fn(*args, **kwargs)

    return wrapper


for _name, _fn in list(globals().items()):
    if callable(_fn) and getattr(_fn, "__module__", None) == __name__ and not isinstance(_fn, type):
        try:
            import inspect as _inspect

            if "dim" in _inspect.signature(_fn).parameters:
                globals()[_name] = _time_dim_only(_fn)
        except (TypeError, ValueError):  # pragma: no cover
            pass
del _name, _fn
