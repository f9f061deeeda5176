"""Host mirror of ``xclim.core.calendar`` for the percentile path (reference: core/calendar.py:395-494, 690-790).

Same names and argument meaning as the reference; arrays are numpy (uploaded) or device arrays with TIME ON AXIS 0
and a :class:`~xclim_amd.timeaxis.TimeAxis` instead of an xarray time coordinate.  All arithmetic runs in the HIP
kernels (``xh_percentile_doy``, ``xh_doy_interp``); this module only builds the integer/real tables they consume.
"""

from __future__ import annotations

import numpy as np

from . import kernels as K
from ._capi import DeviceArray, handle_float64, get_device
from .timeaxis import TimeAxis, _is_leap


class DoyPercentile:
    """Result of :func:`percentile_doy`: device table ``data`` (nper, ndoy, C) float64 + coordinates/attrs.

    Carries what the reference stores in DataArray attrs (cal:487-494): climatology_bounds, window, alpha, beta.
    """

    def __init__(self, data: DeviceArray | None, doys, percentiles, cell_shape, attrs, host=None, device=None):
        """``host``: the same table on the host, (nper, ndoy, *cells) float64 — what the xarray adapter holds when the table
        came in as a DataArray; with it the device copy is made on first use (``data``) and cell blocks of a chunked field
        get their own slab (:meth:`block`).  A callable is evaluated on first use (the adapter recognised the table of an
        earlier ``percentile_doy`` call on the device: its host form is only needed for chunked fields)."""
        self._data = data
        self._host = host
        self._dev = device
        self.dayofyear = np.asarray(doys)
        self.percentiles = np.asarray(percentiles, dtype=np.float64)
        self.cell_shape = tuple(cell_shape)
        self.attrs = dict(attrs)

    @property
    def host(self):
        if callable(self._host):
            self._host = self._host()
        return self._host

    @property
    def data(self) -> DeviceArray:
        if self._data is None:
            h = np.ascontiguousarray(self.host, dtype=np.float64)
            self._data = (self._dev or get_device()).to_device(h.reshape(h.shape[0], h.shape[1], -1), dtype=np.float64)
        return self._data

    def block(self, idx) -> "DoyPercentile":
        """The table of one cell block (``idx``: one slice per cell dimension), uploaded from the host copy."""
        if self.host is None:
            raise ValueError("DoyPercentile.block needs the host table (tables made on the device cover the whole grid)")
        h = np.ascontiguousarray(self.host[(slice(None), slice(None)) + tuple(idx)], dtype=np.float64)
        return DoyPercentile(None, self.dayofyear, self.percentiles, h.shape[2:], self.attrs, host=h, device=self._dev)

    def sel(self, percentiles) -> "DoyPercentile":
        j = int(np.nonzero(self.percentiles == percentiles)[0][0])
        nper, nd, C = self.data.shape
        sub = self.data.dev.wrap(self.data.ptr + j * nd * C * 8, (1, nd, C), np.float64)
        sub._owner = self.data
        return DoyPercentile(sub, self.dayofyear, [percentiles], self.cell_shape, self.attrs)

    def values(self) -> np.ndarray:
        """numpy array (ndoy, *cells, nper) — the reference's dim order up to the position of dayofyear."""
        a = self.data.get()  # (nper, ndoy, C)
        return np.moveaxis(a, 0, -1).reshape((len(self.dayofyear),) + self.cell_shape + (len(self.percentiles),))


def doy_interp_tables(n_src: int, doy_max: int, doy_min: int = 1):
    """Index/offset tables of ``_interpolate_doy_calendar`` (cal:716-726) in scipy interp1d form.

    The n_src source rows are relabelled ``linspace(doy_min, doy_max, n_src)`` and evaluated at the integers
    doy_min..doy_max: idx = searchsorted(x, x_new) clipped to [1, n-1], lo = idx-1, hi = idx.
    """
    x = np.linspace(doy_min, doy_max, n_src)
    xn = np.arange(doy_min, doy_max + 1).astype(np.float64)
    idx = np.clip(np.searchsorted(x, xn), 1, n_src - 1)
    lo, hi = idx - 1, idx
    return lo.astype(np.int32), hi.astype(np.int32), xn - x[lo], x[hi] - x[lo]


def _flatten(arr, dev, f64: bool = False):
    """(T, *cells) numpy/device array -> (DeviceArray (T, C), cell_shape).  ``f64``: the caller has a float64 kernel — a
    float64 field is uploaded as it is; otherwise it is refused (or rounded, ``_capi.handle_float64``)."""
    if isinstance(arr, DeviceArray):
        if arr.dtype != np.float32 and not (f64 and arr.dtype == np.float64):
            # the kernels read float32 fields: a uint8 / float64 buffer must not be reinterpreted
            raise TypeError(f"device arrays handed to the kernels must be float32, got {np.dtype(arr.dtype).name} "
                            "(masks: use the float mask of compare(..., keep=True))")
        cell_shape = arr.shape[1:]
        return arr.reshape(arr.shape[0], -1), cell_shape
    a = np.asarray(arr)
    cell_shape = a.shape[1:]
    # large float32 / float64 fields are recognised again across calls (Device.resident): percentile_doy and the index that
    # consumes its table, an index and the missing-value check after it read the same buffer — one PCIe transfer, not two
    if f64 and a.dtype == np.float64:
        return dev.resident(a.reshape(a.shape[0], -1)) if a.flags.c_contiguous else dev.to_device(a.reshape(a.shape[0], -1)), cell_shape
    handle_float64(a, "field")
    if a.dtype == np.float32 and a.flags.c_contiguous:
        return dev.resident(a.reshape(a.shape[0], -1)), cell_shape
    return dev.to_device(a.reshape(a.shape[0], -1), dtype=np.float32), cell_shape


_SEASON_OF_MONTH = np.array([0, 0, 0, 1, 1, 1, 2, 2, 2, 3, 3, 3, 0])  # index = month (1..12): DJF MAM JJA SON
_SEASONS = {"DJF": 0, "MAM": 1, "JJA": 2, "SON": 3}
_CUM_LEAP = np.array([0, 31, 60, 91, 121, 152, 182, 213, 244, 274, 305, 335])
_CUM_NOLEAP = np.array([0, 31, 59, 90, 120, 151, 181, 212, 243, 273, 304, 334])


def _get_doys(start: int, end: int, inclusive):
    """cal:1137-1163."""
    doys = np.arange(start, end + 1) if start <= end else np.concatenate((np.arange(start, 367), np.arange(0, end + 1)))
    if not inclusive[0]:
        doys = doys[1:]
    if not inclusive[1]:
        doys = doys[:-1]
    return doys


def select_time_mask(time: TimeAxis, *, season=None, month=None, doy_bounds=None, date_bounds=None, include_bounds=True):
    """The boolean time mask of ``select_time`` (cal:1259-1378); None when no indexer is given."""
    n = sum(a is not None for a in (season, month, doy_bounds, date_bounds))
    if n > 1:
        raise ValueError(f"Only one method of indexing may be given, got {n}.")
    if n == 0:
        return None
    if isinstance(include_bounds, bool):
        include_bounds = (include_bounds, include_bounds)
    if season is not None:
        want = [season] if isinstance(season, str) else list(season)
        return np.isin(_SEASON_OF_MONTH[time.month], [_SEASONS[s] for s in want])
    if month is not None:
        return np.isin(time.month, [month] if np.isscalar(month) else list(month))
    if doy_bounds is not None:
        if not all(isinstance(b, (int, np.integer)) for b in doy_bounds):
            raise NotImplementedError("array-like doy bounds give a (time, cells) mask: use select_time()")
        return np.isin(time.doy, _get_doys(int(doy_bounds[0]), int(doy_bounds[1]), include_bounds))
    start, end = date_bounds
    (ms, ds), (me, de) = (tuple(int(v) for v in b.split("-")) for b in (start, end))
    if time.calendar in ("360_day",):
        doy_t, s, e = time.doy, (ms - 1) * 30 + ds, (me - 1) * 30 + de
    elif time.calendar in ("noleap", "365_day"):
        doy_t, s, e = time.doy, _CUM_NOLEAP[ms - 1] + ds, _CUM_NOLEAP[me - 1] + de
    else:  # non-uniform calendars (and all_leap): every date seen in the all_leap calendar (cal:1354-1371)
        doy_t, s, e = _CUM_LEAP[time.month - 1] + time.day, _CUM_LEAP[ms - 1] + ds, _CUM_LEAP[me - 1] + de
    return np.isin(doy_t, _get_doys(int(s), int(e), include_bounds))


def _days_since_bounds(time: TimeAxis, start, end, bounds_time: TimeAxis, freq: str, cell_shape, include_bounds):
    """cal:1211-1246 on the host: the bounds (P_b, *cells) of the periods labelled by `bounds_time` -> (seg, lo, hi) with lo /
    hi (P, C) float32 in days since the first step of each period of ``time.segments(freq)``.  doy_to_days_since
    (cal:1004-1072): a doy before the period's first doy lies in the NEXT year (+ days in the year); NaN -> 0 / 366;
    periods of `time` that the bounds do not label stay masked (lo = +inf)."""
    seg, starts = time.segments(freq)
    if starts and len(starts[0]) != 2:
        raise NotImplementedError("doy bounds with a time dimension: yearly / quarterly / monthly bounds only")
    P = len(seg) - 1
    C_ = int(np.prod(cell_shape, dtype=np.int64))
    s = np.broadcast_to(np.asarray(start, dtype=np.float64), (len(bounds_time),) + tuple(cell_shape)).reshape(len(bounds_time), C_).copy()
    e = np.broadcast_to(np.asarray(end, dtype=np.float64), (len(bounds_time),) + tuple(cell_shape)).reshape(len(bounds_time), C_).copy()
    if not include_bounds[0]:
        s += 1
    if not include_bounds[1]:
        e -= 1
    lo = np.full((P, C_), np.inf, dtype=np.float32)
    hi = np.full((P, C_), -np.inf, dtype=np.float32)
    key = {(int(y), int(m), int(d)): i for i, (y, m, d) in enumerate(zip(bounds_time.year, bounds_time.month, bounds_time.day))}
    ordinal = time.ordinal()
    for p in range(P):
        t0 = int(seg[p])
        if seg[p + 1] <= t0:
            continue
        ly, lm = starts[p]  # the period's LABEL (its first day, present in the data or not): `base_time` of cal:1229
        i = key.get((int(ly), int(lm), 1))
        if i is None:
            continue  # "This group has no defined bounds: put False in the mask" (cal:1239-1243)
        label = TimeAxis(np.array([ly]), np.array([lm]), np.array([1]), time.calendar)
        base = float(label.doy[0])
        # days in the label's year (_doy_days_since_doys: doy_max, cal:982-987)
        ylen = 360.0 if time.calendar == "360_day" else (366.0 if bool(_is_leap(int(ly), time.calendar)) else 365.0)
        with np.errstate(invalid="ignore"):
            sd = np.where(s[i] >= base, s[i], s[i] + ylen) - base
            ed = np.where(e[i] >= base, e[i], e[i] + ylen) - base
        off = float(ordinal[t0] - label.ordinal()[0])  # the kernel counts days from the period's first PRESENT step
        lo[p] = np.where(np.isnan(sd), 0.0, sd) - off
        hi[p] = np.where(np.isnan(ed), 366.0, ed) - off
    return seg, lo, hi


def _infer_bounds_freq(bt: TimeAxis) -> str:
    """xr.infer_freq for the two shapes the reference's own callers produce: yearly labels (same month / day in consecutive
    years -> "YS" / "YS-MON") and monthly labels (day 1 of consecutive months -> "MS")."""
    mon = ["JAN", "FEB", "MAR", "APR", "MAY", "JUN", "JUL", "AUG", "SEP", "OCT", "NOV", "DEC"]
    if len(bt) >= 2 and np.all(bt.month == bt.month[0]) and np.all(bt.day == 1) and np.all(np.diff(bt.year) == 1):
        return "YS" if bt.month[0] == 1 else f"YS-{mon[int(bt.month[0]) - 1]}"
    if len(bt) >= 2 and np.all(bt.day == 1) and np.all(np.diff(bt.year * 12 + bt.month) == 1):
        return "MS"
    raise ValueError("doy bounds with a time dimension: cannot infer the frequency of their time coordinate; pass bounds_freq=")


def select_time(da, time: TimeAxis, drop: bool = False, *, season=None, month=None, doy_bounds=None, date_bounds=None,
                include_bounds=True, bounds_time: TimeAxis | None = None, bounds_freq: str | None = None, device=None, keep=False):
    """core/calendar.py:1259-1378: ``da.where(mask, drop=drop)`` for the time selections season / month / doy_bounds /
    date_bounds (the ``**indexer`` of select_resample_op & co.).  drop=False: same length, NaN outside the selection;
    drop=True: ``(selected rows, their TimeAxis)``.  keep=True returns the (rows, cells) float32 device array.
    Array-like ``doy_bounds``: per cell ``(*cells)``, or — with ``bounds_time`` = the TimeAxis of the bounds' own time
    coordinate (period starts) — per period and cell ``(P_b, *cells)`` (mask_between_doys, cal:1166-1257)."""
    dev = device or get_device()
    if doy_bounds is not None and not all(isinstance(b, (int, np.integer)) for b in doy_bounds):
        # per-cell bounds (mask_between_doys, cal:1199-1257, bounds without a time dimension)
        if sum(a is not None for a in (season, month, date_bounds)) > 0:
            raise ValueError("Only one method of indexing may be given, got 2.")
        if drop:
            raise ValueError("Passing array-like doy bounds is incompatible with drop=True.")  # cal:1343-1345
        x, cell_shape = _flatten(da, dev)
        inc = (include_bounds, include_bounds) if isinstance(include_bounds, bool) else tuple(include_bounds)
        start, end = doy_bounds
        if bounds_time is not None:
            # bounds WITH a time dimension (cal:1211-1246): one pair per period of the bounds' own frequency, compared as
            # days since the period's first step; steps of periods the bounds do not label are masked
            freq = bounds_freq or _infer_bounds_freq(bounds_time)
            seg, lo, hi = _days_since_bounds(time, start, end, bounds_time, freq, cell_shape, inc)
            out = K.mask_days_cells(dev, x, seg, dev.to_device(lo), dev.to_device(hi))
            return out if keep else out.get().reshape((out.shape[0],) + tuple(cell_shape))
        bs, be = (np.broadcast_to(np.asarray(b, dtype=np.float32), cell_shape).reshape(-1).copy() for b in (start, end))
        if not inc[0]:
            bs += 1  # (NaN stays NaN: an open bound)
        if not inc[1]:
            be -= 1
        out = K.mask_doy_cells(dev, x, time.doy, dev.to_device(bs), dev.to_device(be))
        return out if keep else out.get().reshape((out.shape[0],) + tuple(cell_shape))
    mask = select_time_mask(time, season=season, month=month, doy_bounds=doy_bounds, date_bounds=date_bounds,
                            include_bounds=include_bounds)
    x, cell_shape = _flatten(da, dev)
    if mask is None:
        out, sub = x, time
    elif drop:
        rows = np.nonzero(mask)[0]
        out, sub = K.select_rows(dev, x, rows), time.subset(rows)
    else:
        out, sub = K.select_rows(dev, x, np.where(mask, np.arange(len(mask)), -1)), time
    res = out if keep else out.get().reshape((out.shape[0],) + tuple(cell_shape))
    return (res, sub) if drop else res


def percentile_doy(arr, time: TimeAxis, window: int = 5, per=10.0, alpha: float = 1.0 / 3.0, beta: float = 1.0 / 3.0,
                   copy: bool = True, device=None) -> DoyPercentile:
    """Percentile value for each day of the year (cal:395-494).

    ``copy`` is accepted for signature parity; the device kernels never mutate their input, so it is a no-op.
    """
    dev = device or get_device()
    x, cell_shape = _flatten(arr, dev)
    if len(time) != x.shape[0]:
        raise ValueError("time axis length does not match the data")
    tb, years, doys = time.doy_table()
    pers = [per] if np.isscalar(per) else list(per)
    p = K.percentile_doy(dev, x, tb, window, pers, alpha, beta)  # (nper, ndoy, C)
    if doys.max() == 366:
        # cal:484-485: drop doy 366 and re-interpolate 1..365 -> 1..366 (adjust_doy_calendar towards `arr`)
        keep = doys < 366
        nper, nd, C = p.shape
        nsrc = int(keep.sum())
        max_t, min_t = int(time.doy.max()), int(time.doy.min())
        i0, i1, dxn, dxs = doy_interp_tables(nsrc, max_t, min_t)
        out = dev.empty((nper, len(i0), C), np.float64)
        for j in range(nper):
            src = dev.wrap(p.ptr + j * nd * C * 8, (nsrc, C), np.float64)  # rows 0..nsrc-1 are doys < 366
            res = K.doy_interp(dev, src, i0, i1, dxn, dxs, xsrc=doys[keep])  # interpolate_na in the doy coordinate
            dev.call("xh_memcpy_d2d", out.ptr + j * len(i0) * C * 8, res.ptr, res.nbytes)
            dev.sync()
        p = out
        doys = np.arange(min_t, max_t + 1)
    attrs = {
        "window": window,
        "alpha": alpha,
        "beta": beta,
        "climatology_bounds": [f"{time.year[0]:04d}-{time.month[0]:02d}-{time.day[0]:02d}",
                               f"{time.year[-1]:04d}-{time.month[-1]:02d}-{time.day[-1]:02d}"],
        "history": "percentile_doy(arr, window=%d, per=%s, alpha=%r, beta=%r)" % (window, pers, alpha, beta),
    }
    return DoyPercentile(p, doys, pers, cell_shape, attrs)


def climatological_mean_doy(arr, time: TimeAxis, window: int = 5, *, device=None):
    """cal:907-931: (mean, std) per day of year over all years and a centred `window`; numpy (ndoy, *cells) float32,
    plus the day-of-year coordinate."""
    dev = device or get_device()
    x, cell_shape = _flatten(arr, dev)
    tb, years, doys = time.doy_table()
    m, s = K.doy_mean_std(dev, x, tb, window)
    shp = (len(doys),) + tuple(cell_shape)
    return m.get().reshape(shp), s.get().reshape(shp), doys


def adjust_doy_calendar(source: DoyPercentile, target_time: TimeAxis, device=None) -> DoyPercentile:
    """cal:729-760: re-grid the doy axis when the source does not span the target calendar's full year."""
    dev = device or get_device()
    if int(source.dayofyear.max()) == target_time.max_doy():
        return source
    max_t, min_t = int(target_time.doy.max()), int(target_time.doy.min())
    nper, nd, C = source.data.shape
    i0, i1, dxn, dxs = doy_interp_tables(nd, max_t, min_t)
    out = dev.empty((nper, len(i0), C), np.float64)
    for j in range(nper):
        src = dev.wrap(source.data.ptr + j * nd * C * 8, (nd, C), np.float64)
        res = K.doy_interp(dev, src, i0, i1, dxn, dxs, xsrc=source.dayofyear)
        dev.call("xh_memcpy_d2d", out.ptr + j * len(i0) * C * 8, res.ptr, res.nbytes)
        dev.sync()
    return DoyPercentile(out, np.arange(min_t, max_t + 1), source.percentiles, source.cell_shape, source.attrs)


def resample_doy_index(doy: DoyPercentile, time: TimeAxis):
    """cal:763-790 as an index table: row of the (adjusted) doy table for each time step (the gather itself is
    fused into ``xh_threshold_count``; the (T, Y, X) fp64 temporary of the reference is never materialised)."""
    pos = np.searchsorted(doy.dayofyear, time.doy)
    pos_c = np.clip(pos, 0, len(doy.dayofyear) - 1)
    if not np.all(doy.dayofyear[pos_c] == time.doy):
        raise ValueError("day-of-year table does not cover every day of the target time axis")
    return pos_c.astype(np.int32)


def resample_doy(doy: DoyPercentile, time: TimeAxis, *, device=None, keep=False):
    """cal:763-790: the (T, *cells) float64 field whose step t holds the doy table's row for dayofyear(t) (after
    adjust_doy_calendar).  The index functions do not need it (the gather is fused into xh_threshold_count)."""
    dev = device or get_device()
    adoy = adjust_doy_calendar(doy, time, dev)
    if adoy.data.shape[0] != 1:
        raise ValueError("select one percentile first (DoyPercentile.sel)")
    table = adoy.data.reshape(adoy.data.shape[1], adoy.data.shape[2])
    out = K.doy_broadcast(dev, table, resample_doy_index(adoy, time))
    return out if keep else out.get().reshape((len(time),) + doy.cell_shape)


def within_bnds_doy(arr, *, low: DoyPercentile, high: DoyPercentile, time: TimeAxis, device=None, keep=False):
    """cal:934-954: (low_doy < arr) & (arr < high_doy) with the bounds broadcast by day of year."""
    dev = device or get_device()
    x, cell_shape = _flatten(arr, dev)
    lo, hi = adjust_doy_calendar(low, time, dev), adjust_doy_calendar(high, time, dev)
    if lo.data.shape[0] != 1 or hi.data.shape[0] != 1:
        raise ValueError("select one percentile first (DoyPercentile.sel)")
    tl, th = resample_doy_index(lo, time), resample_doy_index(hi, time)
    if not np.array_equal(tl, th):
        raise ValueError("low and high must share their dayofyear coordinate")
    out = K.within_bnds_doy(dev, x, lo.data.reshape(lo.data.shape[1], -1), hi.data.reshape(hi.data.shape[1], -1), tl)
    return out if keep else out.get().reshape((x.shape[0],) + tuple(cell_shape)).astype(bool)
